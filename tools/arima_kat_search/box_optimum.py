#!/usr/bin/env python3
"""Companion of search.py: the CONVERGED optimum of the box-constrained conditional sum of squares (raw coefficients in
[-0.99, 0.99], L-BFGS-B to machine tolerance, two starts) for every order / constant convention, ranked by AICc.

Finding (results/box_optimum.txt): ARIMA(2,1,1) with a free constant has its box-constrained CSS optimum at the corner
phi = (-0.99, -0.99), theta = -0.8888 and forecasts 18.0145126 -- 2.4e-5 (1.3e-6 relative) from the reference's 18.014537,
for every optimiser, start and objective scaling that converges (the cluster at 18.01451-18.01452 in results/summary.txt).
It is the only consistent estimator found inside the north star's 1e-5; but a search that also visits p = 3 (exact fit,
18.000000) or the better-ranked (2,1,2) / (2,1,3) never ends there, so reproducing the pin needs the crate's own search
order and optimiser budget, not just its estimator.

    python tools/arima_kat_search/box_optimum.py > tools/arima_kat_search/results/box_optimum.txt
"""
import sys, numpy as np, math
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import search as S
from scipy import optimize
TARGET=S.TARGET
rows=[]
for d in (0,1,2):
  for p in range(0,4):
    for q in range(0,4):
      for const in ("none","mean_fixed","mean_free","icpt_free"):
        for cond in ("p","zero"):
            m=S.Model(p,d,q,const,"raw",cond)
            if m.dim==0:
                s,nu=m.sse(np.zeros(0)); x=np.zeros(0)
            else:
                f=m.objective("sse")
                bnd=[(-0.99,0.99)]*(p+q)+[(None,None)]*(m.dim-p-q)
                best=None
                for st in ("zero","tenth"):
                    for x0 in (m.start(st),):
                        r=optimize.minimize(f,x0,method="L-BFGS-B",bounds=bnd,options={"maxiter":5000,"ftol":1e-15,"gtol":1e-12})
                        if best is None or r.fun<best.fun: best=r
                x=best.x; s=best.fun; nu=m.n-(p if cond=="p" else 0)
            k=p+q+(1 if const!="none" else 0)+1
            n=m.n
            s2=max(s/nu,1e-300)
            aic=nu*math.log(s2)+2*k
            aicc=aic+2*k*(k+1)/max(nu-k-1,1e-9)
            rows.append((d,p,q,const,cond,s,aicc,aic,m.forecast1(x),x))
for d in (0,1,2):
    for cond in ("p","zero"):
        sub=[r for r in rows if r[0]==d and r[4]==cond]
        sub.sort(key=lambda r:r[6])
        print("d",d,"cond",cond,"best by AICc:")
        for r in sub[:6]:
            print("   ARIMA(%d,%d,%d) %-10s sse %.6g aicc %.3f yhat1 %.7f diff %.2e x=%s"%(r[1],r[0],r[2],r[3],r[5],r[6],r[8],abs(r[8]-TARGET),np.round(r[9],4)))
print("---- d=1 cond=p, all orders p<=2,q<=3 by AICc")
sub=[r for r in rows if r[0]==1 and r[4]=="p" and r[1]<=2]
sub.sort(key=lambda r:r[6])
for r in sub[:25]:
    print("   ARIMA(%d,%d,%d) %-10s sse %.6g aicc %.3f aic %.3f yhat1 %.7f diff %.2e x=%s"%(r[1],r[0],r[2],r[3],r[5],r[6],r[7],r[8],abs(r[8]-TARGET),np.round(r[9],5)))
