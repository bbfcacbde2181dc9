#!/usr/bin/env python3
"""Write cases.json for compare_with_rust: the inputs the oracle and the HIP path are tested with.

  * the reference's known-answer series (test/sql/ts_model_distinctness.test:24-31) for every model on the hot path;
  * the M5-shape synthetic batches of bench.py (anofox-forecast_amd/synth.py; series regenerate from their ids):
    the first N series of the intermittent and of the strictly positive variant, T = 1,913, h = 28, m = 7, for AutoETS,
    every valid explicit ETS spec (on a few series) and AutoARIMA -- the configurations BASELINE.json names, none of which
    the reference's own tests pin numerically;
  * periods other than 7 (2, 12, 24, 52, 168) on shorter series.

Usage: python make_cases.py [--n 32] [--out cases.json]      (run from anywhere; needs numpy only)"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from anofox_forecast_amd import synth  # noqa: E402

KAT = [10, 12, 14, 11, 13, 15, 12, 14, 16, 13, 15, 17, 14, 16, 18, 15, 17, 19, 16, 18, 20, 17, 19, 21]
SPECS = ["ANN", "AAN", "AAdN", "ANA", "AAA", "AAdA", "MNN", "MAN", "MAdN", "MMN", "MMdN", "AMN", "AMdN", "MNM", "MAM", "MAdM", "MMM",
         "MMdM", "ANM", "AAM", "AAdM", "AMA", "AMdA", "AMM", "AMdM"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cases.json"))
    a = ap.parse_args()
    cases = []
    for model in ("SES", "SESOptimized", "Holt", "HoltWinters", "SeasonalES", "SeasonalESOptimized", "AutoETS", "AutoARIMA"):
        cases.append(dict(id=f"kat/{model}", model=model, values=[float(v) for v in KAT], horizon=3, period=0))
    T, h, m = 1913, 28, 7
    for positive in (False, True):
        Y = synth.gen_series(synth.SEED_M5, 0, a.n, T, m, positive)
        tag = "m5pos" if positive else "m5"
        for s in range(a.n):
            v = Y[s].tolist()
            cases.append(dict(id=f"{tag}/{s}/AutoETS", model="AutoETS", values=v, horizon=h, period=m))
            if s < 8:
                cases.append(dict(id=f"{tag}/{s}/AutoARIMA", model="AutoARIMA", values=v, horizon=h, period=m))
                cases.append(dict(id=f"{tag}/{s}/AutoETS/nonseasonal", model="AutoETS", values=v, horizon=h, period=0))
                for pool in ("reduced", "damped_trend_only"):
                    cases.append(dict(id=f"{tag}/{s}/AutoETS/{pool}", model="AutoETS", values=v, horizon=h, period=m, model_pool=pool))
            if s < 4:
                for spec in SPECS:
                    if not positive and "M" in spec:
                        continue
                    cases.append(dict(id=f"{tag}/{s}/ETS/{spec}", model="ETS", ets_model=spec, values=v, horizon=h, period=m))
                for model in ("HoltWinters", "Holt", "SESOptimized", "SeasonalESOptimized", "SeasonalES"):
                    cases.append(dict(id=f"{tag}/{s}/{model}", model=model, values=v, horizon=h, period=m))
    for period, length in ((2, 60), (12, 144), (24, 240), (52, 260), (168, 1008)):
        Y = synth.gen_series(synth.SEED_STRESS, 0, 4, length, period, True)
        for s in range(4):
            v = Y[s].tolist()
            for model in ("AutoETS", "AutoARIMA", "HoltWinters"):
                cases.append(dict(id=f"period{period}/{s}/{model}", model=model, values=v, horizon=12, period=period))
            cases.append(dict(id=f"period{period}/{s}/ETS/AAA", model="ETS", ets_model="AAA", values=v, horizon=12, period=period))
    with open(a.out, "w") as fh:
        json.dump(cases, fh)
    print(f"{len(cases)} cases -> {a.out}")


if __name__ == "__main__":
    main()
