// arima.hip -- AutoARIMA on gfx950: differencing tests, stepwise (p,q,P,Q,constant) search with a
// conditional-sum-of-squares fit per candidate, forecast + integration.  Lane <-> series; every objective
// evaluation is one streamed pass over the lane's column of the differenced time-major block W[t * ld + s].
//
// Reference call site: crates/anofox-fcst-core/src/forecast.rs:1435-1521 (AutoARIMAConfig::default()
// [.with_seasonal_period(m)]); the arithmetic is in the un-vendored anofox-forecast 0.15.3 crate, so this
// follows the published Hyndman-Khandakar procedure exactly as restated by the CPU checker (oracle/arima.c):
// D by seasonal strength > 0.64, d by KPSS (lag trunc(3 sqrt(n)/13), 0.463), CSS over tanh-PACF coefficients
// minimised by Nelder-Mead (absolute initial steps), stepwise neighbourhood search on AICc.
//
// Per-lane search state machine: candidate models differ between lanes, but the CSS recursion is generic in the
// expanded lag polynomials (kept in LDS per lane), so the wave always executes ONE code path: "evaluate the next
// trial point of whatever model this lane is fitting".  No MFMA (scalar recursions).
#include <hip/hip_runtime.h>
#include <type_traits>

#include "det_math.hpp"
#include "kernels.hpp"
#include "nm.hpp"

namespace anofox {

constexpr int AR_MAXP = 5, AR_MAXSP = 2, AR_MAXORDER = 5, AR_MAXDIM = 6, AR_MAXMODELS = 94;

__device__ __forceinline__ int ar_wave_max(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { int w = __shfl_xor(v, o); v = w > v ? w : v; }
    return v;
}

__device__ __forceinline__ double ar_tanh(double u)
{
    if (u > 20.0) return 1.0;
    if (u < -20.0) return -1.0;
    double e2 = dm_exp(2.0 * u);
    return (e2 - 1.0) / (e2 + 1.0);
}

struct ArOrd { int p, q, P, Q, c; };
__device__ __forceinline__ int ar_dim(const ArOrd &o) { return o.p + o.q + o.P + o.Q + o.c; }
__device__ __forceinline__ int ar_key(const ArOrd &o) { return (((o.p * 6 + o.q) * 3 + o.P) * 3 + o.Q) * 2 + o.c; }

// partial autocorrelations -> AR coefficients (Durbin-Levinson), k <= 5
__device__ __forceinline__ void ar_pacf(const double *u, int k, double *phi)
{
    double work[AR_MAXP];
    for (int j = 0; j < k; j++) {
        double a = ar_tanh(u[j]);
        for (int i = 0; i < j; i++) work[i] = phi[i] - a * phi[j - 1 - i];
        for (int i = 0; i < j; i++) phi[i] = work[i];
        phi[j] = a;
    }
}

// LDS layout per wave (lane-minor): [ring: R slots x 64 lanes x {e, v}][simplex 42 + values 7][acoef L1][bcoef L1][tried 11 words]
// The ring holds e_t and v_t of step t side by side in slot t & (R - 1), R = the power of two >= 2 m + 6, so that one
// 128-bit LDS access moves both and the slot index is a wave-uniform mask (no wrap arithmetic per lane).
typedef double ar_ev_t __attribute__((ext_vector_type(2)));
__host__ __device__ inline int ar_ring_slots(int m) { int r = 8; while (r < 2 * m + 6) r <<= 1; return r; }
struct ArLds {
    double *base; int L1; int R;
    __device__ double *coef() const { return base + (size_t)2 * R * NM_BLOCK; }
    __device__ double &sim(int k, int i) const { return coef()[(k * AR_MAXDIM + i) * NM_BLOCK + threadIdx.x]; }
    __device__ double &fs(int k) const { return coef()[((AR_MAXDIM + 1) * AR_MAXDIM + k) * NM_BLOCK + threadIdx.x]; }
    __device__ double &a(int k) const { return coef()[(49 + k) * NM_BLOCK + threadIdx.x]; }
    __device__ double &b(int k) const { return coef()[(49 + L1 + k) * NM_BLOCK + threadIdx.x]; }
    __device__ double e_at(int t) const { return base[((size_t)(t & (R - 1)) * NM_BLOCK + threadIdx.x) * 2]; }
    __device__ uint32_t &tried(int wd) const { return ((uint32_t *)(coef() + (size_t)(49 + 2 * L1) * NM_BLOCK))[wd * NM_BLOCK + threadIdx.x]; }
};
static size_t ar_lds_bytes(int m)
{
    const int L1 = AR_MAXP + AR_MAXSP * m + 1;
    return sizeof(double) * (size_t)(2 * ar_ring_slots(m) + 49 + 2 * L1 + 11) * NM_BLOCK;
}

// the four factor polynomials of a trial point, zero padded (registers)
struct ArFac { double phi[AR_MAXP], th[AR_MAXP], Phi[AR_MAXSP], Th[AR_MAXSP]; double mu; int nc; };
__device__ __forceinline__ void ar_factors(const ArOrd &o, int m, const double *x, ArFac &f)
{
#pragma unroll
    for (int i = 0; i < AR_MAXP; i++) { f.phi[i] = 0.0; f.th[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < AR_MAXSP; i++) { f.Phi[i] = 0.0; f.Th[i] = 0.0; }
    int k = 0;
    ar_pacf(x + k, o.p, f.phi); k += o.p;
    ar_pacf(x + k, o.q, f.th); k += o.q;
    ar_pacf(x + k, o.P, f.Phi); k += o.P;
    ar_pacf(x + k, o.Q, f.Th); k += o.Q;
    f.mu = o.c ? x[k] : 0.0;
    f.nc = o.p + m * o.P;
}

// expanded lag polynomials of the trial point x into LDS
__device__ __forceinline__ void ar_build_poly(const ArOrd &o, int m, const double *x, const ArLds &L, int &La, int &Lb, double &mu)
{
    double phi[AR_MAXP], th[AR_MAXP], Phi[AR_MAXSP], Th[AR_MAXSP];
    int k = 0;
    ar_pacf(x + k, o.p, phi); k += o.p;
    ar_pacf(x + k, o.q, th); k += o.q;
    ar_pacf(x + k, o.P, Phi); k += o.P;
    ar_pacf(x + k, o.Q, Th); k += o.Q;
    mu = o.c ? x[k] : 0.0;
    La = o.p + m * o.P;
    Lb = o.q + m * o.Q;
    for (int i = 0; i <= La; i++) L.a(i) = 0.0;
    for (int i = 1; i <= o.p; i++) L.a(i) = phi[i - 1];
    for (int I = 1; I <= o.P; I++) {
        L.a(m * I) = L.a(m * I) + Phi[I - 1];
        for (int i = 1; i <= o.p; i++) L.a(m * I + i) = L.a(m * I + i) - phi[i - 1] * Phi[I - 1];
    }
    for (int i = 0; i <= Lb; i++) L.b(i) = 0.0;
    for (int i = 1; i <= o.q; i++) L.b(i) = th[i - 1];
    for (int I = 1; I <= o.Q; I++) {
        L.b(m * I) = L.b(m * I) + Th[I - 1];
        for (int i = 1; i <= o.q; i++) L.b(m * I + i) = L.b(m * I + i) - th[i - 1] * Th[I - 1];
    }
    for (int i = 0; i <= Lb; i++) L.b(i) = -L.b(i);
}

// One CSS pass for the whole wave, CASCADED form (oracle/arima.c css_eval): fixed 5 + 2 + 5 + 2 fused
// multiply-adds per step whatever the lane's orders (absent coefficients are exact zeros), the short lags in
// VGPR shift registers, the seasonal lags in LDS rings (slot t % ring, read before write), w streamed from HBM.
//     v_t = w'_t - sum phi_i w'_{t-i} ; z_t = v_t - sum Phi_I v_{t-mI} ; u_t = z_t + sum theta_j u_{t-j} ;
//     e_t = u_t + sum Theta_J e_{t-mJ}   (z, u, e from t >= nc = p + m P on)
// MODE 0: seasonal lags read from the LDS ring step by step (m = 2, 3: a lag can fall inside a sub-block);
// MODE 1: m >= 4, every seasonal lag of a 4-step sub-block was produced before it, so its 8 ring slots are fetched up
//         front (independent LDS reads) and the four steps run in registers; MODE 2: m = 1, no seasonal factors at all.
// The step is branch-free.  A block of S steps runs ungated when every live lane is inside its sample for the whole block
// and past its warm-up (base >= max nc, base + S <= min len over the live lanes; finished lanes compute into their own
// slots and are ignored); otherwise the gated variant predicates the ring stores on t < len and selects exact zeros for
// the warm-up and past-the-end steps (selects, not arithmetic on possibly non-finite padding).
template <int MODE>
__device__ __noinline__ double ar_css_pass_impl(const double *w, size_t ld, int len, int wave_len_v, bool live, const ArFac &fin, int m_v,
                                                const ArLds &L)
{
    // out of line on purpose: the pass gets its own register allocation (the search kernel around it is a large state
    // machine), and the coefficients are copied out of the caller's (scratch-resident) block once per pass
    double phi[AR_MAXP], th[AR_MAXP], Phi[AR_MAXSP], Th[AR_MAXSP];
#pragma unroll
    for (int q = 0; q < AR_MAXP; q++) { phi[q] = -fin.phi[q]; th[q] = fin.th[q]; }
#pragma unroll
    for (int q = 0; q < AR_MAXSP; q++) { Phi[q] = -fin.Phi[q]; Th[q] = fin.Th[q]; }
    const double mu = fin.mu;
    const int nc = fin.nc;
    // wave-uniform quantities in scalar registers
    const int wave_len = __builtin_amdgcn_readfirstlane(wave_len_v);
    const int m = __builtin_amdgcn_readfirstlane(m_v);
    const int mask = __builtin_amdgcn_readfirstlane(L.R) - 1;
    const int lim = live ? len : 0;
    const int nc_max = __builtin_amdgcn_readfirstlane(ar_wave_max(live ? nc : 0));
    const int len_min = -__builtin_amdgcn_readfirstlane(ar_wave_max(live ? -len : -0x3fffffff));
    // explicit address spaces: through the call boundary the pointers are generic, and generic (flat) loads would make
    // every LDS wait also wait for the HBM prefetch in flight
    typedef const __attribute__((address_space(1))) double *gptr_t;
    typedef __attribute__((address_space(3))) ar_ev_t *lptr_t;
    const gptr_t wg = (gptr_t)w;
    const lptr_t ring = (lptr_t)(L.base) + threadIdx.x;        // slot k of this lane: ring[k * NM_BLOCK]
    double css = 0.0;
    double wl[AR_MAXP] = {0, 0, 0, 0, 0}, ul[AR_MAXP] = {0, 0, 0, 0, 0};
    for (int k = 0; k <= mask; k++) ring[k * NM_BLOCK] = ar_ev_t{0.0, 0.0};
    // w is double-buffered in registers: the S rows of the NEXT iteration are requested before the S steps of the current
    // one run, and are only touched (copied into cur) after them, so the HBM latency sits behind S steps of recursion and
    // the loop has a single, already satisfied, wait per iteration.  The loads are unconditional: the block W has 2 S
    // spare rows, and steps past a lane's length contribute nothing.
    constexpr int S = 32;
    double cur[S], nxt[S];
    gptr_t wp_next = wg;
#pragma unroll
    for (int j = 0; j < S; j++) cur[j] = wp_next[(size_t)j * ld];
    wp_next += (size_t)S * ld;

    auto block = [&](const int base, auto gated_tag) __attribute__((always_inline)) {
        constexpr bool GATED = decltype(gated_tag)::value;
#pragma unroll
        for (int sb = 0; sb < S; sb += 4) {
            const int t0 = base + sb;
            ar_ev_t s1[4], s2[4];                 // slots t - m and t - 2m: {e, v}
            double vnew[4], enew[4];
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    s1[j] = ring[((t0 + j - m) & mask) * NM_BLOCK];
                    s2[j] = ring[((t0 + j - 2 * m) & mask) * NM_BLOCK];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int t = t0 + j;
                const double wp = cur[sb + j] - mu;
                double vt = wp;
#pragma unroll
                for (int q = 0; q < AR_MAXP; q++) vt = fma(phi[q], wl[q], vt);
#pragma unroll
                for (int q = AR_MAXP - 1; q > 0; q--) wl[q] = wl[q - 1];
                wl[0] = wp;
                double z = vt;
                if (MODE == 0) {
                    s1[j] = ring[((t - m) & mask) * NM_BLOCK];
                    s2[j] = ring[((t - 2 * m) & mask) * NM_BLOCK];
                }
                if (MODE != 2) {
                    z = fma(Phi[0], s1[j].y, z);
                    z = fma(Phi[1], s2[j].y, z);
                }
                // the newest lag enters last, so consecutive steps are one fused multiply-add apart
                double u = z;
#pragma unroll
                for (int q = AR_MAXP - 1; q >= 0; q--) u = fma(th[q], ul[q], u);
                const bool on = t >= nc;
                if (GATED) u = on ? u : 0.0;
#pragma unroll
                for (int q = AR_MAXP - 1; q > 0; q--) ul[q] = ul[q - 1];
                ul[0] = u;
                double et = u;
                if (MODE != 2) {
                    et = fma(Th[0], s1[j].x, et);
                    et = fma(Th[1], s2[j].x, et);
                    if (GATED) et = on ? et : 0.0;
                }
                if (MODE == 0) {
                    if (!GATED || t < lim) ring[(t & mask) * NM_BLOCK] = ar_ev_t{et, vt};
                }
                vnew[j] = vt; enew[j] = et;
                const double ec = (!GATED || t < lim) ? et : 0.0;
                css = fma(ec, ec, css);
            }
            if (MODE != 0) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (!GATED || t0 + j < lim) ring[((t0 + j) & mask) * NM_BLOCK] = ar_ev_t{enew[j], vnew[j]};
            }
        }
    };

    for (int base = 0; base < wave_len; base += S) {
#pragma unroll
        for (int j = 0; j < S; j++) nxt[j] = wp_next[(size_t)j * ld];
        wp_next += (size_t)S * ld;
        if (base >= nc_max && base + S <= len_min) block(base, std::false_type{});
        else block(base, std::true_type{});
#pragma unroll
        for (int j = 0; j < S; j++) cur[j] = nxt[j];
    }
    return css;
}

__device__ __forceinline__ double ar_css_pass(const double *w, size_t ld, int len, int wave_len, bool live, const ArFac &f, int m,
                                              const ArLds &L)
{
    if (m >= 4) return ar_css_pass_impl<1>(w, ld, len, wave_len, live, f, m, L);
    if (m <= 1) return ar_css_pass_impl<2>(w, ld, len, wave_len, live, f, m, L);
    return ar_css_pass_impl<0>(w, ld, len, wave_len, live, f, m, L);
}

// ------------------------------------------------------------------------------------------------
// prep: D (seasonal strength), d (KPSS), differenced block W, moments, integration constants
// ------------------------------------------------------------------------------------------------
__device__ bool ar_kpss_reject(const double *x, size_t ld, int n)
{
    if (n < 4) return false;
    double s = 0.0;
    for (int i = 0; i < n; i++) s = s + x[(size_t)i * ld];
    const double mean = s / (double)n;
    double cum = 0.0, eta = 0.0, s2 = 0.0;
    for (int i = 0; i < n; i++) {
        double e = x[(size_t)i * ld] - mean;
        cum = cum + e;
        eta = fma(cum, cum, eta);
        s2 = fma(e, e, s2);
    }
    const double dn = (double)n;
    eta = eta / (dn * dn);
    s2 = s2 / dn;
    const int lag = (int)(3.0 * sqrt(dn) / 13.0);
    for (int k = 1; k <= lag; k++) {
        double acc = 0.0;
        for (int t = k; t < n; t++) acc = fma(x[(size_t)t * ld] - mean, x[(size_t)(t - k) * ld] - mean, acc);
        double wgt = 1.0 - (double)k / ((double)lag + 1.0);
        s2 = s2 + 2.0 * wgt * (acc / dn);
    }
    if (!(s2 > 0.0)) return false;
    return (eta / s2) > 0.463;
}

__device__ double ar_seasonal_strength(const double *y, size_t ld, int n, int m, double *fig /* [m] lane-private scratch, stride NM_BLOCK */)
{
    if (m < 2 || n < 3 * m) return 0.0;
    const int half = m / 2;
    const int L = (m % 2 == 0) ? m + 1 : m;
    const double w = 1.0 / (double)m;
    const double wend = (m % 2 == 0) ? 0.5 / (double)m : w;
    double tot = 0.0;
    for (int j = 0; j < m; j++) {
        double sj = 0.0;
        int cnt = 0;
        for (int i = (j >= half ? j : j + m); i < n - half; i += m) {
            double acc = 0.0;
            for (int k = 0; k < L; k++) acc = acc + ((k == 0 || k == L - 1) ? wend : w) * y[(size_t)(i - half + k) * ld];
            sj = sj + (y[(size_t)i * ld] - acc);
            cnt++;
        }
        fig[j * NM_BLOCK] = sj / (double)cnt;
        tot = tot + fig[j * NM_BLOCK];
    }
    const double fmean = tot / (double)m;
    for (int j = 0; j < m; j++) fig[j * NM_BLOCK] = fig[j * NM_BLOCK] - fmean;
    const int nv = n - 2 * half;
    double sd = 0.0, sr = 0.0;
    int ph = half % m;
    for (int i = half; i < n - half; i++) {
        double acc = 0.0;
        for (int k = 0; k < L; k++) acc = acc + ((k == 0 || k == L - 1) ? wend : w) * y[(size_t)(i - half + k) * ld];
        double d = y[(size_t)i * ld] - acc;
        sd = sd + d;
        sr = sr + (d - fig[ph * NM_BLOCK]);
        ph = (ph + 1 == m) ? 0 : ph + 1;
    }
    const double md = sd / (double)nv, mr = sr / (double)nv;
    double vd = 0.0, vr = 0.0;
    ph = half % m;
    for (int i = half; i < n - half; i++) {
        double acc = 0.0;
        for (int k = 0; k < L; k++) acc = acc + ((k == 0 || k == L - 1) ? wend : w) * y[(size_t)(i - half + k) * ld];
        double d = y[(size_t)i * ld] - acc;
        double r = d - fig[ph * NM_BLOCK];
        vd = fma(d - md, d - md, vd);
        vr = fma(r - mr, r - mr, vr);
        ph = (ph + 1 == m) ? 0 : ph + 1;
    }
    if (!(vd > 0.0)) return 0.0;
    double f = 1.0 - vr / vd;
    if (f < 0.0) f = 0.0;
    if (f > 1.0) f = 1.0;
    return f;
}

__global__ __launch_bounds__(NM_BLOCK) void arima_prep_kernel(const ArimaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int s = blockIdx.x * NM_BLOCK + threadIdx.x;
    if (s >= a.n_series) return;
    const int n = a.len[s];
    if (n <= 0) { a.wlen[s] = 0; return; }
    const double *y = a.y + s;
    double *w = a.w + s;
    const size_t ld = a.ld;
    const int m = a.m;
    int len = n, D = 0, d = 0;
    if (m > 1 && ar_seasonal_strength(y, ld, n, m, lds + threadIdx.x) > 0.64 && n > m + 2) {
        D = 1;
        for (int t = m; t < n; t++) w[(size_t)(t - m) * ld] = y[(size_t)t * ld] - y[(size_t)(t - m) * ld];
        len = n - m;
    } else {
        for (int t = 0; t < n; t++) w[(size_t)t * ld] = y[(size_t)t * ld];
    }
    // integration constants of the seasonally differenced series (before the ordinary differences)
    a.last_d0[s] = w[(size_t)(len - 1) * ld];
    a.last_d1[s] = len >= 2 ? w[(size_t)(len - 1) * ld] - w[(size_t)(len - 2) * ld] : 0.0;
    while (d < 2 && len > 3 && ar_kpss_reject(w, ld, len)) {
        double prev = w[0];
        for (int t = 1; t < len; t++) {
            double cur = w[(size_t)t * ld];
            w[(size_t)(t - 1) * ld] = cur - prev;
            prev = cur;
        }
        len -= 1;
        d++;
    }
    double sum = 0.0;
    for (int i = 0; i < len; i++) sum = sum + w[(size_t)i * ld];
    const double wmean = sum / (double)len;
    double v = 0.0;
    for (int i = 0; i < len; i++) { double dd = w[(size_t)i * ld] - wmean; v = fma(dd, dd, v); }
    a.wlen[s] = len;
    a.d[s] = d;
    a.D[s] = D;
    a.wmean[s] = wmean;
    a.wsd[s] = sqrt(v / (double)len);
}

// ------------------------------------------------------------------------------------------------
// stepwise search
// ------------------------------------------------------------------------------------------------
enum { PH_NEXT = 0, PH_INIT, PH_ITER, PH_E, PH_OC, PH_IC, PH_SHRINK, PH_FINAL };

__device__ __forceinline__ double ar_trial(const ArLds &L, int D, int which, int i)
{
    double s = L.sim(0, i);
    for (int k = 1; k < D; k++) s = s + L.sim(k, i);
    const double xb = s / (double)D;
    const double xw = L.sim(D, i);
    const double a = which == 0 ? 2.0 : (which == 1 ? 3.0 : (which == 2 ? 1.5 : 0.5));
    const double b = which == 0 ? 1.0 : (which == 1 ? 2.0 : 0.5);
    return which == 3 ? a * xb + b * xw : a * xb - b * xw;
}

__device__ __forceinline__ void ar_accept(const ArLds &L, int D, int which, double fnew)
{
    double xn[AR_MAXDIM];
    for (int i = 0; i < D; i++) xn[i] = ar_trial(L, D, which, i);
    int j = D;
    while (j > 0 && fnew < L.fs(j - 1)) {
        L.fs(j) = L.fs(j - 1);
        for (int i = 0; i < D; i++) L.sim(j, i) = L.sim(j - 1, i);
        j--;
    }
    L.fs(j) = fnew;
    for (int i = 0; i < D; i++) L.sim(j, i) = xn[i];
}

__device__ __forceinline__ void ar_sort(const ArLds &L, int D)
{
    for (int k = 1; k <= D; k++) {
        const double fk = L.fs(k);
        double tmp[AR_MAXDIM];
        for (int i = 0; i < D; i++) tmp[i] = L.sim(k, i);
        int j = k;
        while (j > 0 && fk < L.fs(j - 1)) {
            L.fs(j) = L.fs(j - 1);
            for (int i = 0; i < D; i++) L.sim(j, i) = L.sim(j - 1, i);
            j--;
        }
        L.fs(j) = fk;
        for (int i = 0; i < D; i++) L.sim(j, i) = tmp[i];
    }
}

__global__ __launch_bounds__(NM_BLOCK) void arima_search_kernel(const ArimaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const int s = blockIdx.x * NM_BLOCK + lane;
    const bool valid = s < a.n_series;
    const int len = valid ? a.wlen[s] : 0;
    const bool live = valid && len >= 3;
    const int m = a.m;
    ArLds L{lds, AR_MAXP + AR_MAXSP * m + 1, ar_ring_slots(m)};
    const double *w = a.w + (valid ? s : 0);
    const size_t ld = a.ld;
    const int wave_len = ar_wave_max(live ? len : 0);
    if (wave_len == 0) {
        if (valid) { a.status[s] = FIT_SHORT; a.evals[s] = 0; a.passes[s] = 0; a.models[s] = 0; }
        return;
    }
    const int d = live ? a.d[s] : 0, Dd = live ? a.D[s] : 0;
    const double wmean = live ? a.wmean[s] : 0.0, wsd = live ? a.wsd[s] : 0.0;
    const int allow_c = (d + Dd <= 1) ? 1 : 0;
    const int maxP = m > 1 ? AR_MAXSP : 0;
    for (int wd = 0; wd < 21; wd++) L.tried(wd) = 0u;

    bool fin = !live;
    ArOrd best{0, 0, 0, 0, 0}, base{0, 0, 0, 0, 0}, cur{0, 0, 0, 0, 0};
    double best_aicc = __builtin_huge_val(), bestx[AR_MAXDIM] = {0, 0, 0, 0, 0, 0};
    bool have = false, improved = false;
    int stage = 0, idx = 0, n_models = 0, evals = 0, passes = 0;
    int ph = PH_NEXT, D = 0, vi = 0, nm_evals = 0, nm_iters = 1;
    double fxr = 0.0;

    for (;;) {
        // ---- 1. next candidate model of this lane (no pass needed) ----------------------------------
        while (!fin && ph == PH_NEXT) {
            ArOrd o;
            if (stage == 0) {
                if (idx >= 5) {
                    if (!have) { fin = true; break; }
                    stage = 1; idx = 0; base = best; improved = false;
                    continue;
                }
                const int sP = maxP ? 1 : 0;
                if (idx == 0) o = ArOrd{2, 2, sP, sP, allow_c};
                else if (idx == 1) o = ArOrd{0, 0, 0, 0, allow_c};
                else if (idx == 2) o = ArOrd{1, 0, sP, 0, allow_c};
                else if (idx == 3) o = ArOrd{0, 1, 0, sP, allow_c};
                else o = ArOrd{0, 0, 0, 0, allow_c ? 0 : -1};   // only when a constant is allowed
                idx++;
            } else {
                if (improved) { base = best; idx = 0; improved = false; }
                if (idx >= 17) { fin = true; break; }
                const int dPv[8] = {-1, 0, 1, 0, -1, -1, 1, 1}, dQv[8] = {0, -1, 0, 1, -1, 1, -1, 1};
                if (idx < 8) o = ArOrd{base.p, base.q, base.P + dPv[idx], base.Q + dQv[idx], base.c};
                else if (idx < 16) o = ArOrd{base.p + dPv[idx - 8], base.q + dQv[idx - 8], base.P, base.Q, base.c};
                else o = ArOrd{base.p, base.q, base.P, base.Q, 1 - base.c};
                idx++;
            }
            if (o.c < 0 || o.p < 0 || o.q < 0 || o.P < 0 || o.Q < 0 || o.p > AR_MAXP || o.q > AR_MAXP || o.P > maxP || o.Q > maxP ||
                o.p + o.q + o.P + o.Q > AR_MAXORDER || (o.c && !allow_c) || n_models >= AR_MAXMODELS)
                continue;
            const int key = ar_key(o);
            if (L.tried(key >> 5) & (1u << (key & 31))) continue;
            L.tried(key >> 5) |= (1u << (key & 31));
            n_models++;
            D = ar_dim(o);
            const int La0 = o.p + m * o.P;
            if (len - La0 <= 0 || len - (D + 1) - 1 <= 0) continue;        // fit impossible: candidate fails
            cur = o;
            for (int i = 0; i < D; i++) L.sim(0, i) = 0.0;
            if (o.c) L.sim(0, D - 1) = wmean;
            for (int k = 0; k < D; k++) {
                for (int i = 0; i < D; i++) L.sim(k + 1, i) = L.sim(0, i);
                const double step = (o.c && k == D - 1) ? (wsd > 0.0 ? 0.1 * wsd : 1.0e-4) : 0.25;
                L.sim(k + 1, k) = L.sim(0, k) + step;
            }
            nm_evals = 0; nm_iters = 1; vi = 0;
            if (D == 0) { ph = PH_FINAL; nm_evals = 1; }
            else ph = PH_INIT;
        }
        // ---- 2. trial point of the running fit ------------------------------------------------------
        double x[AR_MAXDIM] = {0, 0, 0, 0, 0, 0};
        if (!fin) {
            if (ph == PH_ITER) {
                bool stop = !(nm_evals < 200 * D && nm_iters < 200 * D);
                if (!stop) {
                    bool small = true;
                    for (int k = 1; k <= D; k++) {
                        for (int i = 0; i < D; i++)
                            if (!(fabs(L.sim(k, i) - L.sim(0, i)) <= 1.0e-4)) small = false;
                        if (!(fabs(L.fs(0) - L.fs(k)) <= 1.0e-8)) small = false;
                    }
                    stop = small;
                }
                if (stop) ph = PH_FINAL;
            }
            if (ph == PH_INIT) { for (int i = 0; i < D; i++) x[i] = L.sim(vi, i); }
            else if (ph == PH_ITER) { for (int i = 0; i < D; i++) x[i] = ar_trial(L, D, 0, i); }
            else if (ph == PH_E || ph == PH_OC || ph == PH_IC) {
                const int which = ph == PH_E ? 1 : (ph == PH_OC ? 2 : 3);
                for (int i = 0; i < D; i++) x[i] = ar_trial(L, D, which, i);
            } else if (ph == PH_SHRINK) { for (int i = 0; i < D; i++) x[i] = L.sim(1 + vi, i); }
            else { for (int i = 0; i < D; i++) x[i] = L.sim(0, i); }      // PH_FINAL
        }
        ArFac fac;
        ar_factors(cur, m, x, fac);
        if (__all(fin)) break;

        // ---- 3. one streamed pass -------------------------------------------------------------------
        const double css = ar_css_pass(w, ld, len, wave_len, !fin, fac, m, L);
        if (fin) continue;
        passes++;
        const int nu = len - fac.nc;
        double v = css / (double)nu;
        double f = __builtin_huge_val();
        if (fabs(css) <= 1.7976931348623157e308) {
            if (v < 1.0e-300) v = 1.0e-300;
            f = 0.5 * dm_log(v);
        }

        // ---- 4. consume -----------------------------------------------------------------------------
        if (ph == PH_INIT) {
            L.fs(vi) = f; vi++; nm_evals++;
            if (vi == D + 1) { ar_sort(L, D); ph = PH_ITER; }
        } else if (ph == PH_ITER) {
            fxr = f; nm_evals++;
            if (fxr < L.fs(0)) ph = PH_E;
            else if (fxr < L.fs(D - 1)) { ar_accept(L, D, 0, fxr); nm_iters++; }
            else if (fxr < L.fs(D)) ph = PH_OC;
            else ph = PH_IC;
        } else if (ph == PH_E) {
            nm_evals++;
            if (f < fxr) ar_accept(L, D, 1, f); else ar_accept(L, D, 0, fxr);
            nm_iters++; ph = PH_ITER;
        } else if (ph == PH_OC || ph == PH_IC) {
            nm_evals++;
            const bool ok = (ph == PH_OC) ? (f <= fxr) : (f < L.fs(D));
            if (ok) { ar_accept(L, D, ph == PH_OC ? 2 : 3, f); nm_iters++; ph = PH_ITER; }
            else {
                for (int k = 1; k <= D; k++)
                    for (int i = 0; i < D; i++) L.sim(k, i) = L.sim(0, i) + 0.5 * (L.sim(k, i) - L.sim(0, i));
                vi = 0; ph = PH_SHRINK;
            }
        } else if (ph == PH_SHRINK) {
            L.fs(1 + vi) = f; vi++; nm_evals++;
            if (vi == D) { nm_iters++; ar_sort(L, D); ph = PH_ITER; }
        } else { // PH_FINAL: information criterion of the fitted candidate
            evals += nm_evals;
            if (fabs(css) <= 1.7976931348623157e308) {
                const double dn = (double)len, dk = (double)(D + 1);
                const double aicc = dn * dm_log(v) + 2.0 * dk + 2.0 * dk * (dk + 1.0) / (dn - dk - 1.0);
                if (fabs(aicc) <= 1.7976931348623157e308 && aicc < best_aicc) {
                    best_aicc = aicc; best = cur; have = true; improved = true;
                    for (int i = 0; i < AR_MAXDIM; i++) bestx[i] = i < D ? L.sim(0, i) : 0.0;
                }
            }
            ph = PH_NEXT;
        }
    }
    if (valid) {
        a.status[s] = (live && have) ? FIT_OK : FIT_SHORT;
        a.aicc[s] = best_aicc;
        a.order[(size_t)0 * ld + s] = best.p; a.order[(size_t)1 * ld + s] = best.q; a.order[(size_t)2 * ld + s] = best.P;
        a.order[(size_t)3 * ld + s] = best.Q; a.order[(size_t)4 * ld + s] = best.c;
        for (int i = 0; i < AR_MAXDIM; i++) a.xbest[(size_t)i * ld + s] = bestx[i];
        a.evals[s] = evals; a.passes[s] = passes; a.models[s] = n_models;
    }
}

// ------------------------------------------------------------------------------------------------
// forecast of the differenced series with the selected model, then integration
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NM_BLOCK) void arima_forecast_kernel(const ArimaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const int s = blockIdx.x * NM_BLOCK + lane;
    const bool valid = s < a.n_series;
    const int len = valid ? a.wlen[s] : 0;
    const bool live = valid && len >= 3 && a.status[s] == FIT_OK;
    const int m = a.m;
    ArLds L{lds, AR_MAXP + AR_MAXSP * m + 1, ar_ring_slots(m)};
    const double *w = a.w + (valid ? s : 0);
    const size_t ld = a.ld;
    const int wave_len = ar_wave_max(live ? len : 0);
    if (wave_len == 0) return;
    ArOrd o{0, 0, 0, 0, 0};
    double x[AR_MAXDIM] = {0, 0, 0, 0, 0, 0};
    if (live) {
        o.p = a.order[(size_t)0 * ld + s]; o.q = a.order[(size_t)1 * ld + s]; o.P = a.order[(size_t)2 * ld + s];
        o.Q = a.order[(size_t)3 * ld + s]; o.c = a.order[(size_t)4 * ld + s];
        for (int i = 0; i < AR_MAXDIM; i++) x[i] = a.xbest[(size_t)i * ld + s];
    }
    int La = 0, Lb = 0;
    double mu = 0.0;
    ArFac fac;
    ar_factors(o, m, x, fac);
    if (live) ar_build_poly(o, m, x, L, La, Lb, mu);
    (void)ar_css_pass(w, ld, len, wave_len, live, fac, m, L);
    if (!live) return;
    const int n = a.len[s], h = a.h;
    const int d = a.d[s], Dd = a.D[s];
    double *out = a.yhat + (size_t)s * h;
    for (int j = 0; j < h; j++) {
        const int t = len + j;
        double acc = mu;
        for (int k = 1; k <= La; k++)
            if (t - k >= 0) {
                const double wv = (t - k < len) ? w[(size_t)(t - k) * ld] : out[t - k - len];
                acc = fma(L.a(k), wv - mu, acc);
            }
        for (int k = 1; k <= Lb; k++)
            if (t - k >= 0 && t - k < len) acc = fma(L.b(k), L.e_at(t - k), acc);
        out[j] = acc;
    }
    double last_d0 = a.last_d0[s], last_d1 = a.last_d1[s];
    const double *y = a.y + s;
    for (int j = 0; j < h; j++) {
        double val = out[j];
        if (d == 2) { last_d1 = last_d1 + val; val = last_d1; }
        if (d >= 1) { last_d0 = last_d0 + val; val = last_d0; }
        if (Dd) val = val + ((j - m < 0) ? y[(size_t)(n + j - m) * ld] : out[j - m]);
        out[j] = val;
    }
    a.model_code[s] = 1000000 + o.p * 100000 + d * 10000 + o.q * 1000 + o.P * 100 + Dd * 10 + o.Q;
}

void launch_arima(const ArimaArgs &a, hipStream_t stream)
{
    const int grid = (a.n_series + NM_BLOCK - 1) / NM_BLOCK;
    const size_t lds_bytes = ar_lds_bytes(a.m);
    if (lds_bytes > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)arima_search_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute((const void *)arima_forecast_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    }
    const size_t prep_lds = sizeof(double) * (size_t)(a.m > 1 ? a.m : 1) * NM_BLOCK;
    hipLaunchKernelGGL(arima_prep_kernel, dim3(grid), dim3(NM_BLOCK), prep_lds, stream, a);
    hipLaunchKernelGGL(arima_search_kernel, dim3(grid), dim3(NM_BLOCK), lds_bytes, stream, a);
    hipLaunchKernelGGL(arima_forecast_kernel, dim3(grid), dim3(NM_BLOCK), lds_bytes, stream, a);
}

} // namespace anofox
