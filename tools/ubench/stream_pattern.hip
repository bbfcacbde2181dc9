// Micro-benchmark (round 5): what HBM bandwidth does the round kernels' access pattern reach, and what would a time-interleaved block reach?
// One-wave workgroups, lane <-> column, every wave streams ITS 64 columns of a [T x ld] fp64 block for `passes` passes with S rows in
// flight (the double-buffered block loop of ets_device.hpp) and `chain` dependent FMAs per time step (the recursion).
//   layout 1: time-major Y[t * ld + s]                -- 8 B per lane, 512 contiguous bytes per wave-load, rows ld * 8 bytes apart
//   layout 2: Y[(t / 2) * ld * 2 + s * 2 + t % 2]      -- 16 B per lane, 1 KB contiguous per wave-load
//   layout 4: Y[(t / 4) * ld * 4 + s * 4 + t % 4]      -- 32 B per lane (two b128 loads), 2 KB contiguous per wave per four steps
// hipcc --offload-arch=gfx950 -O3 -o stream_pattern stream_pattern.hip && ./stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int L, int S>
__global__ __launch_bounds__(64, 2) void k(const double *y, size_t ld, int T, int passes, int chain, double a, double *out)
{
    const int lane = threadIdx.x;
    const size_t col = (size_t)blockIdx.x * 64 + lane;
    double acc = lane * 1e-3;
    for (int p = 0; p < passes; p++) {
        double cur[S], nxt[S];
        auto load = [&](double (&buf)[S], int t0) __attribute__((always_inline)) {
            if (t0 + S > T) t0 = T - S;
#pragma unroll
            for (int j = 0; j < S; j += L) {
                const double *src = y + ((size_t)((t0 + j) / L) * ld + col) * L;
#pragma unroll
                for (int u = 0; u < L; u++) buf[j + u] = src[u];
            }
        };
        load(cur, 0);
        for (int t = 0; t < T; t += S) {
            load(nxt, t + S);
#pragma unroll
            for (int j = 0; j < S; j++) {
                double v = cur[j];
                for (int c = 0; c < chain; c++) acc = __builtin_fma(acc, a, v);
            }
#pragma unroll
            for (int j = 0; j < S; j++) cur[j] = nxt[j];
        }
    }
    out[col] = acc;
}
template <int L, int S>
void run(int waves, int T, int passes, int chain)
{
    const size_t ld = (size_t)waves * 64;
    double *y, *out;
    hipMalloc(&y, sizeof(double) * ld * T); hipMalloc(&out, sizeof(double) * ld);
    hipMemset(y, 0, sizeof(double) * ld * T);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<L, S><<<waves, 64>>>(y, ld, T, 1, chain, 0.999, out);
    hipEventRecord(e0);
    k<L, S><<<waves, 64>>>(y, ld, T, passes, chain, 0.999, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)ld * T * 8.0 * passes;
    printf("layout %d  S %2d  waves %5d  T %d  chain %2d: %8.2f ms  %6.2f TB/s  (block %.0f MB, %.0f cycles per step per wave at 2.4 GHz)\n", L, S, waves, T, chain, ms,
           bytes / (ms * 1e-3) / 1e12, (double)ld * T * 8 / 1e6, ms * 1e-3 * 2.4e9 / ((double)T * passes));
    hipFree(y); hipFree(out);
}
int main(int argc, char **argv)
{
    const int T = 1024, passes = 40;
    for (int waves : {1024, 2048, 4096}) {
        for (int chain : {4, 16, 32}) {
            run<1, 8>(waves, T, passes, chain);
            run<1, 16>(waves, T, passes, chain);
            run<1, 32>(waves, T, passes, chain);
            run<2, 16>(waves, T, passes, chain);
            run<4, 16>(waves, T, passes, chain);
            run<4, 32>(waves, T, passes, chain);
        }
    }
    return 0;
}
