// Micro-benchmark (round 5): cadence of DEPENDENT fp64 FMAs whose three operands are all vector registers (a Horner chain with per-lane
// coefficients: the b^phi series of ets_device.hpp), against the same chain with scalar coefficients, and Estrin's scheme of the same
// polynomial; one and two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o fma_vgpr fma_vgpr.hip && ./fma_vgpr
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64, 2) void k(double *out, const double *cin, int iters, double sa)
{
    double c[12];
    for (int i = 0; i < 12; i++) c[i] = MODE == 1 ? sa + i : cin[threadIdx.x * 12 + i];     // MODE 1: uniform (scalar) coefficients
    double x = threadIdx.x * 1e-6;
    for (int it = 0; it < iters; it++) {
        if (MODE <= 1) {                // Horner, 12 dependent FMAs
            double p = c[11];
#pragma unroll
            for (int j = 10; j >= 0; j--) p = __builtin_fma(p, x, c[j]);
            x = __builtin_fma(p, x, 1e-9);
        } else {                        // Estrin, depth 5
            const double r = x, r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
            const double p0 = __builtin_fma(c[1], r, c[0]), p1 = __builtin_fma(c[3], r, c[2]), p2 = __builtin_fma(c[5], r, c[4]);
            const double p3 = __builtin_fma(c[7], r, c[6]), p4 = __builtin_fma(c[9], r, c[8]), p5 = __builtin_fma(c[11], r, c[10]);
            const double q0 = __builtin_fma(p1, r2, p0), q1 = __builtin_fma(p3, r2, p2), q2 = __builtin_fma(p5, r2, p4);
            const double s0 = __builtin_fma(q1, r4, q0);
            x = __builtin_fma(q2, r8, s0);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
template <int MODE> void run(int wps, const char *what, int ops)
{
    const int grid = 256 * 4 * wps, iters = 200000;
    double *d, *c; hipMalloc(&d, 8 * grid * 64); hipMalloc(&c, 8 * 64 * 12); hipMemset(c, 0, 8 * 64 * 12);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 64>>>(d, c, 10, 1e-3);
    hipEventRecord(e0); k<MODE><<<grid, 64>>>(d, c, iters, 1e-3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/SIMD %d: %7.1f cycles per evaluation per wave (%d instructions)\n", what, wps, ms * 1e-3 * 2.4e9 / iters, ops);
    hipFree(d); hipFree(c);
}
int main()
{
    for (int w : {1, 2}) {
        run<0>(w, "Horner, vector coefficients", 12);
        run<1>(w, "Horner, scalar coefficients", 12);
        run<2>(w, "Estrin, vector coefficients", 14);
    }
}
