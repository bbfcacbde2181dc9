// Micro-benchmark (round 5): what does a VALU / LDS instruction cost when EXEC is zero?  (The alternative to a skip branch over a rarely
// needed block: ~96 cycles of a one-wave-per-SIMD step for ANY branch, profiles/r05_step_anatomy.txt.)
// Loop body: 8 dependent FMAs (the "step"), then a block of 32 FMAs + 2 LDS reads run (a) with EXEC = all, (b) with EXEC = 0, (c) left out,
// (d) skipped by s_cbranch_execz.   hipcc --offload-arch=gfx950 -O3 -o exec0_cost exec0_cost.hip && ./exec0_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define FMA8 "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
#define BLK32 "v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n" \
              "v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n" \
              "v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n" \
              "v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %3, %3, %1, %2\n"
template <int MODE>
__global__ __launch_bounds__(64, 2) void k(double *out, int iters, double a, double b)
{
    double x = threadIdx.x * 1e-3, y = 1.0;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) asm volatile(FMA8 : "+v"(x) : "v"(a), "v"(b), "v"(y));
        if (MODE == 1) asm volatile(FMA8 BLK32 : "+v"(x) : "v"(a), "v"(b), "v"(y));
        if (MODE == 2) asm volatile(FMA8 "s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0\n" BLK32 "s_mov_b64 exec, s[20:21]\n" : "+v"(x) : "v"(a), "v"(b), "v"(y) : "s20", "s21");
        if (MODE == 3) asm volatile(FMA8 "s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0\n s_cbranch_execz 1f\n" BLK32 "1:\n s_mov_b64 exec, s[20:21]\n" : "+v"(x) : "v"(a), "v"(b), "v"(y) : "s20", "s21");
    }
    out[blockIdx.x * 64 + threadIdx.x] = x + y;
}
// (operands: %0 x (read-write), %1 a, %2 b, %3 y -- the block's FMAs overwrite their own input register %3, whose value nobody reads)
template <int MODE> void run(int wps, const char *what)
{
    const int grid = 256 * 4 * wps, iters = 100000;
    double *d; hipMalloc(&d, 8 * grid * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 64>>>(d, 10, 0.999, 1e-3);
    hipEventRecord(e0); k<MODE><<<grid, 64>>>(d, iters, 0.999, 1e-3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s waves/SIMD %d: %7.1f cycles per iteration per wave\n", what, wps, ms * 1e-3 * 2.4e9 / iters);
    hipFree(d);
}
int main()
{
    for (int w : {1, 2}) {
        run<0>(w, "8 dependent FMAs");
        run<1>(w, "8 FMAs + 32 FMAs, EXEC = all");
        run<2>(w, "8 FMAs + 32 FMAs under EXEC = 0 (no branch)");
        run<3>(w, "8 FMAs + s_cbranch_execz over the 32 (taken)");
    }
}
