// Micro-benchmark: issue cadence of dependent vs independent fp64 FMAs on one wave per SIMD (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o fma_latency fma_latency.hip && ./fma_latency
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k(double *out, int iters, double a, double b)
{
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) x[c] = __builtin_fma(x[c], a, b);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
void run(int waves_per_simd, int iters = 20000)
{
    int dev; hipGetDevice(&dev); hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    const int grid = p.multiProcessorCount * 4 * waves_per_simd;   // one-wave workgroups
    double *d; hipMalloc(&d, sizeof(double) * grid * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS><<<grid, 64>>>(d, 100, 0.999, 1e-3);
    hipEventRecord(e0);
    k<CHAINS><<<grid, 64>>>(d, iters, 0.999, 1e-3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_inst = (double)iters * 16 * CHAINS;             // per wave
    const double clk = p.clockRate * 1e3;                          // Hz
    printf("chains %d, waves/SIMD %d: %.2f ms, %.2f cycles per FMA per wave (at %.0f MHz), %.1f TFLOP/s\n", CHAINS, waves_per_simd, ms,
           ms * 1e-3 * clk / n_inst, clk / 1e6, 2.0 * n_inst * 64 * grid / (ms * 1e-3) / 1e12);
    hipFree(d);
}
int main()
{
    run<1>(1); run<2>(1); run<4>(1); run<8>(1);
    run<1>(2); run<1>(4); run<4>(2);
    run<4>(2, 2000000);      // ~0.5 s of sustained fp64 FMA issue: does the clock hold?
    run<4>(2, 8000000);      // ~2 s
    return 0;
}
