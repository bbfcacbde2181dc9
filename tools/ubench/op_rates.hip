// Micro-benchmark: issue cost of the fp64 instructions of an IEEE division and of the conversions, relative to v_fma_f64,
// with 4 independent chains per wave and 2 waves per SIMD (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o op_rates op_rates.hip && ./op_rates
#include <hip/hip_runtime.h>
#include <cstdio>
enum { OP_FMA, OP_RCP, OP_DIV, OP_RNDNE, OP_CVT, OP_SQRT, OP_MUL, OP_FIXUP, OP_FMAS, OP_SCALE, N_OPS };
template <int OP>
__device__ __forceinline__ double op(double x, double a, double b)
{
    if (OP == OP_FMA) return __builtin_fma(x, a, b);
    if (OP == OP_RCP) return __builtin_amdgcn_rcp(x) + b;                       // v_rcp_f64 + v_add_f64
    if (OP == OP_DIV) return a / x + b;                                          // full IEEE sequence + add
    if (OP == OP_RNDNE) return __builtin_rint(x * a) + b;                        // v_mul + v_rndne + v_add
    if (OP == OP_CVT) return (double)(int)(x * a) + b;                           // mul + cvt_i32_f64 + cvt_f64_i32 + add
    if (OP == OP_SQRT) return __builtin_amdgcn_sqrt(x) + b;
    if (OP == OP_MUL) return x * a + b * a;                                      // 2 mul + add (contract off)
    if (OP == OP_FIXUP) return __builtin_amdgcn_div_fixup(x, a, b);
    if (OP == OP_FMAS) return __builtin_amdgcn_div_fmas(x, a, b, true);
    if (OP == OP_SCALE) return __builtin_amdgcn_div_scale(x, a, true, nullptr) + b;
    return x;
}
template <int OP>
__global__ void k(double *out, int iters, double a, double b)
{
    double x[4];
#pragma unroll
    for (int c = 0; c < 4; c++) x[c] = 1.0 + threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < 4; c++) x[c] = op<OP>(x[c], a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}
template <int OP>
double run(const char *name, double fma_ns)
{
    int dev; (void)hipGetDevice(&dev); hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, dev);
    const int grid = p.multiProcessorCount * 4 * 2;
    double *d; (void)hipMalloc(&d, sizeof(double) * grid * 64);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<grid, 64>>>(d, 100, 0.999, 1e-3);
    (void)hipEventRecord(e0);
    k<OP><<<grid, 64>>>(d, iters, 0.999, 1e-3);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / ((double)iters * 32) / 2;        // ns per op-group per wave-slot (2 waves share a SIMD)
    printf("%-28s %8.2f ms  %6.2f ns per group  = %5.2f x fma\n", name, ms, per, fma_ns > 0 ? per / fma_ns : 1.0);
    (void)hipFree(d);
    return per;
}
int main()
{
    const double f = run<OP_FMA>("fma", 0);
    run<OP_MUL>("2 mul + add", f);
    run<OP_RCP>("rcp + add", f);
    run<OP_SQRT>("sqrt + add", f);
    run<OP_DIV>("IEEE a/x + add", f);
    run<OP_RNDNE>("mul + rndne + add", f);
    run<OP_CVT>("mul + cvt_i32 + cvt_f64 + add", f);
    run<OP_FIXUP>("div_fixup", f);
    run<OP_FMAS>("div_fmas", f);
    run<OP_SCALE>("div_scale + add", f);
    return 0;
}
