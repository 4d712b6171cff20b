// microbenchmark: back-to-back v_mfma_f64_16x16x4_f64, 4 independent accumulators per wave
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double *out, int iters)
{
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = blockIdx.x * 1e-3 + 1.0;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
    }
    d4 s = a0 + a1 + a2 + a3;
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main()
{
    double *out; hipMalloc(&out, sizeof(double) * 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024, 2048}) {
        const int iters = 4096;
        k<<<wgs, 256>>>(out, 16);
        hipEventRecord(e0);
        k<<<wgs, 256>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)wgs * 4 * iters * 4 * 2048.0;
        printf("wgs=%d  %.3f ms  %.1f TFLOP/s\n", wgs, ms, flops / ms / 1e9);
    }
    return 0;
}
