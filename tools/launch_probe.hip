// launch_probe.hip -- what an (almost) empty kernel costs per launch when replayed from a hipGraph as a dependent
// chain, for grids of the shapes the C3 evaluation could take: the "launch" row of profiles/r03_ablation_c3.txt.
//   hipcc --offload-arch=gfx950 -O2 -o tools/launch_probe tools/launch_probe.hip && tools/launch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int NT> __global__ __launch_bounds__(NT) void k_empty(double *p, int lds_touch)
{
    extern __shared__ double sm[];
    if (lds_touch && threadIdx.x == 0) sm[0] = 1.0;
    if (p && blockIdx.x == 0xffffff) p[0] = sm[0];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int NT> static int run(int grid, size_t lds, const char *tag)
{
    hipStream_t s; CK(hipStreamCreate(&s));
    hipGraph_t g; hipGraphExec_t ge;
    const int chain = 250;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(k_empty<NT>, dim3(grid), dim3(NT), lds, s, (double *)nullptr, 1);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int w = 0; w < 8; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-34s grid %5d x %4d threads, %6zu B LDS: %.3f us per launch\n", tag, grid, NT, lds, best * 1e3 / (8 * chain));
    return 0;
}
int main()
{
    CK(hipFuncSetAttribute((const void *)k_empty<768>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_empty<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    run<256>(768, 41000, "C3 today (3 workgroups per CU)");
    run<256>(768, 0, "same, no LDS");
    run<768>(256, 123000, "one 12-wave workgroup per CU");
    run<1024>(256, 123000, "one 16-wave workgroup per CU");
    run<512>(512, 61000, "two 8-wave workgroups per CU");
    run<64>(3072, 10000, "one wave per workgroup");
    run<256>(256, 41000, "256 x 256");
    run<256>(1, 0, "a single workgroup");
    return 0;
}
