// cycles per MFMA of one wave on one SIMD: v_mfma_f32_16x16x16_f16 vs v_mfma_f32_16x16x32_f16, independent accumulators
// (NACC of them, round robin) and one dependent chain.   hipcc --offload-arch=gfx950 -O3 -o mfma_issue_probe mfma_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half4_ __attribute__((ext_vector_type(4)));
typedef _Float16 half8_ __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int K32, int NACC> __global__ void k(float *out, long long *cyc, int iters, float seed)
{
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{seed, 0, 0, 0};
    half4_ a4 = {(_Float16)seed, (_Float16)1, (_Float16)2, (_Float16)3}, b4 = a4;
    half8_ a8 = {(_Float16)seed, 1, 2, 3, 4, 5, 6, 7}, b8 = a8;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (K32) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int K32, int NACC> void run(const char *name)
{
    float *out; long long *cyc, h;
    hipMalloc(&out, 256); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<K32, NACC>), dim3(1), dim3(64), 0, 0, out, cyc, iters, 0.001f);
    hipLaunchKernelGGL((k<K32, NACC>), dim3(1), dim3(64), 0, 0, out, cyc, iters, 0.001f);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s, %d accumulators: %.1f memtime ticks per MFMA (s_memtime runs at 100 MHz: x24 for 2.4 GHz clocks)\n", name, NACC, (double)h / iters / NACC);
}
int main()
{
    run<0, 1>("16x16x16 f16"); run<0, 4>("16x16x16 f16"); run<0, 12>("16x16x16 f16");
    run<1, 1>("16x16x32 f16"); run<1, 4>("16x16x32 f16"); run<1, 12>("16x16x32 f16");
    return 0;
}
