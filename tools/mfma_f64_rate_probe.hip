// Micro-probe: v_mfma_f64_16x16x4_f64 on gfx950 -- cycles per MFMA for a dependent chain and for four independent
// accumulators (one wave per SIMD).  hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_f64 tools/mfma_f64_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double doublex4 __attribute__((ext_vector_type(4)));
// inline asm with the accumulator tied in place: the builtin made the compiler copy every accumulator through the
// other register file once per loop trip, which is what the first version of this probe measured
#define MF(a, b, c) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
template <int MODE> __global__ __launch_bounds__(256) void k(const double *in, doublex4 *out, unsigned long long *t, int iters)
{
    double a0 = in[threadIdx.x], b0 = in[threadIdx.x + 256];
    doublex4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            MF(a0, b0, c0); MF(a0, b0, c0); MF(a0, b0, c0); MF(a0, b0, c0);
        } else {
            MF(a0, b0, c0); MF(a0, b0, c1); MF(a0, b0, c2); MF(a0, b0, c3);
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    unsigned long long t1 = clock64();
    out[threadIdx.x] = c0 + c1 + c2 + c3;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
int main()
{
    double *in; doublex4 *out; unsigned long long *t, h;
    (void)hipMalloc(&in, 1024 * 8); (void)hipMemset(in, 0, 1024 * 8); (void)hipMalloc(&out, 512 * 32); (void)hipMalloc(&t, 8);
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, in, out, t, iters);
            else hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, in, out, t, iters);
        }
        (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        printf("%s: %.1f cycles per v_mfma_f64_16x16x4_f64 (clock64 ticks)\n", mode ? "four independent accumulators" : "one dependent chain", (double)h / (iters * 4.0));
    }
    return 0;
}
