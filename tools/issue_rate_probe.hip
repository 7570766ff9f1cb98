// Micro-probe: how many cycles does one wave per SIMD spend per instruction, by type, alone and in the
// shadow of v_mfma_f32_16x16x32_f16 (16 cycles of matrix pipe each)?  Decides what the scan's step may contain.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
// MODE: 0 = 16 MFMA; 1 = 16 x (MFMA + NS salu); 2 = 16 x (MFMA + NS valu); 3 = 64 salu; 4 = 64 valu (independent);
//       5 = 64 valu dependent chain; 6 = 16 x (MFMA + NS ds_read_b128); 7 = 16 x (MFMA + 2 salu + 2 valu)
template <int MODE, int NS>
__global__ __launch_bounds__(256) void k(const half8 *in, floatx4 *out, unsigned long long *t, int iters)
{
    __shared__ floatx4 sm[1024];
    sm[threadIdx.x] = floatx4{1, 2, 3, 4};
    __syncthreads();
    half8 a0 = in[threadIdx.x], b0 = in[threadIdx.x + 512];
    asm volatile("" : "+a"(a0));
    floatx4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    int s0 = iters, s1 = 1, s2 = 2, s3 = 3;
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    floatx4 r0, r1, r2, r3;
    unsigned la = threadIdx.x * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#define MF4 asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "a"(a0), "v"(b0));
#define M(c) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "a"(a0), "v"(b0));
#define S asm volatile("s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %2, %2, 1\n s_add_i32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
#define S1(x) asm volatile("s_add_i32 %0, %0, 1" : "+s"(x) : : "scc");
#define V1(x) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));
#define L1(x) asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(la));
        if (MODE == 0) { REP4(M(c0) M(c1) M(c2) M(c3)) }
        if (MODE == 1) { REP4(M(c0) if (NS > 0) S1(s0) if (NS > 1) S1(s1) if (NS > 2) S1(s2) if (NS > 3) S1(s3) if (NS > 4) S1(s0) if (NS > 5) S1(s1)
                              M(c1) if (NS > 0) S1(s0) if (NS > 1) S1(s1) if (NS > 2) S1(s2) if (NS > 3) S1(s3) if (NS > 4) S1(s0) if (NS > 5) S1(s1)
                              M(c2) if (NS > 0) S1(s0) if (NS > 1) S1(s1) if (NS > 2) S1(s2) if (NS > 3) S1(s3) if (NS > 4) S1(s0) if (NS > 5) S1(s1)
                              M(c3) if (NS > 0) S1(s0) if (NS > 1) S1(s1) if (NS > 2) S1(s2) if (NS > 3) S1(s3) if (NS > 4) S1(s0) if (NS > 5) S1(s1)) }
        if (MODE == 2) { REP4(M(c0) if (NS > 0) V1(v0) if (NS > 1) V1(v1) if (NS > 2) V1(v2) if (NS > 3) V1(v3) if (NS > 4) V1(v0) if (NS > 5) V1(v1)
                              M(c1) if (NS > 0) V1(v0) if (NS > 1) V1(v1) if (NS > 2) V1(v2) if (NS > 3) V1(v3) if (NS > 4) V1(v0) if (NS > 5) V1(v1)
                              M(c2) if (NS > 0) V1(v0) if (NS > 1) V1(v1) if (NS > 2) V1(v2) if (NS > 3) V1(v3) if (NS > 4) V1(v0) if (NS > 5) V1(v1)
                              M(c3) if (NS > 0) V1(v0) if (NS > 1) V1(v1) if (NS > 2) V1(v2) if (NS > 3) V1(v3) if (NS > 4) V1(v0) if (NS > 5) V1(v1)) }
        if (MODE == 3) { REP16(S1(s0) S1(s1) S1(s2) S1(s3)) }
        if (MODE == 4) { REP16(V1(v0) V1(v1) V1(v2) V1(v3)) }
        if (MODE == 5) { REP16(V1(v0) V1(v0) V1(v0) V1(v0)) }
        if (MODE == 6) { REP4(M(c0) if (NS > 0) L1(r0) if (NS > 1) L1(r1) M(c1) if (NS > 0) L1(r2) if (NS > 1) L1(r3)
                              M(c2) if (NS > 0) L1(r0) if (NS > 1) L1(r1) M(c3) if (NS > 0) L1(r2) if (NS > 1) L1(r3))
                         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (MODE == 7) { REP4(M(c0) S1(s0) V1(v0) S1(s1) V1(v1) M(c1) S1(s2) V1(v2) S1(s3) V1(v3)
                              M(c2) S1(s0) V1(v0) S1(s1) V1(v1) M(c3) S1(s2) V1(v2) S1(s3) V1(v3)) }
        if (MODE == 8) { REP16(S1(s0) S1(s0) S1(s0) S1(s0)) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = c0 + c1 + c2 + c3 + floatx4{v0 + v1, v2 + v3, (float)(s0 + s1), (float)(s2 + s3)};
    if (MODE == 6) out[threadIdx.x + 256] = r0 + r1 + r2 + r3;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
template <int MODE, int NS> void run(const char *name, half8 *in, floatx4 *out, unsigned long long *t, double per)
{
    int iters = 2000;
    unsigned long long h;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE, NS>), dim3(1), dim3(256), 0, 0, in, out, t, iters);
    (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-46s cycles/group = %.2f\n", name, (double)h / (iters * per));
    fflush(stdout);
}
int main()
{
    half8 *in; floatx4 *out; unsigned long long *t;
    (void)hipMalloc(&in, 1024 * 16); (void)hipMemset(in, 0x3c, 1024 * 16); (void)hipMalloc(&out, 1024 * 16); (void)hipMalloc(&t, 8);
    run<0, 0>("MFMA only (per MFMA)", in, out, t, 16);
    run<1, 1>("MFMA + 1 salu (per MFMA)", in, out, t, 16);
    run<1, 2>("MFMA + 2 salu", in, out, t, 16);
    run<1, 3>("MFMA + 3 salu", in, out, t, 16);
    run<1, 4>("MFMA + 4 salu", in, out, t, 16);
    run<1, 6>("MFMA + 6 salu", in, out, t, 16);
    run<2, 1>("MFMA + 1 valu", in, out, t, 16);
    run<2, 2>("MFMA + 2 valu", in, out, t, 16);
    run<2, 3>("MFMA + 3 valu", in, out, t, 16);
    run<2, 4>("MFMA + 4 valu", in, out, t, 16);
    run<2, 6>("MFMA + 6 valu", in, out, t, 16);
    run<7, 0>("MFMA + 2 salu + 2 valu", in, out, t, 16);
    run<6, 1>("MFMA + 1 ds_read_b128", in, out, t, 16);
    run<6, 2>("MFMA + 2 ds_read_b128", in, out, t, 16);
    run<3, 0>("salu independent (per instr)", in, out, t, 64);
    run<8, 0>("salu dependent chain (per instr)", in, out, t, 64);
    run<4, 0>("valu independent (per instr)", in, out, t, 64);
    run<5, 0>("valu dependent chain (per instr)", in, out, t, 64);
    return 0;
}
