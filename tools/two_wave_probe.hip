// Micro-probe: does a "helper" wave on the same SIMD take the non-matrix work of a scan step off the chain wave?
// A step of the MFMA scan (one row tile, C = 64, 3x3) is modelled as
//   main   : barrier; 18 leading MFMAs carrying 13 ds_read_b128; wait; 12 chain MFMAs on the data read;
//            18 trailing MFMAs carrying 36 VALU (the epilogue) ; 2 ds_write_b64; wait
//   helper : barrier; NS SALU + NV VALU (row-operation addresses), 5 ds_read_b128, NM MFMAs (the z product),
//            1 global store, 1 LDS-DMA
// MODE 0: main alone (4 waves); 1: + helper that only takes the barrier; 2: + helper without MFMAs; 3: + full helper;
//      4: one wave does both (the round-1 structure); 5: as 3, main at s_setprio 3; 6: as 3, roles swapped (helper = waves 0-3)
//      7: as 3 with the helper at s_setprio 0 and main at 1
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define MFMA(c, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define DSR(r, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(r) : "v"(addr))
#define VALU(x, y) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x) : "v"(y))
#define SALU(x) asm volatile("s_add_i32 %0, %0, 1" : "+s"(x) : : "scc")

struct MainState {
    half8 a0, a1, f2a, f2b;
    floatx4 l0, l1, l2, c0, c1;
    half8 r[13];
    float e0, e1, e2, e3;
};

__device__ __forceinline__ void main_step(MainState &s, unsigned la, unsigned wa)
{
    asm volatile("s_barrier" ::: "memory");
    // leading 18 MFMAs (operands in registers), one LDS request behind each of the first 13
#define LEAD(i, acc) MFMA(acc, s.a0, (i & 1) ? s.f2a : s.f2b); if (i < 13) DSR(s.r[i], la, 0);
    LEAD(0, s.l0) LEAD(1, s.l1) LEAD(2, s.l2) LEAD(3, s.l0) LEAD(4, s.l1) LEAD(5, s.l2) LEAD(6, s.l0) LEAD(7, s.l1) LEAD(8, s.l2)
    LEAD(9, s.l0) LEAD(10, s.l1) LEAD(11, s.l2) LEAD(12, s.l0) LEAD(13, s.l1) LEAD(14, s.l2) LEAD(15, s.l0) LEAD(16, s.l1) LEAD(17, s.l2)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // chain: 12 MFMAs on the fragments read, two accumulators (hi: 4, mid: 8)
    MFMA(s.c0, s.a0, s.r[0]); MFMA(s.c1, s.a0, s.r[1]); MFMA(s.c0, s.a0, s.r[2]); MFMA(s.c1, s.a0, s.r[3]);
    MFMA(s.c1, s.a1, s.r[0]); MFMA(s.c1, s.a1, s.r[2]);
    MFMA(s.c0, s.a0, s.r[4]); MFMA(s.c1, s.a0, s.r[5]); MFMA(s.c0, s.a0, s.r[6]); MFMA(s.c1, s.a0, s.r[7]);
    MFMA(s.c1, s.a1, s.r[4]); MFMA(s.c1, s.a1, s.r[6]);
    // trailing 18 MFMAs, two epilogue VALU behind each (they depend on the chain's accumulators)
#define TRAIL(i, acc) MFMA(acc, s.a1, s.r[i % 13]); VALU(s.e0, s.c0[i & 3]); VALU(s.e1, s.c1[i & 3]);
    TRAIL(0, s.l0) TRAIL(1, s.l1) TRAIL(2, s.l2) TRAIL(3, s.l0) TRAIL(4, s.l1) TRAIL(5, s.l2) TRAIL(6, s.l0) TRAIL(7, s.l1) TRAIL(8, s.l2)
    TRAIL(9, s.l0) TRAIL(10, s.l1) TRAIL(11, s.l2) TRAIL(12, s.l0) TRAIL(13, s.l1) TRAIL(14, s.l2) TRAIL(15, s.l0) TRAIL(16, s.l1) TRAIL(17, s.l2)
    {
        typedef float floatx2 __attribute__((ext_vector_type(2)));
        const floatx2 w0 = {s.e0, s.e1}, w1 = {s.e2, s.e3};
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" ::"v"(wa), "v"(w0), "v"(w1) : "memory");
    }
}

struct HelpState {
    half8 a0;
    floatx4 z0, z1;
    half8 r[5];
    float v0, v1, v2, v3;
    int s0, s1, s2, s3;
};

template <bool WITH_MFMA, bool WITH_BARRIER>
__device__ __forceinline__ void helper_step(HelpState &h, unsigned la, floatx4 *gout, const floatx4 *gin, unsigned dmadst)
{
    if (WITH_BARRIER) asm volatile("s_barrier" ::: "memory");
#define S4 SALU(h.s0); SALU(h.s1); SALU(h.s2); SALU(h.s3);
#define V4 VALU(h.v0, h.v1); VALU(h.v1, h.v2); VALU(h.v2, h.v3); VALU(h.v3, h.v0);
    S4 V4 S4 V4 S4 V4 S4 V4 S4 V4 S4 V4 S4 V4 // 28 SALU, 28 VALU
    DSR(h.r[0], la, 0); DSR(h.r[1], la, 1024); DSR(h.r[2], la, 2048); DSR(h.r[3], la, 3072); DSR(h.r[4], la, 4096);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (WITH_MFMA) {
        MFMA(h.z0, h.r[0], h.a0); MFMA(h.z1, h.r[1], h.a0); MFMA(h.z0, h.r[2], h.a0); MFMA(h.z1, h.r[3], h.a0);
        MFMA(h.z1, h.r[0], h.a0); MFMA(h.z1, h.r[2], h.a0);
    }
    V4
    floatx4 zz = h.z0 + h.z1;
    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(gout), "v"(zz) : "memory");
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_waitcnt vmcnt(6)" ::"s"(dmadst), "v"(gin) : "memory", "m0");
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const half8 *in, floatx4 *out, floatx4 *scratch, unsigned long long *t, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid * 16; i < 65536; i += blockDim.x * 16) *(floatx4 *)(lds + i) = floatx4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
    __syncthreads();
    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned la = ldsbase + (lane >> 4) * 256 + (lane & 15) * 16 + (wave & 3) * 8192;
    const unsigned wa = ldsbase + 40960 + tid * 8;
    floatx4 *gout = scratch + (size_t)blockIdx.x * 512 + tid;
    const floatx4 *gin = scratch + (size_t)(gridDim.x + blockIdx.x) * 512 + tid;
    const unsigned dmadst = __builtin_amdgcn_readfirstlane(ldsbase + 49152 + (wave & 3) * 1024);
    const bool is_main = MODE == 6 ? wave >= 4 : wave < 4;
    unsigned long long t0 = 0, t1 = 0;
    if (MODE == 4 || is_main) {
        MainState s;
        s.a0 = in[lane]; s.a1 = in[lane + 64]; s.f2a = in[lane + 128]; s.f2b = in[lane + 192];
        s.l0 = s.l1 = s.l2 = s.c0 = s.c1 = floatx4{0, 0, 0, 0};
        s.e0 = s.e1 = s.e2 = s.e3 = 0.f;
        HelpState h;
        h.a0 = in[lane + 256]; h.z0 = h.z1 = floatx4{0, 0, 0, 0};
        h.v0 = lane; h.v1 = 1e-3f; h.v2 = 2e-3f; h.v3 = 3e-3f; h.s0 = h.s1 = h.s2 = h.s3 = 0;
        if (MODE == 5) asm volatile("s_setprio 3");
        if (MODE == 7) asm volatile("s_setprio 1");
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            main_step(s, la, wa);
            if (MODE == 4) helper_step<true, false>(h, la, gout, gin, dmadst);
        }
        t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        out[(size_t)blockIdx.x * 512 + tid] = s.l0 + s.l1 + s.l2 + s.c0 + s.c1 + floatx4{s.e0, s.e1, s.e2, s.e3} + h.z0 + h.z1 +
                                              floatx4{h.v0 + h.v1, h.v2 + h.v3, (float)(h.s0 + h.s1), (float)(h.s2 + h.s3)};
        if (blockIdx.x == 0 && lane == 0 && (wave & 3) == 0) t[0] = t1 - t0;
    } else {
        HelpState h;
        h.a0 = in[lane + 256]; h.z0 = h.z1 = floatx4{0, 0, 0, 0};
        h.v0 = lane; h.v1 = 1e-3f; h.v2 = 2e-3f; h.v3 = 3e-3f; h.s0 = h.s1 = h.s2 = h.s3 = 0;
        for (int i = 0; i < iters; ++i) {
            if (MODE == 1) asm volatile("s_barrier" ::: "memory");
            else if (MODE == 2) helper_step<false, true>(h, la, gout, gin, dmadst);
            else helper_step<true, true>(h, la, gout, gin, dmadst);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        out[(size_t)blockIdx.x * 512 + tid] = h.z0 + h.z1 + floatx4{h.v0 + h.v1, h.v2 + h.v3, (float)(h.s0 + h.s1), (float)(h.s2 + h.s3)};
    }
}

template <int MODE> void run(const char *name, int blocks, const half8 *in, floatx4 *out, floatx4 *scratch, unsigned long long *t)
{
    const int iters = 3000, threads = (MODE == 0 || MODE == 4) ? 256 : 512;
    unsigned long long h = 0;
    (void)hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 90 * 1024);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 90 * 1024, 0, in, out, scratch, t, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%-58s blocks=%3d  cycles/step = %.0f   (48 MFMAs = 768)\n", name, blocks, (double)h / iters);
    fflush(stdout);
}

int main()
{
    half8 *in; floatx4 *out, *scratch; unsigned long long *t;
    (void)hipMalloc(&in, 1024 * 16);
    _Float16 hin[8192];
    for (int i = 0; i < 8192; ++i) hin[i] = (_Float16)(((i * 37) % 17 - 8) * 0.01f);
    (void)hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 256 * 512 * 16); (void)hipMalloc(&scratch, 2 * 256 * 512 * 16); (void)hipMalloc(&t, 8);
    (void)hipMemset(scratch, 0, 2 * 256 * 512 * 16);
    for (int blocks : {1, 256}) {
        run<0>("0 main alone (4 waves)", blocks, in, out, scratch, t);
        run<1>("1 main + barrier-only helper", blocks, in, out, scratch, t);
        run<2>("2 main + helper (28 SALU 32 VALU 5 LDS 2 VMEM)", blocks, in, out, scratch, t);
        run<3>("3 main + helper + 6 MFMA", blocks, in, out, scratch, t);
        run<4>("4 one wave does both (round-1 structure)", blocks, in, out, scratch, t);
        run<5>("5 as 3, main at s_setprio 3", blocks, in, out, scratch, t);
        run<6>("6 as 3, helper = waves 0-3, main = waves 4-7", blocks, in, out, scratch, t);
        run<7>("7 as 3, main at s_setprio 1", blocks, in, out, scratch, t);
    }
    return 0;
}
