// Split-fp16 MFMA building blocks shared by the scan and the dense convolution (gfx950).  Internal header.
#pragma once
#include <hip/hip_runtime.h>

namespace ifl {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

static constexpr float LO_SCALE = 2048.0f;
static constexpr float LO_INV = 1.0f / 2048.0f;

// LDS fragment read / counted wait as inline asm: hipcc's own waitcnt insertion answers a block of
// outstanding ds_reads with lgkmcnt(0) (measured: every prefetched fragment waited for the youngest one),
// so the fragment pipeline is counted by hand.  LDS operations of a wave complete in order; the wait
// statement redefines the fragments it guards, which keeps their MFMAs behind it.
__device__ __forceinline__ void lds_read_b128(half8 &v, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
}
// ... with the constant part of the address in the instruction's offset field (one address register per row shift)
template <int OFF> __device__ __forceinline__ void lds_read_b128_o(half8 &v, unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
// the hi and lo planes of all k-steps of one fragment set
template <int NQ, int OFF> __device__ __forceinline__ void lds_read_set(half8 (&h)[NQ], half8 (&l)[NQ], unsigned addr)
{
    lds_read_b128_o<OFF>(h[0], addr);
    lds_read_b128_o<OFF + 4 * 256>(l[0], addr);
    if constexpr (NQ == 2) {
        lds_read_b128_o<OFF + 8 * 256>(h[1], addr);
        lds_read_b128_o<OFF + 12 * 256>(l[1], addr);
    }
}
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4_ __attribute__((ext_vector_type(4)));
// two floats 16 B apart (elements r, r+1 of a staged quad group) / one staged quad
template <int DW0> __device__ __forceinline__ void lds_read2_f32(floatx2 &v, unsigned addr)
{
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(DW0), "n"(DW0 + 4));
}
__device__ __forceinline__ void lds_read_f32x4(floatx4_ &v, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
}
template <int N> __device__ __forceinline__ void lgkm_wait(half8 &a, half8 &b)
{
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

// counted LDS wait with a compile-time-foldable argument
__device__ __forceinline__ void lgkm_wait_n(int n)
{
#define IFL_L(N) \
    case N: asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); break;
    switch (n < 0 ? 0 : (n > 15 ? 15 : n)) {
        IFL_L(0) IFL_L(1) IFL_L(2) IFL_L(3) IFL_L(4) IFL_L(5) IFL_L(6) IFL_L(7) IFL_L(8) IFL_L(9) IFL_L(10) IFL_L(11)
        IFL_L(12) IFL_L(13) IFL_L(14) IFL_L(15)
    }
#undef IFL_L
    __builtin_amdgcn_sched_barrier(0); // nothing that consumes the data may be scheduled above the wait
}

// power of two s with s*max in [2^6, 2^7)  (1 if max is 0 or not finite)
__device__ __forceinline__ float pow2_scale(unsigned maxbits)
{
    const int e = (int)((maxbits >> 23) & 0xff);
    if (maxbits == 0u || e == 0xff) return 1.0f;
    int se = 127 + 6 - (e - 127); // exponent field of the scale
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __uint_as_float((unsigned)se << 23);
}

// Split fp16 keeps 22 bits below each tensor's maximum.  Tensors that have grown by 2^24 or more along the sweep (an
// ill-conditioned layer: max|z| or max|dx| >= 1.7e7 for inputs of order one) peak in opposite corners, so that the
// products that make up dW pair the large entries of one with the small ones of the other: there the contraction is
// done in plain fp32 instead (by the reduce kernel).  Also taken when a maximum is not finite.
__device__ __forceinline__ bool wgrad_wide_range(unsigned a_bits, unsigned b_bits)
{
    const unsigned m = a_bits > b_bits ? a_bits : b_bits;
    return m >= 0x4B800000u; // 2^24 as a float bit pattern (non-negative floats order like unsigned; Inf/NaN above)
}

} // namespace ifl
