// bf16 as a storage format of activations (the *_bf16 entry points): 16-bit patterns, widened to fp32 at the load and
// rounded to nearest even at the store.  Internal header.
#pragma once
#include <hip/hip_runtime.h>

namespace ifl {

typedef unsigned short bf16_t;
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float widen(float v) { return v; }
__device__ __forceinline__ float widen(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
// round to nearest even; NaN stays NaN (quiet bit set, so that a payload in the dropped half cannot turn it into Inf)
__device__ __forceinline__ bf16_t narrow_bf16(float f)
{
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

} // namespace ifl
