// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 3).  A scan for 9..32 channels on images of at most 16x16 pixels
// with one WAVE per image and the recurrence in registers: the D fragment of v_mfma_f32_16x16x16_f16 is, lane for lane, the
// B fragment of the next product, a tap's row offset is a DPP row shift.  It is CORRECT -- routed in place of k_scan_mfma it
// passed all 119 tests of tests/test_hip_parity.py -- and it is SLOWER than what it was meant to replace: 30-41 us at
// (32, 12, 16, 16) against 25 us, 37-44 us at (32, 24, 8, 8) against 16 us.  rocprofv3 --pmc on it: 11.7 k vector
// instructions per wave (SQ_INSTS_VALU / 32), the wave issuing 56 % of its cycles: a single wave is bound by instruction
// issue -- ~150 instructions a diagonal at 4-5 cycles each (the split into fp16 hi/lo, masks, DPP moves, accumulator
// traffic) next to 27 MFMAs of 16 cycles -- not by the MFMA chain (reordering the products into independent rounds changed
// nothing: tools/probes/mfma_issue_probe.hip measures 17 cycles per independent and 45 per dependent MFMA, for 16x16x16
// and 16x16x32 alike), not by the memory round trips (x, z and the folded weights staged through LDS: no change).  The
// whole-image kernel spreads the same work over four waves.  Kept as a record of the layout identity and of the measurement.
// The scan for the small images of the 32x32x3 Glows (BASELINE configs[3], [4]): 9..32 channels on images of at most 16x16
// pixels (12 channels at 16x16, 24 at 8x8 after the squeezes) -- one WAVE per image, the recurrence in registers.
//
// There the whole-image MFMA scan (scan_mfma.hip) pads the channels to 32 and pays ~0.8 us per anti-diagonal for the trip
// of the new diagonal through LDS (convert, write, barrier, read as the right-hand fragment): 25 us a scan at 16x16, 16 us
// at 8x8, 64-300 scans a training step -- the largest item of the configs[3] step.  With at most 16 pixels on a diagonal and
// at most 32 channels the trip is unnecessary: v_mfma_f32_16x16x16_f16 leaves D[row 4q+i][col n] in lane 16q+n, and takes
// B[k 4q+j][col n] from that same lane -- the output fragment of one product IS the right-hand fragment of the next
// (channels 16..31 the same way in the second half of a 16x16x32 fragment, with A's columns ordered to match).  The columns
// are the image rows h of the diagonal's pixels (h, d-h): tap (dh, dw) sends pixel h to column h + dh of diagonal
// d + dh + dw, a DPP row shift of the fragment by dh.  So a step is: r_d = x_d + acc_d, split into fp16 hi/lo, shift, and
// 3 x 9 (x 2 output tiles) MFMAs into the accumulators of the next four diagonals and into z_d -- no LDS, no barrier, and only
// the two taps with dh + dw = 1 on the dependent chain.
//
// Right-fold form (k_foldpack's fp32 copy wf[slot][in][out]): r_p = x_p + sum_t Wr_t r_{p-t}, z_p = L^-1 r_p: the recurrence is
// on r, x enters by one addition, and the z product of a step is off the chain.
// Arithmetic: split fp16 as everywhere on the path (hi = fp16(v), lo = fp16((v - hi) 2048); hi*hi in one accumulator,
// hi*lo + lo*hi in a second one scaled back by 2^-11), fp32 accumulation.  An image whose z leaves the fp16 range (or is not
// finite) is redone by the same wave with the exact fp32 body (scan_general_body.h) -- slow, never silent.
#include "ifl_common.h"
#include "mfma_util.h"
#include "scan_general_body.h"
#include <type_traits>

namespace ifl {

typedef _Float16 half4_ __attribute__((ext_vector_type(4)));

static constexpr int REG_SLOTS = 5; // accumulator sets of diagonals d .. d+4 (3x3 taps reach four diagonals ahead)

template <int NTL> struct RegFrag; // a fragment of 16 x (16 NTL) halfs: hi and scaled lo
template <> struct RegFrag<1> {
    half4_ hi, lo;
};
template <> struct RegFrag<2> {
    half8 hi, lo;
};

template <int NTL> __device__ __forceinline__ floatx4 reg_mfma(const typename std::conditional<NTL == 1, half4_, half8>::type &a,
                                                              const typename std::conditional<NTL == 1, half4_, half8>::type &b,
                                                              floatx4 c)
{
    if constexpr (NTL == 1)
        return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// split 4 NTL floats (this lane's channels of one pixel) into the fragment's hi / scaled-lo halves
template <int NTL> __device__ __forceinline__ void reg_split(RegFrag<NTL> &f, const float (&v)[4 * NTL])
{
#pragma unroll
    for (int j = 0; j < 4 * NTL; ++j) {
        const _Float16 h = (_Float16)v[j];
        f.hi[j] = h;
        f.lo[j] = (_Float16)((v[j] - (float)h) * LO_SCALE);
    }
}

// the fragment moved dh columns to the right inside its row of 16 lanes (zeros shifted in)
template <int DH, int NW> __device__ __forceinline__ void reg_shift_words(unsigned (&o)[NW], const unsigned (&i)[NW])
{
#pragma unroll
    for (int k = 0; k < NW; ++k) o[k] = DH == 0 ? i[k] : (unsigned)__builtin_amdgcn_update_dpp(0, (int)i[k], 0x110 + DH, 0xf, 0xf, true);
}
template <int DH, int NTL> __device__ __forceinline__ RegFrag<NTL> reg_shift(const RegFrag<NTL> &f)
{
    constexpr int NW = 2 * NTL;
    RegFrag<NTL> o;
    unsigned a[NW], b[NW];
    __builtin_memcpy(a, &f.hi, sizeof(a));
    reg_shift_words<DH, NW>(b, a);
    __builtin_memcpy(&o.hi, b, sizeof(b));
    __builtin_memcpy(a, &f.lo, sizeof(a));
    reg_shift_words<DH, NW>(b, a);
    __builtin_memcpy(&o.lo, b, sizeof(b));
    return o;
}

// channel of fragment element j of lane quarter q: element j < 4 from the first 16 channels, j >= 4 from the second 16
__device__ __forceinline__ int reg_channel(int q, int j) { return (j < 4 ? 0 : 16) + 4 * q + (j & 3); }

template <int NTL, int KH, int KW>
__global__ __launch_bounds__(64) void k_scan_reg(const float *__restrict__ xin, const float *__restrict__ wf, float *__restrict__ zout,
                                                 Geom g, int rh, int rw, int *__restrict__ flags, unsigned *__restrict__ amax)
{
    constexpr int NT = KH * KW, NE = 4 * NTL;
    using Frag = RegFrag<NTL>;
    extern __shared__ float smem[]; // only the fp32 fallback uses it
    const int b = blockIdx.x, lane = threadIdx.x, q = lane >> 4, n = lane & 15;
    const int C = g.C, H = g.H, W = g.W, ND = H + W - 1;

    // ---- left-hand fragments: A_s[To][row = out channel 16 To + n][k = this lane's NE input channels] = wf[s][in][out]
    //      (right fold: s < NT-1 is tap t = s + 1 with its sign, s = NT-1 is L^-1).  The folded weights come through LDS in
    //      one batch of coalesced loads: gathered straight from memory, a lane's 36 / 144 elements are as many round trips
    //      as the compiler has registers to spare -- 10-25 us before the first diagonal.
    {
        const int NW = NT * C * C;
        for (int i0 = lane; i0 < NW; i0 += 64 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = wf[min(i0 + 64 * u, NW - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + 64 * u < NW) smem[i0 + 64 * u] = v[u];
        }
    }
    __syncthreads();
    Frag A[NT][NTL];
#pragma unroll
    for (int sI = 0; sI < NT; ++sI)
#pragma unroll
        for (int To = 0; To < NTL; ++To) {
            float v[NE];
            const int co = 16 * To + n;
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int ci = reg_channel(q, j);
                const float w = smem[(sI * C + min(ci, C - 1)) * C + min(co, C - 1)];
                v[j] = (ci < C && co < C) ? w : 0.f;
            }
            reg_split<NTL>(A[sI][To], v);
        }
    __syncthreads(); // (the image's x takes the same block next)

    // this lane's pixel on diagonal d is (h, w) = (n, d - n); its elements: channels reg_channel(q, j) (as inputs) and
    // 16 To + 4 q + i (as outputs) -- the same set
    bool chan[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) chan[j] = reg_channel(q, j) < C;
    auto valid = [&](int d) { return n < H && d - n >= 0 && d - n < W; };
    // x of the whole image into LDS first, in logical order [c][h][w] (one batch of coalesced loads: a register prefetch a
    // step or two ahead is shorter than the memory latency and stalls every step); a step then reads its 4 NTL values
    const int HW = H * W, NX = C * HW;
    {
        const float *xb_ = xin + (size_t)b * NX;
        for (int i0 = lane; i0 < NX; i0 += 64 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = xb_[min(i0 + 64 * u, NX - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 64 * u;
                if (i < NX) {
                    const int c = i / HW, r = i - c * HW, hst = r / W, wst = r - hst * W;
                    smem[(c * H + (rh ? H - 1 - hst : hst)) * W + (rw ? W - 1 - wst : wst)] = v[u];
                }
            }
        }
    }
    float *zs = smem + NX; // z of the image, same order
    int xoff[NE]; // LDS index of (channel j, row n, column 0) of this lane; column d - n is added per step
#pragma unroll
    for (int j = 0; j < NE; ++j) xoff[j] = (min(reg_channel(q, j), C - 1) * H + min(n, H - 1)) * W - n;
    auto load_x = [&](int d, float (&v)[NE]) {
        const bool ok = d < ND && valid(d);
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const float t = smem[ok ? xoff[j] + d : 0];
            v[j] = (ok && chan[j]) ? t : 0.f;
        }
    };
    __syncthreads();

    // Accumulators: per diagonal slot three of them (hi hi | hi lo | lo hi), so that no chain of dependent MFMAs is longer
    // than the taps that share a slot (a dependent MFMA waits out its predecessor's whole latency, ~4x its issue time, and a
    // single wave has nothing else to issue); tap (1, 0) -- with (0, 1) the dependent chain into the next diagonal -- has a
    // set of its own (F), so that the two issue back to back.
    struct Acc {
        floatx4 hi[NTL], m1[NTL], m2[NTL];
    };
    const floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};
    Acc acc[REG_SLOTS], F;
#pragma unroll
    for (int sl = 0; sl < REG_SLOTS; ++sl)
#pragma unroll
        for (int To = 0; To < NTL; ++To) acc[sl].hi[To] = acc[sl].m1[To] = acc[sl].m2[To] = zero4;
#pragma unroll
    for (int To = 0; To < NTL; ++To) F.hi[To] = F.m1[To] = F.m2[To] = zero4;

    // the three products of one tap into one set; FRESH: the set starts from zero
    auto mma3 = [&](Acc &c, const Frag (&a)[NTL], const Frag &bf, auto fresh_c) {
        constexpr bool FRESH = decltype(fresh_c)::value;
#pragma unroll
        for (int To = 0; To < NTL; ++To) c.hi[To] = reg_mfma<NTL>(a[To].hi, bf.hi, FRESH ? zero4 : c.hi[To]);
#pragma unroll
        for (int To = 0; To < NTL; ++To) c.m1[To] = reg_mfma<NTL>(a[To].hi, bf.lo, FRESH ? zero4 : c.m1[To]);
#pragma unroll
        for (int To = 0; To < NTL; ++To) c.m2[To] = reg_mfma<NTL>(a[To].lo, bf.hi, FRESH ? zero4 : c.m2[To]);
    };
    using Fresh = std::true_type;
    using Add = std::false_type;
    // tap (dh, dw) = fold slot dh KW + dw - 1
    auto tapA = [&](int dh, int dw) -> const Frag(&)[NTL] { return A[dh * KW + dw - 1]; };

    unsigned rbits = 0, zbits = 0; // largest |r|, |z| as bit patterns (a NaN or Inf is the largest of all)
    float xa[NE], xb[NE];          // x of this diagonal and of the next
    load_x(0, xa);
    load_x(1, xb);

    // one diagonal; SL = d mod 5 selects the accumulator slots statically
    auto step = [&](auto sl_c, int d) {
        constexpr int SL = decltype(sl_c)::value;
        constexpr int S0 = SL % REG_SLOTS, S1 = (SL + 1) % REG_SLOTS, S2 = (SL + 2) % REG_SLOTS, S3 = (SL + 3) % REG_SLOTS,
                      S4 = (SL + 4) % REG_SLOTS;
        const bool ok = valid(d);
        // r_d = x_d + what the earlier diagonals pushed here
        float rv[NE];
#pragma unroll
        for (int To = 0; To < NTL; ++To)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float mid = (acc[S0].m1[To][i] + acc[S0].m2[To][i]) + (F.m1[To][i] + F.m2[To][i]);
                const float v = xa[4 * To + i] + fmaf(mid, LO_INV, acc[S0].hi[To][i] + F.hi[To][i]);
                rv[4 * To + i] = ok ? v : 0.f;
                rbits = max(rbits, __float_as_uint(rv[4 * To + i]) & 0x7fffffffu);
            }
        Frag rf;
        reg_split<NTL>(rf, rv);
        const Frag r1 = reg_shift<1, NTL>(rf);
        // round 1: one tap per set, all independent -- the two into the next diagonal first.  The farthest tap is the first
        // writer of its diagonal's set (nothing earlier reaches that far): it starts the set from zero.
        mma3(acc[S1], tapA(0, 1), rf, Add{}); // same column
        mma3(F, tapA(1, 0), r1, Fresh{});     // one column on
        Acc Z;
        mma3(Z, A[NT - 1], rf, Fresh{}); // z_d = L^-1 r_d
        if constexpr (KH == 2) {
            mma3(acc[S2], tapA(1, 1), r1, Fresh{});
        } else {
            const Frag r2 = reg_shift<2, NTL>(rf);
            mma3(acc[S2], tapA(1, 1), r1, Add{});
            mma3(acc[S3], tapA(1, 2), r1, Add{});
            mma3(acc[S4], tapA(2, 2), r2, Fresh{});
            // rounds 2 and 3: the second and third tap of a set, a round behind their predecessors
            mma3(acc[S2], tapA(0, 2), rf, Add{});
            mma3(acc[S3], tapA(2, 1), r2, Add{});
            mma3(acc[S2], tapA(2, 0), r2, Add{});
        }
        // x: this step's buffer is free
#pragma unroll
        for (int j = 0; j < NE; ++j) xa[j] = xb[j];
        load_x(d + 2, xb);
        // z_d out (its products were issued ahead of most taps: finished by now)
#pragma unroll
        for (int To = 0; To < NTL; ++To)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = fmaf(Z.m1[To][i] + Z.m2[To][i], LO_INV, Z.hi[To][i]);
                if (ok && chan[4 * To + i]) { // (into the LDS image: 64 scattered 4-byte stores per instruction cost the
                                              // memory pipeline more than the whole step; the image leaves coalesced at the end)
                    zs[xoff[4 * To + i] + d] = v;
                    zbits = max(zbits, __float_as_uint(v) & 0x7fffffffu);
                }
            }
    };

    for (int d0 = 0; d0 < ND; d0 += REG_SLOTS) { // (steps past the last diagonal have no valid pixel: they store nothing)
        step(std::integral_constant<int, 0>{}, d0);
        step(std::integral_constant<int, 1>{}, d0 + 1);
        step(std::integral_constant<int, 2>{}, d0 + 2);
        step(std::integral_constant<int, 3>{}, d0 + 3);
        step(std::integral_constant<int, 4>{}, d0 + 4);
    }

    __syncthreads();
    { // z out, in stored order
        float *zb_ = zout + (size_t)b * NX;
        for (int i0 = lane; i0 < NX; i0 += 64 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 64 * u;
                if (i < NX) {
                    const int c = i / HW, r = i - c * HW, hst = r / W, wst = r - hst * W;
                    zb_[i] = zs[(c * H + (rh ? H - 1 - hst : hst)) * W + (rw ? W - 1 - wst : wst)];
                }
            }
        }
    }
    // r out of the fp16 range (or not finite) anywhere in the image: the exact fp32 body, by this wave, from x
    for (int o = 32; o > 0; o >>= 1) {
        rbits = max(rbits, (unsigned)__shfl_xor((int)rbits, o, 64));
        zbits = max(zbits, (unsigned)__shfl_xor((int)zbits, o, 64));
    }
    if (rbits >= __float_as_uint(6.0e4f)) {
        if (flags && lane == 0) flags[b] = 4;
        __syncthreads();
        scan_general_body<64>(xin, wf, zout, g, rh, rw, 1, smem, b, lane);
        if (amax) { // the maximum of what the redo wrote
            __syncthreads();
            float zm = 0.f;
            for (int i = lane; i < C * H * W; i += 64) zm = fmaxf(zm, fabsf(zout[(size_t)b * C * H * W + i]));
            zbits = __float_as_uint(zm);
            for (int o = 32; o > 0; o >>= 1) zbits = max(zbits, (unsigned)__shfl_xor((int)zbits, o, 64));
        }
    }
    if (amax && lane == 0) atomicMax(amax, zbits); // one atomic per wave; max is order-independent
}

bool scan_reg_supported(const Geom &g)
{
    return g.C >= 9 && g.C <= 32 && g.H >= 1 && g.H <= 16 && g.W >= 1 && g.W <= 16 &&
           ((g.KH == 3 && g.KW == 3) || (g.KH == 2 && g.KW == 2));
}

// wf: the fp32 right fold [slot][in][out] (k_foldpack's fp32 copy: slot < KH KW - 1: the taps with their sign, last: L^-1)
int launch_scan_reg(const float *x, const float *wf, float *z, const Geom &g, int rh, int rw, int *flags, unsigned *amax, hipStream_t s)
{
    if (!scan_reg_supported(g)) IFL_FAIL(IFL_EUNSUPPORTED, "launch_scan_reg: C=%d %dx%d K=%dx%d", g.C, g.H, g.W, g.KH, g.KW);
    if (g.B == 0) return IFL_OK;
    size_t lds = scan_general_lds_bytes(g); // (the fp32 fallback's; the sweep stages the image's x in the same block)
    if (lds < (size_t)2 * g.C * g.H * g.W * sizeof(float)) lds = (size_t)2 * g.C * g.H * g.W * sizeof(float); // x and z images
    if (lds < (size_t)g.KH * g.KW * g.C * g.C * sizeof(float)) lds = (size_t)g.KH * g.KW * g.C * g.C * sizeof(float); // (and the folded weights)
    if (lds > 64 * 1024) IFL_FAIL(IFL_EUNSUPPORTED, "launch_scan_reg: workspace of %zu bytes", lds);
#define IFL_REG(NTL, K) hipLaunchKernelGGL((k_scan_reg<NTL, K, K>), dim3(g.B), dim3(64), lds, s, x, wf, z, g, rh, rw, flags, amax)
    if (g.C <= 16) {
        if (g.KH == 3) IFL_REG(1, 3); else IFL_REG(1, 2);
    } else {
        if (g.KH == 3) IFL_REG(2, 3); else IFL_REG(2, 2);
    }
#undef IFL_REG
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

} // namespace ifl
