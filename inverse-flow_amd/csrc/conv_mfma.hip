// Dense same-size convolution on the matrix cores (gfx950, wave64) for the shapes of the hot path:
//     out[b][co][h][w] = bias[co] + sum_{ci,kh,kw} w[co][ci][kh][kw] * in[b][ci][h-pt+kh][w-pl+kw]     (zero outside)
// with Ci = Co = C in {32, 64}, K in {2x2, 3x3}, W in {16, 32}, 0 <= pt < KH, 0 <= pl < KW.  It serves
//   * x^ = A z, the layer's reverse (inv_conv.py:249-267,442-460 -> inv_conv_with_bp.forward,
//     inv_conv_with_bp_general.cpp:44-53): corner padding, effective weight,
//   * F.conv2d of SelfNormConv (inf/layers/selfnorm.py:43) and cudnn_convolution_backward_input
//     (inf/utils/convbackward/conv2d_backward.cpp:32-53; a conv with the flipped kernel).
// The first-correct direct kernel (conv_general.hip) stays for every other shape.
//
// One workgroup = four bands of 4 output rows of one image, one after the other (the loads of the next band are in
// flight while the current one multiplies); one wave = 16 output channels (one wave per SIMD).
// The band's input (4 + KH-1 rows, W + KW-1 columns, zero halo) is staged once in LDS as split fp16
// (hi = fp16(v), lo' = fp16((v-hi) 2^11)) in the MFMA B-fragment layout [plane][pixel], plane = (k-step, hi/lo,
// k-group of 8 channels), one 16-byte piece per pixel: the fragment of ANY tap is then a plain ds_read_b128 at
// a pixel offset -- no im2col, no per-tap copies.  The weights live in AGPRs as split-fp16 A fragments (packed
// by k_convpack).  Per 16-pixel tile and tap three v_mfma_f32_16x16x32_f16 per 32-deep k-step, fp32 accumulate
// (the arithmetic of the scan); two tiles are in flight so that dependent MFMAs are 2+ apart,
// and the fragments of the next tap are requested one per MFMA while the current tap multiplies.
// A band whose input leaves the fp16 range (|v| >= 6e4) is redone in plain fp32 by the same workgroup.
#include <type_traits>
#include <utility>

#include <stdlib.h>

#include "ifl_common.h"
#include "mfma_util.h"

namespace ifl {

template <int... S, class Fn> __device__ __forceinline__ void for_seq(std::integer_sequence<int, S...>, Fn &&f)
{
    (f(std::integral_constant<int, S>{}), ...);
}

template <int C, int KH, int KW, int WT> struct ConvCfg {
    static constexpr int NW = C / 16;  // waves = 16-channel output groups
    static constexpr int NQ = C / 32;  // 32-deep k-steps per tap
    static constexpr int NT = KH * KW;
    static constexpr int RB = 4;                            // output rows per band: 52 KB of LDS at C=64, W=32 per staging
                                                            // buffer, 64 accumulator registers
    static constexpr int PR = RB + KH - 1, PC = WT + KW - 1; // staged rows / columns (halo included)
    static constexpr int PP = PR * PC;                      // staged pixels
    static constexpr int PB = PP * 16;                      // bytes of one plane
    static constexpr int NPL = NQ * 8;                      // planes: (k-step, hi/lo, k-group)
    static constexpr int LDSB = NPL * PB;                   // one staged band
    static constexpr int LDSB2 = 2 * LDSB;                  // two: a band is staged while its predecessor may still be read
    static constexpr int THREADS = 64 * NW;
    static constexpr int TPR = WT / 16;      // 16-pixel tiles per row
    static constexpr int NP = RB * TPR / 2;  // tile pairs per band
    static constexpr int BPW = 4;            // bands per workgroup: the next band's loads fly while this one multiplies
    static_assert(WT % 16 == 0 && (RB * TPR) % 2 == 0 && NP % 2 == 0, "tiles come in pairs, pairs in pairs");
    static_assert(LDSB2 <= 160 * 1024, "band staging must fit the CU's LDS");
    static_assert(4 * PB + (KH * PC + KW) * 16 < 65536, "fragment offsets must fit the ds offset field");
};

// apack[wv][t][q][hl][lane][8] (fp16): lane = m + 16 gk holds row co = 16 wv + m, k = ci = 32 q + 8 gk + j of tap t.
// With eff.weff != nullptr the launch is the whole preparation of the layer's reverse pass (ifl_forward_f32) in one:
// the effective weight What (diagonal tap forced unit-lower-triangular, or lower-triangular with its own diagonal:
// inf/utils/solve_mc.py:105-109, inf/layers/emerging/inverse_op_cython.pyx:64) is what gets packed, its fp32 copy goes to
// eff.weff (the kernel's fp32 redo of a band reads it), and workgroup 0 writes log|det A| per image
// (inf/layers/emerging/emerging_module.py:26-32; 0 for the unit diagonal) -- the reference's reverse is
// inv_conv_fwd_cuda_inverse, inv_conv_with_bp_kernel_general.cu:203-264.
__global__ __launch_bounds__(256) void k_convpack(const float *__restrict__ w, _Float16 *__restrict__ apack, int C, int KH,
                                                  int KW, ConvEff eff)
{
    const int NT = KH * KW, NQ = C / 32;
    const int tdiag = eff.dkh * KW + eff.dkw;
    const size_t total = (size_t)(C / 16) * NT * NQ * 64 * 8;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % 8), lane = (int)((i / 8) % 64);
        const int q = (int)((i / 512) % NQ), t = (int)((i / (512 * (size_t)NQ)) % NT), wv = (int)(i / (512 * (size_t)NQ * NT));
        const int co = 16 * wv + (lane & 15), ci = 32 * q + 8 * (lane >> 4) + j;
        float v = w[((size_t)co * C + ci) * NT + t];
        if (eff.weff) {
            if (t == tdiag) {
                if (ci > co) v = 0.f;
                else if (ci == co && !eff.general_diag) v = 1.f;
            }
            eff.weff[((size_t)co * C + ci) * NT + t] = v;
        }
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)((v - (float)hi) * LO_SCALE);
        const size_t base = ((((size_t)wv * NT + t) * NQ + q) * 2) * 64 * 8;
        apack[base + (size_t)lane * 8 + j] = hi;
        apack[base + (size_t)64 * 8 + (size_t)lane * 8 + j] = lo;
    }
    if (eff.logdet && blockIdx.x == 0) {
        // fixed-order tree reduction over channels -> deterministic
        __shared__ double red[256];
        double sum = 0.0;
        if (eff.general_diag)
            for (int c = threadIdx.x; c < C; c += 256) sum += log(fabs((double)w[((size_t)c * C + c) * NT + tdiag]));
        red[threadIdx.x] = sum;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        const float v = (float)(red[0] * (double)eff.H * (double)eff.W);
        for (int b = threadIdx.x; b < eff.B; b += 256) eff.logdet[b] = v;
    }
}

#ifdef IFL_STAMPS
__device__ unsigned long long *g_cstamps = nullptr;
#endif

template <int C, int KH, int KW, int WT>
__global__ __launch_bounds__(64 * (C / 16)) void k_conv_mfma(const float *__restrict__ in, const half8 *__restrict__ apack,
                                                              const float *__restrict__ w32,
                                                              const float *__restrict__ bias, float *__restrict__ out,
                                                              int H, int pt, int pl, ConvMix mix)
{
    using Cfg = ConvCfg<C, KH, KW, WT>;
    constexpr int NQ = Cfg::NQ, NT = Cfg::NT, RB = Cfg::RB, PC = Cfg::PC, PP = Cfg::PP, PB = Cfg::PB, TPR = Cfg::TPR;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int band0 = blockIdx.x * Cfg::BPW, nbands = (H + RB - 1) / RB;
    const int band1 = band0 + Cfg::BPW < nbands ? band0 + Cfg::BPW : nbands; // this workgroup's bands [band0, band1)
    const float *inb = in + (size_t)b * C * H * WT;
    float *outb = out + (size_t)b * C * H * WT;
    // reconstruction epilogue (ConvMix): what is stored is dx + coef (x - A z); the squared residuals are summed
    const float *mxb = mix.x ? mix.x + (size_t)b * C * H * WT : nullptr, *mdb = mix.x ? mix.dx + (size_t)b * C * H * WT : nullptr;
    float rsq = 0.f;
    auto finish = [&](float v, size_t idx) -> float {
        if (!mxb) return v;
        float r = mxb[idx] - v;
        r = (r == r) ? r : 0.f;
        rsq += r * r;
        return mdb[idx] + mix.coef * r;
    };

#ifdef IFL_STAMPS
    const unsigned long long cs0 = __builtin_amdgcn_s_memtime();
#endif
    // ---- weights -> registers (pinned in the accumulator half of the register file, see scan_mfma.hip) ------------
    half8 A[NT][NQ][2];
    {
        // all loads first, then the pins (a pin right behind its load makes every load wait for its own data)
        half8 Aload[NT][NQ][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
                    Aload[t][q][hl] = apack[((((size_t)wv * NT + t) * NQ + q) * 2 + hl) * 64 + lane];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    A[t][q][hl] = Aload[t][q][hl];
                    asm volatile("" : "+a"(A[t][q][hl]));
                }
    }


    // ---- staging.  Interior: item = (k-group of 8 channels, staged row, quad of 4 columns): eight 16-byte loads
    //      (one per channel, lanes along w: fully coalesced) into registers -- issued for the NEXT band before the
    //      current one is multiplied, so that the memory time of a band hides behind the MFMAs of its predecessor --
    //      then split and written as eight 16-byte LDS pieces per plane half.  Rows outside the image arrive as zeros,
    //      the halo columns are zero pieces written once. ----------------------------------------------------------
    constexpr int QPR = WT / 4, PR = Cfg::PR;
    constexpr int NITEM = (C / 8) * PR * QPR, NIT = (NITEM + Cfg::THREADS - 1) / Cfg::THREADS;
    floatx4 raw[NIT][8];
    auto fetch = [&](int band) {
        const int h0 = band * RB;
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int it = tid + u * Cfg::THREADS;
            const int qd = it % QPR, rr = (it / QPR) % PR, kg = (it / (QPR * PR)) % (C / 8);
            const int ih = h0 - pt + rr;
            const bool ok = it < NITEM && ih >= 0 && ih < H;
            const float *src = inb + ((size_t)(8 * kg) * H + (ok ? ih : 0)) * WT + 4 * qd;
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[u][j] = ok ? *(const floatx4 *)(src + (size_t)j * H * WT) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto convert = [&](unsigned char *lds) -> float { // (lds: the buffer of the band being staged)
        float vmax = 0.f;
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int it = tid + u * Cfg::THREADS;
            if (it < NITEM) {
                const int qd = it % QPR, rr = (it / QPR) % PR, kg = it / (QPR * PR);
                const int q = kg / 4, gk = kg % 4;
                const int p0 = rr * PC + pl + 4 * qd; // staged pixel of the quad's first column
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    half8 hi, lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float x = raw[u][j][e];
                        vmax = fmaxf(vmax, fabsf(x));
                        const _Float16 h16 = (_Float16)x;
                        hi[j] = h16;
                        lo[j] = (_Float16)((x - (float)h16) * LO_SCALE);
                    }
                    *(half8 *)(lds + ((q * 2 + 0) * 4 + gk) * PB + (p0 + e) * 16) = hi;
                    *(half8 *)(lds + ((q * 2 + 1) * 4 + gk) * PB + (p0 + e) * 16) = lo;
                }
            }
        }
        return vmax;
    };
    {
        const half8 zero8 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f,
                             (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        constexpr int NHC = KW - 1; // halo columns: pl on the left, KW-1-pl on the right
        for (int it = tid; it < Cfg::NPL * PR * NHC; it += Cfg::THREADS) {
            const int hc = it % NHC, rr = (it / NHC) % PR, plane = it / (NHC * PR);
            const int cc = hc < pl ? hc : WT + hc; // columns 0..pl-1 and pl+WT..PC-1
            *(half8 *)(lds + plane * PB + (rr * PC + cc) * 16) = zero8;
            *(half8 *)(lds + Cfg::LDSB + plane * PB + (rr * PC + cc) * 16) = zero8;
        }
    }
    // a band that leaves the fp16 range (or holds a NaN/Inf): plain fp32, straight from memory.  Rare and slow.
    auto band_fp32 = [&](int h0) {
        for (int o = tid; o < C * RB * WT; o += Cfg::THREADS) {
            const int ow = o % WT, r = (o / WT) % RB, co = o / (WT * RB);
            const int oh = h0 + r;
            if (oh >= H) continue;
            float acc = bias ? bias[co] : 0.f;
            for (int ci = 0; ci < C; ++ci)
                for (int kh = 0; kh < KH; ++kh) {
                    const int ih = oh - pt + kh;
                    if (ih < 0 || ih >= H) continue;
                    for (int kw = 0; kw < KW; ++kw) {
                        const int iw = ow - pl + kw;
                        if (iw < 0 || iw >= WT) continue;
                        acc = fmaf(w32[((size_t)co * C + ci) * NT + kh * KW + kw], inb[((size_t)ci * H + ih) * WT + iw], acc);
                    }
                }
            outb[((size_t)co * H + oh) * WT + ow] = finish(acc, ((size_t)co * H + oh) * WT + ow);
        }
    };

    // ---- multiply: tile pairs, taps double-buffered ---------------------------------------------------------
    int h0 = 0; // first output row of the band being multiplied
    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    unsigned fa_lane = ldsbase + g * PB + n * 16; // this lane's piece of pixel 0 in plane (0, hi, g) of the band's buffer
    const int c0 = 16 * wv + 4 * g;                    // C/D layout: lane (n, g) holds channels c0..c0+3 of pixel n
    floatx4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = floatx4{bias[c0], bias[c0 + 1], bias[c0 + 2], bias[c0 + 3]};

    // "Push" form (as in the scan): a fragment of staged row ri shifted by kw columns is the B operand of the taps
    // (kh, kw) of the output rows ri - kh, kh = 0..KH-1.  It is read from LDS once and multiplied into up to KH
    // accumulator rows -- a third of the LDS requests of a tap-by-tap loop, and 12..36 MFMAs between a stage's
    // requests and their use.  All RB x TPR output tiles of the band accumulate side by side.
    //   stage S = (ri, kw) = (S / KW, S % KW); fragments [buffer][tile of the row][k-step][hi/lo], buffers alternate
    half8 F[2][TPR][NQ][2];
    floatx4 ahi[RB][TPR], amid[RB][TPR];
    constexpr int NRD = TPR * NQ * 2; // requests of one stage
    constexpr int NST = PR * KW;      // stages of a band
    // (request / MFMA numbers are compile-time constants: with a run-time number the selection below turns into a
    // chain of branches per MFMA as soon as the enclosing loop is too long to be unrolled)
    auto request = [&](auto buf_c, auto st_c, auto j_c) {
        constexpr int BUF = decltype(buf_c)::value, S = decltype(st_c)::value, j = decltype(j_c)::value;
        constexpr int OFF = ((S / KW) * PC + (S % KW)) * 16; // the stage's pixel offset
        int c = 0;
#pragma unroll
        for (int T = 0; T < TPR; ++T) {
            const unsigned a0 = fa_lane + (16 * T) * 16;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                // plane (q, hl, g): q and the tile in the address register, hl and the stage in the offset field
                if (c++ == j) lds_read_b128_o<OFF>(F[BUF][T][q][0], a0 + q * 8 * PB);
                if (c++ == j) lds_read_b128_o<4 * PB + OFF>(F[BUF][T][q][1], a0 + q * 8 * PB);
            }
        }
    };
    // MFMA number k of stage S from buffer BUF: per k-step the hi.hi and hi.lo products of all (row, tile) targets, then
    // the lo.hi products (dependent MFMAs stay >= 2 apart); an output row's first product starts from zero
    auto mfma = [&](auto buf_c, auto st_c, auto k_c) {
        constexpr int BUF = decltype(buf_c)::value, S = decltype(st_c)::value, RI = S / KW, KWI = S % KW, k = decltype(k_c)::value;
        const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
        int c = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int T = 0; T < TPR; ++T)
                    if (RI - kh >= 0 && RI - kh < RB)
                        if (c++ == k) {
                            ahi[RI - kh][T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                A[kh * KW + KWI][q][0], F[BUF][T][q][0], (kh == 0 && KWI == 0 && q == 0) ? zero : ahi[RI - kh][T], 0, 0, 0);
                            asm volatile("" : "+a"(ahi[RI - kh][T]));
                        }
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int T = 0; T < TPR; ++T)
                    if (RI - kh >= 0 && RI - kh < RB)
                        if (c++ == k) {
                            amid[RI - kh][T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                A[kh * KW + KWI][q][0], F[BUF][T][q][1], (kh == 0 && KWI == 0 && q == 0) ? zero : amid[RI - kh][T], 0, 0, 0);
                            asm volatile("" : "+a"(amid[RI - kh][T]));
                        }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int T = 0; T < TPR; ++T)
                    if (RI - kh >= 0 && RI - kh < RB)
                        if (c++ == k) {
                            amid[RI - kh][T] =
                                __builtin_amdgcn_mfma_f32_16x16x32_f16(A[kh * KW + KWI][q][1], F[BUF][T][q][0], amid[RI - kh][T], 0, 0, 0);
                            asm volatile("" : "+a"(amid[RI - kh][T]));
                        }
    };
    auto epilogue = [&](int ro) {
        const int oh = h0 + ro;
        if (oh < H) {
#pragma unroll
            for (int T = 0; T < TPR; ++T)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    outb[((size_t)(c0 + e) * H + oh) * WT + 16 * T + n] =
                        finish(ahi[ro][T][e] + amid[ro][T][e] * LO_INV + bv[e], ((size_t)(c0 + e) * H + oh) * WT + 16 * T + n);
        }
    };
    // Stage S: its fragments (requested during the previous stage) have landed; it multiplies while the next stage's
    // requests go out one per MFMA (measured: an MFMA hides one ds_read_b128, tools/issue_rate_probe.hip).
    auto stage = [&](auto st_c) {
        constexpr int S = decltype(st_c)::value, RI = S / KW, KWI = S % KW;
        using BUF = std::integral_constant<int, S & 1>;
        using NBUF = std::integral_constant<int, 1 - (S & 1)>;
        using NEXT = std::integral_constant<int, (S + 1 < NST ? S + 1 : S)>;
        constexpr int RLO = RI - (KH - 1) > 0 ? RI - (KH - 1) : 0, RHI = RI < RB - 1 ? RI : RB - 1;
        constexpr int NMF = (RHI - RLO + 1) * TPR * NQ * 3; // (RB >= KH: every staged row has a target)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        for_seq(std::make_integer_sequence<int, (NMF > NRD ? NMF : NRD)>{}, [&](auto k_c) {
            constexpr int k = decltype(k_c)::value;
            // (MFMAs are pure: each ties its result to an opaque statement, or they sink below the requests)
            if constexpr (k < NMF) mfma(BUF{}, st_c, k_c);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (k < NRD && S + 1 < NST) request(NBUF{}, NEXT{}, k_c);
            __builtin_amdgcn_sched_barrier(0);
        });
        // the output row whose last contribution this was
        if constexpr (KWI == KW - 1 && RI - (KH - 1) >= 0 && RI - (KH - 1) < RB) epilogue(RI - (KH - 1));
    };
    static_assert(RB >= KH, "a band is at least as tall as the kernel");

#ifdef IFL_STAMPS
    unsigned long long cph[4] = {0, 0, 0, 0}, clast = __builtin_amdgcn_s_memtime();
#define IFL_CSTAMP(k)                                                  \
    do {                                                               \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
        cph[k] += t_ - clast;                                          \
        clast = t_;                                                    \
    } while (0)
#else
#define IFL_CSTAMP(k) \
    do {              \
    } while (0)
#endif
    fetch(band0);
    for (int band = band0; band < band1; ++band) {
        h0 = band * RB;
        // Staging buffers alternate: this band's conversion writes the buffer that was last read two bands ago, and every
        // wave has passed the barrier below since -- one barrier per band.
        const int boff = ((band - band0) & 1) * Cfg::LDSB;
        fa_lane = ldsbase + boff + g * PB + n * 16;
        IFL_CSTAMP(3);
        const float vmax = convert(lds + boff); // (waits for the band's loads)
        IFL_CSTAMP(0); // loads landed + split + LDS writes
        // (also the barrier between staging and use)
        const int ovf = __syncthreads_or(vmax < 6.0e4f ? 0 : 1);
        if (band + 1 < band1) fetch(band + 1); // in flight while this band multiplies
        IFL_CSTAMP(1); // barrier + issue of the next band's loads
        if (ovf) {
            band_fp32(h0);
        } else {
            for_seq(std::make_integer_sequence<int, NRD>{},
                    [&](auto j_c) { request(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, j_c); });
            for_seq(std::make_integer_sequence<int, NST>{}, [&](auto st_c) { stage(st_c); });
        }
        IFL_CSTAMP(2); // multiply + stores
    }
    if (mix.loss) { // one atomic per wave
        for (int o = 32; o > 0; o >>= 1) rsq += __shfl_down(rsq, o, 64);
        if (lane == 0) atomicAdd(mix.loss, rsq * mix.loss_scale);
    }
#ifdef IFL_STAMPS
    if (g_cstamps && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) g_cstamps[wv] = __builtin_amdgcn_s_memtime() - cs0;
    if (g_cstamps && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
        for (int k = 0; k < 4; ++k) g_cstamps[4 + k] = cph[k];
#endif
}

bool conv_mfma_supported(int Ci, int Co, int H, int W, int OH, int OW, int KH, int KW, int pt, int pl)
{
    if (Ci != Co || !(Ci == 32 || Ci == 64)) return false;
    if (!((KH == 3 && KW == 3) || (KH == 2 && KW == 2))) return false;
    if (!(W == 16 || W == 32) || OH != H || OW != W) return false;
    if (pt < 0 || pt >= KH || pl < 0 || pl >= KW) return false;
    return true;
}

size_t conv_mfma_pack_bytes(int C, int KH, int KW) { return (size_t)KH * KW * C * C * 2 * sizeof(_Float16); }

template <int C, int KH, int KW, int WT>
static int launch_conv_one(const float *in, const void *apack, const float *w, const float *bias, float *out, int B, int H,
                           int pt, int pl, const ConvMix &mix, hipStream_t s)
{
    using Cfg = ConvCfg<C, KH, KW, WT>;
    static LdsOptIn opt_in;
    if (int rc = lds_opt_in(opt_in, (const void *)k_conv_mfma<C, KH, KW, WT>, Cfg::LDSB2)) return rc;
#ifdef IFL_STAMPS
    if (const char *e = getenv("IFL_CSTAMPS")) {
        unsigned long long *ptr = (unsigned long long *)strtoull(e, nullptr, 0);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cstamps), &ptr, sizeof(ptr));
    }
#endif
    const int nbands = (H + Cfg::RB - 1) / Cfg::RB;
    const dim3 grid((nbands + Cfg::BPW - 1) / Cfg::BPW, B);
    hipLaunchKernelGGL((k_conv_mfma<C, KH, KW, WT>), grid, dim3(Cfg::THREADS), Cfg::LDSB2, s, in, (const half8 *)apack, w,
                       bias, out, H, pt, pl, mix);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

// apack: conv_mfma_pack_bytes of workspace; w: (C, C, KH, KW) fp32 (also read by the fp32 redo of a band)
int launch_conv_mfma(const float *in, const float *w, const float *bias, float *out, void *apack, int B, int C, int H,
                     int W, int KH, int KW, int pt, int pl, hipStream_t s, const ConvEff *eff, const ConvMix *mix)
{
    if (B == 0) return IFL_OK;
    const ConvMix mx = mix ? *mix : ConvMix{nullptr, nullptr, 0.f, nullptr, 0.f};
    const size_t total = (size_t)KH * KW * C * C * 2;
    ConvEff e{nullptr, nullptr, 0, 0, 0, B, H, W};
    if (eff) e = *eff;
    hipLaunchKernelGGL(k_convpack, dim3((unsigned)((total / 2 + 255) / 256)), dim3(256), 0, s, w, (_Float16 *)apack, C, KH, KW, e);
    IFL_HIP(hipGetLastError());
    if (e.weff) w = e.weff; // (the fp32 redo of a band multiplies the effective weight)
#define IFL_CASE(CC, KK, WW) \
    if (C == CC && KH == KK && W == WW) return launch_conv_one<CC, KK, KK, WW>(in, apack, w, bias, out, B, H, pt, pl, mx, s);
    IFL_CASE(64, 3, 32)
    IFL_CASE(64, 3, 16)
    IFL_CASE(32, 3, 32)
    IFL_CASE(32, 3, 16)
    IFL_CASE(64, 2, 32)
    IFL_CASE(64, 2, 16)
    IFL_CASE(32, 2, 32)
    IFL_CASE(32, 2, 16)
#undef IFL_CASE
    IFL_FAIL(IFL_EUNSUPPORTED, "launch_conv_mfma: no instantiation for C=%d K=%d W=%d", C, KH, W);
}

} // namespace ifl
