// MFMA weight gradient of the inverse conv (gfx950, wave64):
//     dW[co][ci][kh][kw] = scale * sum_{b,oh,ow} gz[b][co][oh][ow] * x[b][ci][oh-pt+kh][ow-pl+kw]
// for the four padding corners (pt in {0,KH-1}, pl in {0,KW-1}; inf/layers/inv_conv.py:126-144),
// i.e. the dense contraction M=C, N=C*KH*KW, K=B*H*W of SURVEY 2.2 (reference: 2*80 serial
// launches + a 302 MB scratch tensor, inv_conv_with_bp_kernel_general.cu:496-735).
//
// Everything is expressed in *stored* coordinates through one canonical form
//     G[p][q][i][j] = sum_{b,r,s} a[b][p][r][s+j] * bb[b][q][r - sg*i][s]        (zero outside the image)
// with (a, bb, sg) = (gz, x, +1) TL, (gz, x, -1) BL, (x, gz, -1) TR, (x, gz, +1) BR; for the R orders
// G is the transpose of dW (handled by the reduce kernel).  The reduction index of an MFMA is the
// column s of one image row: both operands are then 8 consecutive floats per lane straight from
// NCHW memory -- no im2col, no LDS transposes.  Row r - sg*i of bb is simply the fragment converted
// i row-steps earlier (rolling registers), the column shift s+j of a is made in registers from the
// unshifted fragment (v_alignbit + one cross-half shuffle for the carried-in head).
//
// Work split: output block = 32x32 channels (one v_mfma_f32_32x32x16_f16 tile per tap), one wave owns
// one block for a run of RPW consecutive rows of one image (its slice of the K dimension); the four
// waves of a workgroup own four K-slices of the same block and sum them through LDS, so one fp32
// partial per workgroup goes to memory (9.4 MB at the north-star shape) and a second kernel sums the
// partials in a fixed order (deterministic; no float atomics), applies sign/scale, the gradient mask
// (inf/layers/inv_conv.py:223-248) and the tap index map.
//
// Arithmetic: split fp16 (hi = fp16(v), lo = fp16(v - hi), denormals kept -- verified on gfx950 by
// tools/mfma_f16_denorm_probe.hip) after a power-of-two per-tensor prescale that puts max|v| at 2^6,
// three MFMAs per k-step (hi*hi + hi*lo + lo*hi), fp32 accumulation.
#include "ifl_common.h"
#include "mfma_util.h"
#include <type_traits>
#include <stdlib.h>

namespace ifl {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

static constexpr int WG_RPW = 16; // image rows per wave (K-slice)

// ---- max|v| of two tensors (bit pattern of a non-negative float orders like an unsigned) ---------
__global__ __launch_bounds__(256) void k_absmax2(const float *__restrict__ a, const float *__restrict__ b, size_t n,
                                                 unsigned *__restrict__ out)
{
    float ma = 0.f, mb = 0.f;
    const size_t n4 = n / 4;
    const floatx4 *a4 = (const floatx4 *)a, *b4 = (const floatx4 *)b;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const floatx4 va = a4[i], vb = b4[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ma = fmaxf(ma, fabsf(va[j]));
            mb = fmaxf(mb, fabsf(vb[j]));
        }
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
            ma = fmaxf(ma, fabsf(a[i]));
            mb = fmaxf(mb, fabsf(b[i]));
        }
    for (int o = 32; o > 0; o >>= 1) {
        ma = fmaxf(ma, __shfl_down(ma, o, 64));
        mb = fmaxf(mb, __shfl_down(mb, o, 64));
    }
    __shared__ float red[2][4];
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = ma;
        red[1][threadIdx.x >> 6] = mb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // one atomic per block and tensor (same-address atomics cost ~12 ns each); max is exact in any
        // order.  A NaN/Inf input gives a huge pattern -> tiny scale: the result stays NaN/Inf (loud).
        ma = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        mb = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        atomicMax(out + 0, __float_as_uint(ma));
        atomicMax(out + 1, __float_as_uint(mb));
    }
}

__device__ __forceinline__ void split8(const floatx4 &v0, const floatx4 &v1, float s, half8 &hi, half8 &lo)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x0 = v0[j] * s, x1 = v1[j] * s;
        const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
        hi[j] = h0;
        hi[4 + j] = h1;
        lo[j] = (_Float16)(x0 - (float)h0);
        lo[4 + j] = (_Float16)(x1 - (float)h1);
    }
}

// fragment shifted by J columns: element k <- element k+J.  `head` = first dword of the next 8-column block,
// `tail` = last dword of the previous one (both already exchanged across the wave halves)
template <int J> __device__ __forceinline__ half8 shift_frag(const half8 &f, unsigned head, unsigned tail)
{
    const uintx4 d = __builtin_bit_cast(uintx4, f);
    uintx4 o;
    if (J == 0) {
        o = d;
    } else if (J == 1) {
        o[0] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
        o[1] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
        o[2] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
        o[3] = __builtin_amdgcn_alignbit(head, d[3], 16);
    } else if (J == 2) {
        o[0] = d[1];
        o[1] = d[2];
        o[2] = d[3];
        o[3] = head;
    } else { // J == -1
        o[0] = __builtin_amdgcn_alignbit(d[0], tail, 16);
        o[1] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
        o[2] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
        o[3] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
    }
    return __builtin_bit_cast(half8, o);
}

#ifdef IFL_STAMPS
__device__ unsigned long long *g_wstamps = nullptr;
#define IFL_WSTAMP(k)                                                \
    do {                                                             \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
        wst[k] += t_ - wlast;                                        \
        wlast = t_;                                                  \
    } while (0)
#else
#define IFL_WSTAMP(k) \
    do {              \
    } while (0)
#endif

// C channels, KHxKW taps (KW <= 3), NKS = W/16 k-steps per image row.
// CEN = 1: the centered ("same") padding of a 3x3 kernel (SelfNormConv, selfnorm.py:52-82): the canonical form with
// the a operand one row late and one column to the right, G[i][j] = sum a[r - sg][s + j - 1] bb[r - sg i][s]; the
// slice that ends the image runs one more step against a zero row of bb.
template <int C, int KH, int KW, int NKS, int CEN>
__global__ __launch_bounds__(256) void k_wgrad_mfma(const float *__restrict__ a, const float *__restrict__ bb,
                                                    float *__restrict__ partial, const unsigned *__restrict__ amax_a,
                                                    const unsigned *__restrict__ amax_b, int B, int H, int sg,
                                                    int ntask)
{
    constexpr int W = 16 * NKS;
    constexpr int NT = KH * KW;
    constexpr int NB = C / 32; // 32-channel blocks per side
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = lane & 31, hh = lane >> 5;
    // The NB x NB channel blocks of one split read the same image rows (block (p, q): channels p of a, q of bb).  Workgroups go
    // to the 8 XCDs round robin (id mod 8) and every XCD has its own L2: numbered consecutively, a split's blocks sit on
    // different XCDs and every operand row is fetched from memory NB times (rocprofv3 FETCH_SIZE, calibrated with
    // tools/probes/fetch_size_probe.hip: 146 MB for the 67 MB of operands at the north-star shape).  So the blocks of a
    // split are given ids of the same residue mod 8 -- they run at the same time on one XCD and share its L2.
    int blk, split;
    {
        constexpr int NB2 = NB * NB;
        const int w = blockIdx.x, nsplit = gridDim.x / NB2, full = (nsplit / 8) * 8 * NB2;
        if (w < full) {
            const int xcd = w & 7, j = w >> 3;
            split = (j / NB2) * 8 + xcd;
            blk = j % NB2;
        } else {
            blk = w % NB2;
            split = w / NB2;
        }
    }
    const int bp = blk / NB, bq = blk % NB; // block row (a channels) / column (bb channels)

    // max|a|, max|bb| as float bit patterns (upper bounds are fine: they only pick a power-of-two scale)
    if (wgrad_wide_range(*amax_a, *amax_b)) return; // (uniform) the reduce kernel computes dW in fp32 instead
    const float sa = pow2_scale(*amax_a);
    const float sb = pow2_scale(*amax_b);

    floatx16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#ifdef IFL_STAMPS
    unsigned long long wst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wlast = __builtin_amdgcn_s_memtime();
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime(); // (100 MHz, the same clock on every compute unit)
#endif

    const int rgroups = (H + WG_RPW - 1) / WG_RPW;
    const int task = split * 4 + wv; // (image, row group)
    if (task < ntask) {
        const int b = task / rgroups, rg = task % rgroups;
        const int r_lo = rg * WG_RPW, r_hi = (r_lo + WG_RPW < H ? r_lo + WG_RPW : H) - 1;
        // walk rows in direction sg so that rows r - sg*i were seen i steps earlier
        const int r_first = sg > 0 ? r_lo : r_hi;
        const int nrows = r_hi - r_lo + 1;
        const float *ap = a + ((size_t)b * C + 32 * bp + m) * H * W + 8 * hh;
        const float *bpz = bb + ((size_t)b * C + 32 * bq + m) * H * W + 8 * hh;

        // rows of bb in a ring of four register sets, rotated by NAME (the row loop is unrolled by four): in phase P the
        // current row is set P, the row i steps earlier set (P - i) mod 4, and the next row is converted into set
        // (P + 1) mod 4 -- the one whose row has just left the kernel's reach.  (Rolling the VALUES cost 32-64 register
        // moves per row.)
        constexpr int RB = 4;
        static_assert(KH < RB, "ring: KH rows in use + the one being converted");
        half8 Bz[RB][NKS][2];
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
#pragma unroll
                    for (int j = 0; j < 8; ++j) Bz[i][ks][hl][j] = (_Float16)0.f;

        // halo: the KH-1 rows of bb before the first row of this slice (loads issued together, then converted)
        {
            floatx4 hv[KH][NKS][2];
#pragma unroll
            for (int i = KH - 1; i >= 1; --i) {
                const int r = r_first - sg * i;
                const int rc = (r >= 0 && r < H) ? r : r_first;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    hv[i][ks][0] = *(const floatx4 *)(bpz + (size_t)rc * W + 16 * ks);
                    hv[i][ks][1] = *(const floatx4 *)(bpz + (size_t)rc * W + 16 * ks + 4);
                }
            }
#pragma unroll
            for (int i = KH - 1; i >= 1; --i) {
                const int r = r_first - sg * i;
                if (r >= 0 && r < H) {
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) split8(hv[i][ks][0], hv[i][ks][1], sb, Bz[(RB - i) % RB][ks][0], Bz[(RB - i) % RB][ks][1]);
                }
            }
        }

        // Software pipeline (one wave per SIMD: nothing else hides latency or the conversion work):
        //   iteration `step` multiplies row r (fragments Au, Bz converted one iteration ago) while it converts the
        //   raw registers of row r+1 in the MFMAs' shadow and then reloads them with row r+3.  Two raw sets
        //   alternate (loop unrolled by two), so a load has two iterations to land and nothing ever waits on
        //   a load it has just issued.
        floatx4 ra[2][NKS][2], rb[2][NKS][2];
        // steps of this slice: one per row, plus (CEN) the step past the image's last row in walking direction
        const int nsteps = nrows + ((CEN && (sg > 0 ? r_hi == H - 1 : r_lo == 0)) ? 1 : 0);
        auto fetch = [&](auto set_c, int k) { // step k of the slice (steps past it: a valid row, never used)
            constexpr int SET = decltype(set_c)::value;
            const int rb_ = r_first + sg * k, ra_ = rb_ - sg * CEN; // a runs CEN rows late
            const bool okb = rb_ >= 0 && rb_ < H, oka = ra_ >= 0 && ra_ < H;
            const int rbc = okb ? rb_ : r_first, rac = oka ? ra_ : r_first;
            const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                ra[SET][ks][0] = *(const floatx4 *)(ap + (size_t)rac * W + 16 * ks);
                ra[SET][ks][1] = *(const floatx4 *)(ap + (size_t)rac * W + 16 * ks + 4);
                rb[SET][ks][0] = *(const floatx4 *)(bpz + (size_t)rbc * W + 16 * ks);
                rb[SET][ks][1] = *(const floatx4 *)(bpz + (size_t)rbc * W + 16 * ks + 4);
                if (CEN) { // rows outside the image are zero rows (without CEN they are never used)
                    if (!oka) ra[SET][ks][0] = ra[SET][ks][1] = zero;
                    if (!okb) rb[SET][ks][0] = rb[SET][ks][1] = zero;
                }
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        half8 Ab[2][NKS][2]; // a fragments of the current row (set P mod 2) and of the next one
        fetch(S0{}, 0);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            split8(ra[0][ks][0], ra[0][ks][1], sa, Ab[0][ks][0], Ab[0][ks][1]);
            split8(rb[0][ks][0], rb[0][ks][1], sb, Bz[0][ks][0], Bz[0][ks][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch(S1{}, 1); // row 1 -> set 1 (converted in iteration 0), row 2 -> set 0 (converted in iteration 1)
        fetch(S0{}, 2);
        __builtin_amdgcn_sched_barrier(0);

        IFL_WSTAMP(0); // prologue
        // iteration `step`: converts set SET = (step+1) mod 2 (row step+1), reloads it with row step+3
        auto row_step = [&](auto ph_c, const int step) {
            constexpr int P = decltype(ph_c)::value; // step mod 4
            constexpr int SET = (P + 1) & 1;
            using set_t = std::integral_constant<int, SET>;
            auto &Au = Ab[P & 1];
            auto &An = Ab[(P + 1) & 1];
            auto &Bn = Bz[(P + 1) % RB];
            // heads of the next 8-column block, exchanged across the two wave halves:
            // a lane with hh=0 needs the hh=1 lane's block of the same k-step, a lane with hh=1 the
            // hh=0 lane's block of the next k-step (zero past the row end)
            unsigned head[NKS][2], tail[NKS][2];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    const unsigned own = __builtin_bit_cast(uintx4, Au[ks][hl])[0];
                    const unsigned nxt = ks + 1 < NKS ? __builtin_bit_cast(uintx4, Au[ks + 1 < NKS ? ks + 1 : ks][hl])[0] : 0u;
                    const unsigned send = hh ? own : nxt;
                    // exchange across the wave halves without the LDS crossbar: after the swap the first result holds
                    // the low half's values in both halves, the second the high half's
                    const auto sw = __builtin_amdgcn_permlane32_swap(send, send, false, false);
                    head[ks][hl] = hh ? sw[0] : sw[1];
                    tail[ks][hl] = 0u;
                    if (CEN) {
                        // tails of the previous 8-column block: a lane with hh=1 needs the hh=0 lane's block of the
                        // same k-step, a lane with hh=0 the hh=1 lane's block of the previous k-step (zero at the row start)
                        const unsigned ownt = __builtin_bit_cast(uintx4, Au[ks][hl])[3];
                        const unsigned prvt = ks > 0 ? __builtin_bit_cast(uintx4, Au[ks > 0 ? ks - 1 : 0][hl])[3] : 0u;
                        const unsigned sendt = hh ? prvt : ownt;
                        const auto swt = __builtin_amdgcn_permlane32_swap(sendt, sendt, false, false);
                        tail[ks][hl] = hh ? swt[0] : swt[1];
                    }
                }
            // note: the hh=1 lane of the LAST k-step receives `nxt` of its partner = 0 (row end) as
            // intended; the hh=0 lane receives the partner's own block.
            half8 As[KW][NKS][2];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    As[0][ks][hl] = shift_frag<0 - CEN>(Au[ks][hl], head[ks][hl], tail[ks][hl]);
                    if constexpr (KW > 1) As[1][ks][hl] = shift_frag<1 - CEN>(Au[ks][hl], head[ks][hl], tail[ks][hl]);
                    if constexpr (KW > 2) As[2][ks][hl] = shift_frag<2 - CEN>(Au[ks][hl], head[ks][hl], tail[ks][hl]);
                }
            // (letting the shifts of tap columns 1 and 2 slide into the MFMAs' shadow -- no barrier here, five VALU
            // instructions per MFMA below -- was measured 0.8 us SLOWER; so was requesting the slice's first three rows
            // and the halo in one go: 3400 cycles more in the prologue, with every wave of the grid asking at once)
            __builtin_amdgcn_sched_barrier(0);
            IFL_WSTAMP(1); // shifts
            // the next row's fragments
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                split8(ra[SET][ks][0], ra[SET][ks][1], sa, An[ks][0], An[ks][1]);
                split8(rb[SET][ks][0], rb[SET][ks][1], sb, Bn[ks][0], Bn[ks][1]);
            }
            // consecutive MFMAs go to different accumulators (rows i of the tap column j): back-to-back MFMAs into
            // one accumulator issue ~1.5x slower from a single wave (tools/mfma_rate_probe.hip)
#pragma unroll
            for (int j = 0; j < KW; ++j)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int i = 0; i < KH; ++i)
                            acc[i * KW + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(As[j][ks][pr == 2 ? 1 : 0], Bz[(P - i + RB) % RB][ks][pr == 1 ? 1 : 0],
                                                                                     acc[i * KW + j], 0, 0, 0);
            // the conversion rides in the MFMAs' shadow: an MFMA holds the matrix pipe for 32 cycles
#pragma unroll
            for (int k = 0; k < KW * NKS * 3 * KH; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, P);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, P);
            }
            __builtin_amdgcn_sched_barrier(0);
            IFL_WSTAMP(3); // MFMAs + conversion of the next row
            fetch(set_t{}, step + 3); // (unconditional: a branch would park the loads elsewhere)
            __builtin_amdgcn_sched_barrier(0);
            IFL_WSTAMP(2); // loads
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        using P2 = std::integral_constant<int, 2>;
        using P3 = std::integral_constant<int, 3>;
        int step = 0;
        for (; step + 3 < nsteps; step += 4) {
            row_step(P0{}, step);
            row_step(P1{}, step + 1);
            row_step(P2{}, step + 2);
            row_step(P3{}, step + 3);
        }
        if (step < nsteps) row_step(P0{}, step);
        if (step + 1 < nsteps) row_step(P1{}, step + 1);
        if (step + 2 < nsteps) row_step(P2{}, step + 2);
    }
    IFL_WSTAMP(4);

    // ---- sum of the four waves' accumulators through LDS, fixed order (a0 + a2) + (a1 + a3) ---------------------------
    // Every wave dumps its registers, one barrier, then each wave sums and stores a quarter of the vectors: the
    // three-stage tree this replaces (two waves dump, two add, one dumps, one adds and stores: three barriers, the
    // store by one wave) cost 7000 + 2200 of the kernel's 52000 cycles; this, 2900 + 1400.  Same additions, same order.
    floatx4 *red = (floatx4 *)smem; // [4][NT*4][64] float4
    constexpr int NV = NT * 4;      // float4 vectors per lane
    {
        floatx4 *dst = red + (size_t)wv * NV * 64;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                dst[(t * 4 + v) * 64 + lane] = floatx4{acc[t][4 * v], acc[t][4 * v + 1], acc[t][4 * v + 2], acc[t][4 * v + 3]};
    }
    __syncthreads();
    IFL_WSTAMP(5); // dump + barrier
    {
        // partial[split][block][t][v][lane] (float4): the accumulator registers as they are -- register 4v+e of lane
        // (m, hh) is row (e) + 8v + 4hh, column m of the 32x32 block (C/D layout of the 32x32 MFMA); the reduce
        // kernel undoes the layout.  One 16-byte store per register quad (unscaled: the reduce kernel divides).
        floatx4 *out = (floatx4 *)partial + ((size_t)split * NB * NB + blk) * NT * 4 * 64;
#pragma unroll
        for (int k = 0; k < (NV + 3) / 4; ++k) {
            const int v = 4 * k + wv;
            if (v < NV) {
                const floatx4 s0 = red[(0 * NV + v) * 64 + lane], s1 = red[(1 * NV + v) * 64 + lane];
                const floatx4 s2 = red[(2 * NV + v) * 64 + lane], s3 = red[(3 * NV + v) * 64 + lane];
                out[v * 64 + lane] = (s0 + s2) + (s1 + s3);
            }
        }
    }
    IFL_WSTAMP(6); // partial store
#ifdef IFL_STAMPS
    if (g_wstamps && blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 8; ++k) g_wstamps[wv * 8 + k] = wst[k];
    if (g_wstamps && tid == 0 && blockIdx.x < 1024) { // every workgroup's entry and exit on the common clock
        g_wstamps[32 + 2 * blockIdx.x] = wg_t0;
        g_wstamps[32 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// dw[co][ci][kh][kw] = scale/(sa*sb) * sum_splits partial[s][block][t=(i,j)][v][lane][e]; the block's (row, column) is
// (co,ci) or (ci,co).  A thread owns one float4 of the register dump (four rows of one column); four waves share
// it (each sums every 4th partial, then a fixed-order sum): deterministic.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ partial, float *__restrict__ dw,
                                                      const unsigned *__restrict__ amax_a,
                                                      const unsigned *__restrict__ amax_b, int nsplit, int C, int KH,
                                                      int KW, int swapped, int top, int left, float scale,
                                                      int mask_mode, int mkh, int mkw, const float *__restrict__ gz,
                                                      const float *__restrict__ x, int B, int H, int W, int pt, int pl)
{
    const int NT = KH * KW, NB = C / 32;
    const size_t total4 = (size_t)NB * NB * NT * 4 * 64; // float4 elements of one partial
    if (wgrad_wide_range(*amax_a, *amax_b)) {
        // plain fp32 contraction (see wgrad_wide_range): a wave owns row e = its index of each of the block's 64
        // float4 elements in turn; its lanes stride over the pixels (coalesced along w), fixed-order butterfly
        const int l = threadIdx.x & 63, e = threadIdx.x >> 6;
        const size_t npix = (size_t)B * H * W;
        for (int o = 0; o < 64; ++o) {
            const size_t ido = (size_t)blockIdx.x * 64 + o;
            if (ido >= total4) break;
            const int lane = (int)(ido % 64), v = (int)((ido / 64) % 4), t = (int)((ido / 256) % NT), blk = (int)(ido / (256 * (size_t)NT));
            const int bp = blk / NB, bq = blk % NB, m = lane & 31, hh = lane >> 5;
            const int i = t / KW, j = t % KW;
            const int kh = top ? KH - 1 - i : i, kw = left ? KW - 1 - j : j;
            const int q = 32 * bq + m, p = 32 * bp + e + 8 * v + 4 * hh;
            const int co = swapped ? q : p, ci = swapped ? p : q;
            float acc = 0.f;
            for (size_t pix = l; pix < npix; pix += 64) {
                const int ow = (int)(pix % W), oh = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
                const int ih = oh - pt + kh, iw = ow - pl + kw;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W)
                    acc = fmaf(gz[(((size_t)b * C + co) * H + oh) * W + ow], x[(((size_t)b * C + ci) * H + ih) * W + iw], acc);
            }
            for (int sft = 32; sft > 0; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
            if (l == 0) {
                float val = acc * scale;
                if (mask_mode && kh == mkh && kw == mkw) {
                    if (mask_mode == 1 && ci >= co) val = 0.f;
                    if (mask_mode == 2 && ci > co) val = 0.f;
                }
                dw[(((size_t)co * C + ci) * KH + kh) * KW + kw] = val;
            }
        }
        return;
    }
    const float inv = scale / (pow2_scale(*amax_a) * pow2_scale(*amax_b));
    const int sub = threadIdx.x >> 6;                                // wave index = partial residue class
    const size_t idx = (size_t)blockIdx.x * 64 + (threadIdx.x & 63); // float4 element
    __shared__ floatx4 red[4][64];
    floatx4 s = {0.f, 0.f, 0.f, 0.f};
    if (idx < total4) {
        // (all loads of a batch in flight before the first add: the sum keeps its order, the latencies overlap)
        int k = sub;
        for (; k + 28 < nsplit; k += 32) {
            floatx4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ((const floatx4 *)partial)[(size_t)(k + 4 * u) * total4 + idx];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nsplit; k += 4) s += ((const floatx4 *)partial)[(size_t)k * total4 + idx];
    }
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && idx < total4) {
        const int l = threadIdx.x & 63;
        s = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
        const int lane = (int)(idx % 64), v = (int)((idx / 64) % 4), t = (int)((idx / 256) % NT), blk = (int)(idx / (256 * (size_t)NT));
        const int bp = blk / NB, bq = blk % NB, m = lane & 31, hh = lane >> 5;
        const int i = t / KW, j = t % KW;
        const int kh = top ? KH - 1 - i : i, kw = left ? KW - 1 - j : j;
        const int q = 32 * bq + m;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int p = 32 * bp + e + 8 * v + 4 * hh;
            const int co = swapped ? q : p, ci = swapped ? p : q;
            float val = s[e] * inv;
            if (mask_mode && kh == mkh && kw == mkw) {
                if (mask_mode == 1 && ci >= co) val = 0.f;
                if (mask_mode == 2 && ci > co) val = 0.f;
            }
            dw[(((size_t)co * C + ci) * KH + kh) * KW + kw] = val;
        }
    }
}

int launch_absmax(const float *a, size_t n, unsigned *out, hipStream_t s)
{
    // out[0] <- max|a|; the second result of k_absmax2 goes to the spare word behind it (carry header slack)
    IFL_HIP(hipMemsetAsync(out, 0, 2 * sizeof(unsigned), s));
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_absmax2, dim3((unsigned)blocks), dim3(256), 0, s, a, a, n, out);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

int launch_absmax2(const float *a, const float *b, size_t n, unsigned *out, hipStream_t s)
{
    IFL_HIP(hipMemsetAsync(out, 0, 2 * sizeof(unsigned), s));
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_absmax2, dim3((unsigned)blocks), dim3(256), 0, s, a, b, n, out);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

bool wgrad_mfma_supported(int B, int C, int H, int W, int KH, int KW, int pt, int pl, const void *gz, const void *x)
{
    if (!(C == 32 || C == 64 || C == 128 || C == 256)) return false; // (wide layers: corner pads only, below)
    if (!(W == 16 || W == 32)) return false;
    if (KH < 1 || KH > 3 || KW < 1 || KW > 3 || KH != KW) return false;
    if (!(KH == 3 || KH == 2)) return false;
    const bool corner = (pt == 0 || pt == KH - 1) && (pl == 0 || pl == KW - 1);
    const bool centered = KH == 3 && KW == 3 && pt == 1 && pl == 1; // SelfNormConv's "same" padding
    if (!corner && !centered) return false;
    if (C > 64 && centered) return false;
    if (KH == 1 || KW == 1) return false;
    if (B < 1 || H < 1) return false;
    if (((uintptr_t)gz | (uintptr_t)x) & 15) return false;
    return true;
}

static int wgrad_ntask(int B, int H) { return B * ((H + WG_RPW - 1) / WG_RPW); }

size_t wgrad_mfma_workspace_bytes(int B, int C, int H, int KH, int KW)
{
    const int nsplit = (wgrad_ntask(B, H) + 3) / 4;
    return 256 + (size_t)nsplit * KH * KW * C * C * sizeof(float);
}

template <int C, int KH, int KW, int NKS, int CEN>
static int launch_wg(const float *a, const float *bb, float *partial, const unsigned *amax_a, const unsigned *amax_b,
                     int B, int H, int sg, hipStream_t s)
{
    const int ntask = wgrad_ntask(B, H);
    const int nsplit = (ntask + 3) / 4;
    constexpr int NB = C / 32;
    const size_t lds = (size_t)4 * KH * KW * 4 * 64 * sizeof(floatx4); // the four waves' accumulators
    static LdsOptIn opt_in;
    if (int rc = lds_opt_in(opt_in, (const void *)k_wgrad_mfma<C, KH, KW, NKS, CEN>, (int)lds)) return rc;
#ifdef IFL_STAMPS
    if (const char *e = getenv("IFL_WSTAMPS")) {
        unsigned long long *ptr = (unsigned long long *)strtoull(e, nullptr, 0);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wstamps), &ptr, sizeof(ptr));
    }
#endif
    hipLaunchKernelGGL((k_wgrad_mfma<C, KH, KW, NKS, CEN>), dim3(nsplit * NB * NB), dim3(256), lds, s, a, bb, partial, amax_a,
                       amax_b, B, H, sg, ntask);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

int launch_wgrad_mfma(const float *gz, const float *x, float *dw, void *ws, int B, int C, int H, int W, int KH, int KW,
                      int pt, int pl, float scale, int mask_mode, int mkh, int mkw, const unsigned *amax_gz,
                      const unsigned *amax_x, hipStream_t s)
{
    unsigned *absmax = (unsigned *)ws;
    float *partial = (float *)((char *)ws + 256);
    const size_t n = (size_t)B * C * H * W;
    if (!amax_gz || !amax_x) {
        // no producer handed the maxima over: one streaming pass over both tensors
        IFL_HIP(hipMemsetAsync(absmax, 0, 2 * sizeof(unsigned), s));
        size_t blocks = (n / 4 + 255) / 256;
        if (blocks > 512) blocks = 512;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_absmax2, dim3((unsigned)blocks), dim3(256), 0, s, gz, x, n, absmax);
        IFL_HIP(hipGetLastError());
        amax_gz = absmax;
        amax_x = absmax + 1;
    }
    const bool centered = KH == 3 && KW == 3 && pt == 1 && pl == 1;
    // canonical form (see header): L orders shift gz, R orders shift x and transpose the result; the centered
    // padding is the transposed form walked downwards with x one row late and one column right (kernel's CEN)
    const int top = centered ? 0 : pt != 0, left = centered ? 0 : pl != 0;
    const int swapped = centered ? 1 : !left;
    const float *a = swapped ? x : gz, *bb = swapped ? gz : x;
    const int sg = centered ? +1 : ((top != swapped) ? +1 : -1); // TL:+1  BL:-1  TR:-1  BR:+1
    int rc = IFL_EUNSUPPORTED;
#define IFL_CASE(CC, KK, NN)                                                                                               \
    if (C == CC && KH == KK && W == 16 * NN && !centered)                                                                   \
        rc = launch_wg<CC, KK, KK, NN, 0>(a, bb, partial, swapped ? amax_x : amax_gz, swapped ? amax_gz : amax_x, B, H, sg, s);
    IFL_CASE(64, 3, 2)
    IFL_CASE(64, 3, 1)
    IFL_CASE(32, 3, 2)
    IFL_CASE(32, 3, 1)
    IFL_CASE(64, 2, 2)
    IFL_CASE(64, 2, 1)
    IFL_CASE(32, 2, 2)
    IFL_CASE(32, 2, 1)
    IFL_CASE(128, 3, 2)
    IFL_CASE(128, 3, 1)
    IFL_CASE(256, 3, 2)
    IFL_CASE(256, 3, 1)
    IFL_CASE(128, 2, 2)
    IFL_CASE(128, 2, 1)
    IFL_CASE(256, 2, 2)
    IFL_CASE(256, 2, 1)
#undef IFL_CASE
#define IFL_CASE(CC, KK, NN)                                                                                               \
    if (C == CC && KH == KK && W == 16 * NN && centered)                                                                    \
        rc = launch_wg<CC, KK, KK, NN, 1>(a, bb, partial, swapped ? amax_x : amax_gz, swapped ? amax_gz : amax_x, B, H, sg, s);
    IFL_CASE(64, 3, 2)
    IFL_CASE(64, 3, 1)
    IFL_CASE(32, 3, 2)
    IFL_CASE(32, 3, 1)
#undef IFL_CASE
    if (rc == IFL_EUNSUPPORTED) IFL_FAIL(rc, "launch_wgrad_mfma: no instantiation for C=%d K=%d W=%d", C, KH, W);
    if (rc) return rc;
    const int nsplit = (wgrad_ntask(B, H) + 3) / 4;
    const size_t total4 = (size_t)KH * KW * C * C / 4;
    size_t blocks = (total4 + 63) / 64;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)blocks), dim3(256), 0, s, partial, dw, amax_gz, amax_x, nsplit, C, KH, KW,
                       swapped, top, left, scale, mask_mode, mkh, mkw, gz, x, B, H, W, pt, pl);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

} // namespace ifl
