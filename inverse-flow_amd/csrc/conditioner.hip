// The conditioner of the affine coupling (inf/layers/coupling.py:47-62: `Coupling.net` = 3x3 conv C/2 -> width, ReLU,
// 1x1 conv width -> C, ReLU, Conv2dZero C -> C = 3x3 conv + bias, times exp(3 logs) per channel: coupling.py:9-45), forward
// and backward: two launches forward, six backward.
//
// Why it is here: the three convolutions are tiny (C <= 48 channels on <= 16x16 images; 0.05-0.5 GFLOP) and a training step
// of the configs[2]/[3]/[4] models runs 33-150 of these nets.  On library convolutions one net is ~55 launches forward +
// backward (layout and precision casts, convolutions, their zero-fills and reductions, ReLUs, the gain) at ~4.5 us each
// in a graph replay: two thirds of the whole step (tools/time_stub_conditioner.py).  Here:
//
//   forward   k_cond_fwd1m  a2 = relu(W2 relu(W1 * x1))     both products on the fp32 matrix cores; the `width`-channel hidden
//                                                            activation goes from accumulators through an LDS tile
//                                                            into the second product and never leaves the CU
//             k_cond_fwd2   h  = (W3 * a2 + b3) exp(3 logs)
//   backward  k_cond_bwd1   g3 = dh gain, per-tile sums of d logs / d b3, g2 = relu'(a2) (W3^T * g3), operand matrices of dW3
//             k_cond_bwd2m  hidden activation recomputed, g1 = relu'(a1) W2^T g2 (matrix cores), operand matrices of dW1, dW2
//             k_cond_bwd3u  u = W1^T g1 per pixel and tap (matrix cores);  k_cond_bwd3g  dx1 += the nine taps of u
//             k_cond_wgrad  the three weight gradients: [rows x pixels] x [pixels x cols] products with a long reduction and
//                           a small output, from operand matrices the kernels above write pixel-major (bf16 under autocast,
//                           fp32 otherwise), a 16x16 tile and 512 pixels per wave;  k_cond_wreduce adds the slices in order
//
// Arithmetic: fp32 on the fp32 master weights (nothing is cast per step): v_mfma_f32_16x16x4_f32 is exact fp32 at the
// vector FMA rate, but its operands are fragments -- 5 LDS words per 1024 MACs.  [The first forms of these kernels used
// FMAs with wave-uniform weights: through the scalar cache every 64-byte line of a weight stream read once is an exposed
// miss (s_waitcnt lgkmcnt(0): scalar loads return out of order); broadcast from LDS, 17 words per 16 FMAs of a lane made
// them LDS-bound at 20-120 us a kernel.  The two 3x3 convolutions over C channels (forward 2, backward 1) are small enough
// to stay on that form: the reduction channels split over the eight waves of a workgroup, each wave's weights staged in a
// 4 KB LDS block, partial sums meeting in LDS slots.]  A workgroup owns 64 pixels.  The work is far from any roofline --
// what decides the time is the number of exposed memory round trips and launches: neighbourhood loads are unconditional,
// from clamped coordinates, and issued in one batch (a conditional load is a branch, and a branch per load serialises the
// round trips); the next chunk of weights is in flight while the current one is multiplied.  No float atomics: every sum
// has one owner and a fixed order (results are reproducible bit for bit).
#include "ifl_common.h"
#include "bf16_util.h"

namespace ifl {

typedef float f4 __attribute__((ext_vector_type(4)));

static constexpr int HC = 16;      // hidden units per register chunk
static constexpr int STAGE = 1024; // floats of a wave's weight stage (256 quads: four per lane)

struct CondShape {
    int B, H, W, Wd, Cx; // Cx: channels of the tensor x1 is the head of
};

__device__ __forceinline__ float wave_sum(float v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// operand matrices of the weight-gradient GEMMs: bf16 (the autocast step) or fp32 (an fp32 step keeps fp32 gradients)
__device__ __forceinline__ void put_op(bf16_t *p, float v) { *p = narrow_bf16(v); }
__device__ __forceinline__ void put_op(float *p, float v) { *p = v; }

// A block of weights: nq 16-byte quads, either contiguous or rows of four quads (16 floats) `stride` floats apart.
struct Blk {
    const float *g;
    int nq, stride; // stride 0: contiguous
};
__device__ __forceinline__ void blk_copy(float *wl, const Blk &b, int lane)
{
    f4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int qd = lane + 64 * i;
        if (qd < b.nq) r[i] = *(const f4 *)(b.g + (b.stride ? (size_t)(qd >> 2) * b.stride + (qd & 3) * 4 : (size_t)qd * 4));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int qd = lane + 64 * i;
        if (qd < b.nq) *(f4 *)(wl + qd * 4) = r[i];
    }
}

// ---- weights: transposed copies + the gain ---------------------------------------------------------------------------
// wt = [ W1t: K1 x Wd | W2t: Wd x C | W3t: 9C x C (k3 = ci*9+tap major, co minor) | W3b: 9C x C ((co*9+tap) major, ci
//        minor) | gain: C ]
__host__ __device__ inline size_t cond_wt_floats(int C, int Wd) { return (size_t)9 * (C / 2) * Wd + (size_t)Wd * C + (size_t)18 * C * C + C; }

__global__ __launch_bounds__(256) void k_cond_prep(const float *__restrict__ w1, const float *__restrict__ w2,
                                                    const float *__restrict__ w3, const float *__restrict__ logs,
                                                    float *__restrict__ wt, int C, int Wd, float logscale)
{
    const int K1 = 9 * (C / 2);
    const size_t n1 = (size_t)K1 * Wd, n2 = (size_t)Wd * C, n3 = (size_t)9 * C * C;
    const size_t total = n1 + n2 + 2 * n3 + C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v;
        if (i < n1) { // W1t[k][w] = W1[w][k]
            const int k = (int)(i / Wd), w = (int)(i % Wd);
            v = w1[(size_t)w * K1 + k];
        } else if (i < n1 + n2) { // W2t[w][c] = W2[c][w]
            const size_t j = i - n1;
            const int w = (int)(j / C), c = (int)(j % C);
            v = w2[(size_t)c * Wd + w];
        } else if (i < n1 + n2 + n3) { // W3t[ci*9+tap][co] = W3[co][ci][tap]
            const size_t j = i - n1 - n2;
            const int k3 = (int)(j / C), co = (int)(j % C);
            v = w3[(size_t)co * 9 * C + k3];
        } else if (i < n1 + n2 + 2 * n3) { // W3b[co*9+tap][ci] = W3[co][ci][tap]
            const size_t j = i - n1 - n2 - n3;
            const int q = (int)(j / C), ci = (int)(j % C), co = q / 9, tap = q % 9;
            v = w3[((size_t)co * C + ci) * 9 + tap];
        } else {
            v = expf(logs[i - n1 - n2 - 2 * n3] * logscale);
        }
        wt[i] = v;
    }
}

// the same for many couplings in one launch (blockIdx.y = job): a training step prepares the weight images of all its
// couplings up front instead of once per layer inside the forward -- 30 (configs[3]) to 144 (configs[4]) launches fewer a step
struct PrepJob { // = ifl_cond_prep_job (include/invflow.h)
    const float *w1, *w2, *w3, *logs;
    float *wt;
    int C, Wd;
    float logscale;
    int pad;
};
static_assert(sizeof(PrepJob) == 56, "ifl_cond_prep_job layout");
__global__ __launch_bounds__(256) void k_cond_prep_many(const PrepJob *__restrict__ jobs)
{
    const PrepJob j = jobs[blockIdx.y];
    const float *w1 = j.w1, *w2 = j.w2, *w3 = j.w3, *logs = j.logs;
    const int C = j.C, Wd = j.Wd, K1 = 9 * (C / 2);
    const size_t n1 = (size_t)K1 * Wd, n2 = (size_t)Wd * C, n3 = (size_t)9 * C * C;
    const size_t total = n1 + n2 + 2 * n3 + C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v; // (the cases of k_cond_prep)
        if (i < n1) {
            v = w1[(size_t)(i % Wd) * K1 + i / Wd];
        } else if (i < n1 + n2) {
            const size_t q = i - n1;
            v = w2[(size_t)(q % C) * Wd + q / C];
        } else if (i < n1 + n2 + n3) {
            const size_t q = i - n1 - n2;
            v = w3[(size_t)(q % C) * 9 * C + q / C];
        } else if (i < n1 + n2 + 2 * n3) {
            const size_t q = i - n1 - n2 - n3;
            const int r = (int)(q / C), ci = (int)(q % C), co = r / 9, tap = r % 9;
            v = w3[((size_t)co * C + ci) * 9 + tap];
        } else {
            v = expf(logs[i - n1 - n2 - 2 * n3] * j.logscale);
        }
        j.wt[i] = v;
    }
}

// pixel of this lane
struct Pix {
    int p, b, r, y, x;
    bool live;
};
__device__ __forceinline__ Pix pix_of(int p, int HW, int W, int NPX)
{
    Pix q;
    q.p = p;
    q.live = p < NPX;
    const int pp = q.live ? p : 0;
    q.b = pp / HW;
    q.r = pp - q.b * HW;
    q.y = q.r / W;
    q.x = q.r - q.y * W;
    return q;
}

// acc[c] += sum_j rows[j][c] v[j]  for a staged block of NJ rows of RL floats (broadcast reads), in segments of up to 16
// floats, the next segment's reads issued ahead of this segment's FMAs.  Unrolled (v and acc are registers) but fenced
// segment by segment: left alone, every ds_read of the block is hoisted to the top and hundreds of registers spill.
template <int RL, int NJ> __device__ __forceinline__ void rows_fma(float (&acc)[RL], const float *wl, const float (&v)[NJ])
{
    constexpr int SQ = RL <= 16 ? RL / 4 : (RL % 16 == 0 ? 4 : 3), NSEG = RL / (4 * SQ); // quads per segment, segments per row
    static_assert(RL % (4 * SQ) == 0, "rows are whole segments");
    f4 wn[SQ];
#pragma unroll
    for (int u = 0; u < SQ; ++u) wn[u] = ((const f4 *)wl)[u];
#pragma unroll
    for (int i = 0; i < NJ * NSEG; ++i) {
        const int j = i / NSEG, sg = i % NSEG;
        f4 w[SQ];
#pragma unroll
        for (int u = 0; u < SQ; ++u) w[u] = wn[u];
        if (i + 1 < NJ * NSEG) {
#pragma unroll
            for (int u = 0; u < SQ; ++u) wn[u] = ((const f4 *)wl)[(i + 1) * SQ + u];
        }
#pragma unroll
        for (int u = 0; u < SQ; ++u) {
            const int c = 4 * (sg * SQ + u);
            acc[c + 0] = fmaf(w[u][0], v[j], acc[c + 0]);
            acc[c + 1] = fmaf(w[u][1], v[j], acc[c + 1]);
            acc[c + 2] = fmaf(w[u][2], v[j], acc[c + 2]);
            acc[c + 3] = fmaf(w[u][3], v[j], acc[c + 3]);
        }
        asm volatile("" ::: "memory"); // (later reads stay below this line)
        __builtin_amdgcn_sched_barrier(0);
    }
}

// offsets of the nine neighbours (dy, dx in -1..1, or mirrored) from clamped coordinates -- always inside the image, so that
// the loads need no branch -- and the mask of the ones really inside
__device__ __forceinline__ unsigned nine_offsets(int (&off)[9], const Pix &q, const CondShape &s, int sign)
{
    unsigned ok = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = q.y + sign * (tap / 3 - 1), xx = q.x + sign * (tap % 3 - 1);
        if (q.live && yy >= 0 && yy < s.H && xx >= 0 && xx < s.W) ok |= 1u << tap;
        off[tap] = (min(max(yy, 0), s.H - 1) - q.y) * s.W + (min(max(xx, 0), s.W - 1) - q.x);
    }
    return ok;
}

// The two 3x3 convolutions over C channels (forward 2, backward 1) split their reduction channels over the JC waves of a
// workgroup; a wave's NI channels are NI blocks of nine rows of C weights.
template <int C> struct CSplit {
    static constexpr int JC = C >= 8 ? 8 : 4, NI = (C + JC - 1) / JC;
    static_assert(9 * C <= STAGE, "nine rows of C weights fit the stage");
};

// ---- forward 2: h = (W3 * a2 + b3) gain --------------------------------------------------------------------------------
// LDS: slots [JC][C][64] | stage [JC][STAGE]
template <int C>
__global__ __launch_bounds__(512) void k_cond_fwd2(const float *__restrict__ a2, const float *__restrict__ wt,
                                                   const float *__restrict__ b3, float *__restrict__ h, CondShape s)
{
    constexpr int K1 = 9 * (C / 2), JC = CSplit<C>::JC, NI = CSplit<C>::NI;
    extern __shared__ float lds[];
    float *red = lds;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *wl = lds + JC * C * 64 + wv * STAGE;
    const int HW = s.H * s.W, NPX = s.B * HW;
    const Pix q = pix_of(blockIdx.x * 64 + lane, HW, s.W, NPX);
    const float *__restrict__ w3t = wt + (size_t)K1 * s.Wd + (size_t)s.Wd * C;
    const float *__restrict__ gain = w3t + (size_t)18 * C * C;
    // (a wave whose channel index runs past C works on channel C-1 with zero inputs: no branches)
    auto blk = [&](int it) { return Blk{w3t + (size_t)min(wv + it * JC, C - 1) * 9 * C, 9 * C / 4, 0}; };
    int off[9];
    const unsigned ok = nine_offsets(off, q, s, 1);
    const float *__restrict__ ab = a2 + (size_t)q.b * C * HW + q.r;
    float pv[NI][9];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int ci = wv + it * JC, cc = min(ci, C - 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float v = ab[(ptrdiff_t)cc * HW + off[tap]];
            pv[it][tap] = (ci < C && ((ok >> tap) & 1)) ? v : 0.f;
        }
    }
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        __syncthreads();
        blk_copy(wl, blk(it), lane);
        __syncthreads();
        rows_fma<C, 9>(acc, wl, pv[it]);
    }
#pragma unroll
    for (int co = 0; co < C; ++co) red[(wv * C + co) * 64 + lane] = acc[co];
    __syncthreads();
    if (q.live) {
        const bool has_b = b3 != nullptr;
        for (int co = wv; co < C; co += JC) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < JC; ++j) t += red[(j * C + co) * 64 + lane];
            h[((size_t)q.b * C + co) * HW + q.r] = (t + (has_b ? b3[co] : 0.f)) * gain[co];
        }
    }
}

// ---- backward 1 -----------------------------------------------------------------------------------------------------------
// g3 = dh gain;  d logs[c] = logscale sum h dh;  d b3[c] = sum g3 (per tile of 64 pixels: part[tile][2C], summed by the
// caller);  g2 = [a2 > 0] (W3^T * g3)
// operand matrices (16-bit, [row][NPXp], NPXp = pixels rounded up to 64, the padding written as zeros):
//   g3t [C][NPXp], p3t [9C][NPXp] (the 3x3 neighbourhoods of a2: dW3 = g3t p3t^T), g2t [C][NPXp]
// LDS as forward 2.
template <int C, class T>
__global__ __launch_bounds__(512) void k_cond_bwd1(const float *__restrict__ dh, const float *__restrict__ h, const float *__restrict__ a2,
                                                   const float *__restrict__ wt, T *__restrict__ g3t, T *__restrict__ p3t,
                                                   T *__restrict__ g2t, float *__restrict__ part, CondShape s, int NPXp,
                                                   float logscale)
{
    constexpr int K1 = 9 * (C / 2), JC = CSplit<C>::JC, NI = CSplit<C>::NI;
    extern __shared__ float lds[];
    float *red = lds;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *wl = lds + JC * C * 64 + wv * STAGE;
    const int HW = s.H * s.W, NPX = s.B * HW;
    const int p = blockIdx.x * 64 + lane;
    const Pix q = pix_of(p, HW, s.W, NPX);
    const float *__restrict__ w3b = wt + (size_t)K1 * s.Wd + (size_t)s.Wd * C + (size_t)9 * C * C;
    const float *__restrict__ gain = w3b + (size_t)9 * C * C;
    auto blk = [&](int it) { return Blk{w3b + (size_t)min(wv + it * JC, C - 1) * 9 * C, 9 * C / 4, 0}; };
    int offf[9], offb[9]; // forward taps (y+dy-1) and mirrored taps (y-dy+1)
    const unsigned okf = nine_offsets(offf, q, s, 1), okb = nine_offsets(offb, q, s, -1);
    const size_t base = (size_t)q.b * C * HW + q.r;
    // this wave's channels: c = wv, wv + JC, ... (past C: channel C-1 read, zero used, nothing stored)
    // the neighbourhoods of a2 go straight out
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int c = wv + it * JC, cc = min(c, C - 1);
        float pa[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) pa[tap] = a2[base + (ptrdiff_t)cc * HW + offf[tap]];
        if (c < C) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) put_op(p3t + (size_t)(c * 9 + tap) * NPXp + p, (okf >> tap) & 1 ? pa[tap] : 0.f);
        }
    }
    float a0[NI], gv[NI][9];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int c = wv + it * JC, cc = min(c, C - 1);
        const bool on = q.live && c < C;
        const float gn = gain[cc];
        const float d0v = dh[base + (size_t)cc * HW], h0v = h[base + (size_t)cc * HW], a0v = a2[base + (size_t)cc * HW];
        const float d0 = on ? d0v : 0.f, h0 = on ? h0v : 0.f;
        a0[it] = on ? a0v : 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float v = dh[base + (ptrdiff_t)cc * HW + offb[tap]];
            gv[it][tap] = (c < C && ((okb >> tap) & 1)) ? v * gn : 0.f;
        }
        const float g = d0 * gn;
        const float sl = wave_sum(logscale * h0 * d0), sb = wave_sum(g);
        if (c < C) {
            put_op(g3t + (size_t)c * NPXp + p, g);
            if (lane == 0) {
                part[(size_t)blockIdx.x * 2 * C + c] = sl;
                part[(size_t)blockIdx.x * 2 * C + C + c] = sb;
            }
        }
    }
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        __syncthreads();
        blk_copy(wl, blk(it), lane);
        __syncthreads();
        rows_fma<C, 9>(acc, wl, gv[it]);
    }
#pragma unroll
    for (int ci = 0; ci < C; ++ci) red[(wv * C + ci) * 64 + lane] = acc[ci];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int ci = wv + it * JC;
        if (ci < C) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < JC; ++j) t += red[(j * C + ci) * 64 + lane];
            put_op(g2t + (size_t)ci * NPXp + p, a0[it] > 0.f ? t : 0.f);
        }
    }
}

// ---- the hidden layer on the matrix cores (forward 1 and backward 2) -------------------------------------------------------
// a1[hid][px] = sum_k W1[hid][k] patch[k][px] is a [width x 9C/2] x [9C/2 x 64] product per workgroup: with broadcast weights
// from LDS the FMA form moves 17 LDS words per 16 FMAs of a lane and is LDS-bound (forward 1 at 18-56 us); as fragments of
// v_mfma_f32_16x16x4_f32 (exact fp32) it moves 5 words per 1024 MACs.  The workgroup walks the hidden units in chunks of
// 128 -- one 16-unit tile per wave -- and the k index in chunks of 32: W1 rows [128][32] staged in LDS (the next chunk's
// loads in flight), the patch [k][64] staged once.  The second product takes the wave's own 16 x 64 tile of relu(a1) -- out
// through a wave-private LDS tile, back as the right-hand fragment -- against W2's 16 columns.
template <int C> struct HCfg {
    static constexpr int CIN = C / 2, K1 = 9 * CIN, K1P = (K1 + 3) / 4 * 4, KC = K1P > 64 ? 64 : 32, NKC = (K1P + KC - 1) / KC; // (64-wide chunks of W1 from C = 16: half the barrier-separated stages; configs[4] step 31.7 -> 31.4 ms.  From C = 8 on: configs[2] 5.86 -> 6.03 ms)
    static constexpr int PST = 64 + 16, AST = KC + 1, HCH = 128;            // LDS row strides; hidden units per chunk
    static constexpr int P_FL = NKC * KC * PST, A_FL = HCH * AST, T_FL = 8 * 16 * PST; // patch (k padded to whole chunks), W1 chunk, tiles
    static constexpr int CP = (C + 15) / 16 * 16, MT2 = CP / 16;
};

// stage the 3x3 neighbourhoods of the tile as patch[k][PST] (k padded with zero rows)
template <int C>
__device__ __forceinline__ void stage_patch_m(float *Ps, const float *__restrict__ x, const Pix &q, const CondShape &s, int lane, int wv)
{
    using Hc = HCfg<C>;
    const int HW = s.H * s.W;
#pragma unroll 4 // (sixteen in flight measured slower: configs[3] 5.41 -> 5.50 ms)
    for (int k = wv; k < Hc::NKC * Hc::KC; k += 8) {
        const int kk = k < Hc::K1 ? k : 0, ci = kk / 9, tap = kk - ci * 9, dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int yy = q.y + dy, xx = q.x + dx;
        const bool in = k < Hc::K1 && q.live && yy >= 0 && yy < s.H && xx >= 0 && xx < s.W;
        const int yc = min(max(yy, 0), s.H - 1), xc = min(max(xx, 0), s.W - 1);
        const float v = x[((size_t)q.b * s.Cx + ci) * HW + yc * s.W + xc];
        Ps[k * Hc::PST + lane] = in ? v : 0.f;
    }
}

// one chunk of 128 hidden units: acc[n] (n: four 16-pixel column tiles) of this wave's 16-unit tile.  w1: [Wd][K1].
template <int C>
__device__ __forceinline__ void hidden_tile_mfma(f4 (&acc)[4], const float *__restrict__ w1, const float *Ps, float *As, int h0, int Wd,
                                                 int tid, int lane, int wv)
{
    using Hc = HCfg<C>;
    constexpr int NA = Hc::HCH * Hc::KC / 512; // 8 elements per thread and chunk
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = f4{0.f, 0.f, 0.f, 0.f};
    float ra[NA];
    auto load = [&](int kc) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + 512 * i, hid = e / Hc::KC, k = kc + e % Hc::KC;
            const int hc = min(h0 + hid, Wd - 1), kcl = min(k, Hc::K1 - 1);
            const float v = w1[(size_t)hc * Hc::K1 + kcl];
            ra[i] = (h0 + hid < Wd && k < Hc::K1) ? v : 0.f;
        }
    };
    load(0);
#pragma unroll 1
    for (int c = 0; c < Hc::NKC; ++c) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + 512 * i;
            As[(e / Hc::KC) * Hc::AST + e % Hc::KC] = ra[i];
        }
        __syncthreads();
        if (c + 1 < Hc::NKC) load((c + 1) * Hc::KC);
        const int ksteps = min(Hc::KC, Hc::K1P - c * Hc::KC) / 4;
        for (int st = 0; st < ksteps; ++st) {
            const int kk = 4 * st + (lane >> 4);
            const float af = As[(16 * wv + (lane & 15)) * Hc::AST + kk];
            const float *pr = Ps + (size_t)(c * Hc::KC + kk) * Hc::PST + (lane & 15);
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, pr[16 * n], acc[n], 0, 0, 0);
        }
    }
}

// ---- forward 1 (matrix cores): a2 = relu(W2 relu(W1 * x1)) ------------------------------------------------------------------
// LDS: patch | W1 chunk | eight 16 x 64 tiles; the slots [8][CP][64] of the final sum reuse it from the start.
template <int C>
__global__ __launch_bounds__(512) void k_cond_fwd1m(const float *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ w2,
                                                    float *__restrict__ a2, CondShape s)
{
    using Hc = HCfg<C>;
    extern __shared__ float lds[];
    float *Ps = lds, *As = lds + Hc::P_FL, *Ts = As + Hc::A_FL;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *mine = Ts + wv * 16 * Hc::PST;
    const int HW = s.H * s.W, NPX = s.B * HW;
    const Pix q = pix_of(blockIdx.x * 64 + lane, HW, s.W, NPX);
    stage_patch_m<C>(Ps, x, q, s, lane, wv);
    f4 acc2[Hc::MT2][4];
#pragma unroll
    for (int m = 0; m < Hc::MT2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc2[m][n] = f4{0.f, 0.f, 0.f, 0.f};
    for (int h0 = 0; h0 < s.Wd; h0 += Hc::HCH) {
        const int hb = h0 + 16 * wv; // this wave's 16 hidden units (past Wd: zero weights)
        // W2 fragments of this tile: A[m = c][kk = hidden]  (w2: [C][Wd])
        float w2f[Hc::MT2][4];
#pragma unroll
        for (int m = 0; m < Hc::MT2; ++m)
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int c = 16 * m + (lane & 15), hh = hb + 4 * st + (lane >> 4);
                const float v = w2[(size_t)min(c, C - 1) * s.Wd + min(hh, s.Wd - 1)];
                w2f[m][st] = (c < C && hh < s.Wd) ? v : 0.f;
            }
        f4 acc[4];
        hidden_tile_mfma<C>(acc, w1, Ps, As, h0, s.Wd, tid, lane, wv);
        // relu(a1) tile -> LDS [hidden 16][PST] -> right-hand fragments
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(4 * (lane >> 4) + r) * Hc::PST + 16 * n + (lane & 15)] = fmaxf(acc[n][r], 0.f);
        __syncthreads();
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const float *br = mine + (4 * st + (lane >> 4)) * Hc::PST + (lane & 15);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float bf = br[16 * n];
#pragma unroll
                for (int m = 0; m < Hc::MT2; ++m) acc2[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[m][st], bf, acc2[m][n], 0, 0, 0);
            }
        }
    }
    // sum over the eight waves (slots, wave order), ReLU, store
    __syncthreads();
    f4 *slot = (f4 *)lds; // [wave][m][n][lane]
#pragma unroll
    for (int m = 0; m < Hc::MT2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) slot[((wv * Hc::MT2 + m) * 4 + n) * 64 + lane] = acc2[m][n];
    __syncthreads();
    for (int t = wv; t < Hc::MT2 * 4; t += 8) { // (m, n) tiles over the waves
        const int m = t / 4, n = t % 4;
        f4 v = slot[((0 * Hc::MT2 + m) * 4 + n) * 64 + lane];
        for (int g = 1; g < 8; ++g) v += slot[((g * Hc::MT2 + m) * 4 + n) * 64 + lane];
        const Pix qq = pix_of(blockIdx.x * 64 + 16 * n + (lane & 15), HW, s.W, NPX);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * m + 4 * (lane >> 4) + r;
            if (qq.live && c < C) a2[((size_t)qq.b * C + c) * HW + qq.r] = fmaxf(v[r], 0.f);
        }
    }
}

// ---- backward 2 (matrix cores) ------------------------------------------------------------------------------------------------
// a1 recomputed as above; ga[hid][px] = sum_c W2[c][hid] g2[c][px] the same way (left fragments from W2t [Wd][C], right
// fragments from g2 staged [c][PST]); g1 = [a1 > 0] ga.  a1t, g1t rows leave in 16-pixel segments; p1t from the staged patch.
template <int C, class T>
__global__ __launch_bounds__(512) void k_cond_bwd2m(const float *__restrict__ x, const T *__restrict__ g2t, const float *__restrict__ w1,
                                                    const float *__restrict__ w2t, T *__restrict__ a1t, T *__restrict__ g1t,
                                                    T *__restrict__ p1t, CondShape s, int NPXp)
{
    using Hc = HCfg<C>;
    constexpr int CP4 = (C + 3) / 4 * 4;
    extern __shared__ float lds[];
    float *Ps = lds, *As = lds + Hc::P_FL, *Gs = As + Hc::A_FL; // Gs: g2 [CP4][PST]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = s.H * s.W, NPX = s.B * HW, p0 = blockIdx.x * 64;
    const Pix q = pix_of(p0 + lane, HW, s.W, NPX);
    stage_patch_m<C>(Ps, x, q, s, lane, wv);
    for (int c = wv; c < CP4; c += 8) Gs[c * Hc::PST + lane] = c < C ? widen(g2t[(size_t)c * NPXp + p0 + lane]) : 0.f;
    __syncthreads();
    for (int k = wv; k < Hc::K1; k += 8) put_op(p1t + (size_t)k * NPXp + p0 + lane, Ps[k * Hc::PST + lane]);
    for (int h0 = 0; h0 < s.Wd; h0 += Hc::HCH) {
        const int hb = h0 + 16 * wv;
        float w2f[CP4 / 4]; // A[m = hidden][kk = c] = W2t[hidden][c]
#pragma unroll
        for (int st = 0; st < CP4 / 4; ++st) {
            const int hh = hb + (lane & 15), c = 4 * st + (lane >> 4);
            const float v = w2t[(size_t)min(hh, s.Wd - 1) * C + min(c, C - 1)];
            w2f[st] = (hh < s.Wd && c < C) ? v : 0.f;
        }
        f4 acc[4], ga[4];
        hidden_tile_mfma<C>(acc, w1, Ps, As, h0, s.Wd, tid, lane, wv);
#pragma unroll
        for (int n = 0; n < 4; ++n) ga[n] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < CP4 / 4; ++st) {
            const float *br = Gs + (4 * st + (lane >> 4)) * Hc::PST + (lane & 15);
#pragma unroll
            for (int n = 0; n < 4; ++n) ga[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[st], br[16 * n], ga[n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hh = hb + 4 * (lane >> 4) + r;
                if (hh < s.Wd) {
                    const size_t o = (size_t)hh * NPXp + p0 + 16 * n + (lane & 15);
                    put_op(a1t + o, fmaxf(acc[n][r], 0.f));
                    put_op(g1t + o, acc[n][r] > 0.f ? ga[n][r] : 0.f);
                }
            }
    }
}

// ---- backward 3: dx[:, :C/2] += W1^T * g1 (3x3, transposed) ------------------------------------------------------------------
// Two steps.  (a) u[k][p] = sum_w W1[w][k] g1[w][p], k = ci*9 + tap: what pixel p sends to its neighbour at `tap` -- a
// [9C/2 x width] x [width x pixels] product on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, the vector FMA
// rate, but operands come as fragments: 2 LDS words per 1024 MACs where the broadcast-weight FMA form reads 17 -- that
// form was LDS-bound at 40-120 us).  A workgroup owns 64 pixels and all of k; width goes through LDS in chunks of 32
// hidden units (W1t rows and g1t rows, loaded for the next chunk while this one is multiplied); its eight waves split
// the k tiles, and the hidden units of a chunk where there are fewer than eight k tiles.  (b) dx1[ci][p] = sum_tap
// u[ci*9+tap][p - offset(tap)]: nine reads per element.
template <int C> struct UCfg {
    static constexpr int K1 = 9 * (C / 2), MT = (K1 + 15) / 16, K1P = MT * 16;
    static constexpr int MS = MT >= 8 ? 8 : (MT >= 4 ? 4 : (MT >= 2 ? 2 : 1)), KS = 8 / MS; // waves = MS x KS
    static constexpr int MW = (MT + MS - 1) / MS;                                          // k tiles of a wave
    static constexpr int WCH = K1P <= 32 ? 128 : (K1P <= 64 ? 64 : 32); // hidden units per chunk (few k rows: longer chunks,
                                                                        // fewer exposed round trips)
    static constexpr int AST = WCH + 1, BST = 64 + 16;                  // padded LDS row strides
    static constexpr int A_FL = K1P * AST, B_FL = WCH * BST;
    static constexpr int NA = (K1P * WCH + 511) / 512, NBL = WCH * 64 / 512;               // elements per thread and chunk
    static constexpr int SLOT_FL = KS > 1 ? 8 * MW * 4 * 256 : 0, LDS_FL = A_FL + B_FL > SLOT_FL ? A_FL + B_FL : SLOT_FL;
};

template <int C, class T>
__global__ __launch_bounds__(512) void k_cond_bwd3u(const T *__restrict__ g1t, const float *__restrict__ wt, float *__restrict__ u,
                                                    int Wd, int NPXp)
{
    using U = UCfg<C>;
    constexpr int K1 = U::K1, MS = U::MS, KS = U::KS, MW = U::MW, WCH = U::WCH, AST = U::AST, BST = U::BST;
    extern __shared__ float lds[];
    float *As = lds, *Bs = lds + U::A_FL;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ms = wv % MS, ks = wv / MS;
    const int p0 = blockIdx.x * 64;
    const float *__restrict__ w1t = wt; // [K1][Wd]
    float ra[U::NA], rb[U::NBL];
    auto load = [&](int w0) {
#pragma unroll
        for (int i = 0; i < U::NA; ++i) { // A chunk: rows k (zero past K1), 32 consecutive w
            const int e = tid + 512 * i, k = e / WCH, w = e % WCH;
            const int kc = k < K1 ? k : K1 - 1, wc = w0 + w < Wd ? w0 + w : Wd - 1; // (width is a multiple of 16, not of 32)
            const float v = w1t[(size_t)kc * Wd + wc];
            ra[i] = (e < U::K1P * WCH && k < K1 && w0 + w < Wd) ? v : 0.f;
        }
#pragma unroll
        for (int i = 0; i < U::NBL; ++i) { // B chunk: 32 rows of g1t, this tile's 64 pixels
            const int e = tid + 512 * i, w = e / 64, px = e % 64;
            const float v = widen(g1t[(size_t)(w0 + w < Wd ? w0 + w : Wd - 1) * NPXp + p0 + px]);
            rb[i] = w0 + w < Wd ? v : 0.f;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < U::NA; ++i) {
            const int e = tid + 512 * i, k = e / WCH, w = e % WCH;
            if (e < U::K1P * WCH) As[k * AST + w] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < U::NBL; ++i) {
            const int e = tid + 512 * i, w = e / 64, px = e % 64;
            Bs[w * BST + px] = rb[i];
        }
    };
    f4 acc[MW][4];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f4{0.f, 0.f, 0.f, 0.f};
    load(0);
    for (int w0 = 0; w0 < Wd; w0 += WCH) {
        __syncthreads();
        store();
        __syncthreads();
        if (w0 + WCH < Wd) load(w0 + WCH);
#pragma unroll
        for (int st = 0; st < WCH / 4 / KS; ++st) {
            const int kk = 4 * (ks + KS * st) + (lane >> 4); // hidden unit of this lane's fragment element
            float bf[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) bf[n] = Bs[kk * BST + 16 * n + (lane & 15)];
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const int mt = ms + MS * m;
                if (mt < U::MT) {
                    const float af = As[(16 * mt + (lane & 15)) * AST + kk];
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[n], acc[m][n], 0, 0, 0);
                }
            }
        }
    }
    // sum over the KS hidden-unit groups (LDS slots, group order), then u[k][p]
    if (KS > 1) {
        __syncthreads();
        f4 *slot = (f4 *)lds; // [ks][ms-local tile m][n][lane]
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) slot[((wv * MW + m) * 4 + n) * 64 + lane] = acc[m][n];
        __syncthreads();
        if (ks == 0) {
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    f4 t = acc[m][n];
                    for (int g = 1; g < KS; ++g) t += slot[(((g * MS + ms) * MW + m) * 4 + n) * 64 + lane];
                    acc[m][n] = t;
                }
        }
    }
    if (ks == 0) {
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int mt = ms + MS * m;
            if (mt < U::MT) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 16 * mt + 4 * (lane >> 4) + r;
                        if (k < K1) u[(size_t)k * NPXp + p0 + 16 * n + (lane & 15)] = acc[m][n][r];
                    }
            }
        }
    }
}

template <int C>
__global__ __launch_bounds__(256) void k_cond_bwd3g(const float *__restrict__ u, float *__restrict__ dx, CondShape s, int NPXp)
{
    constexpr int CIN = C / 2;
    const int HW = s.H * s.W, NPX = s.B * HW;
    const int lane = threadIdx.x & 63, ci0 = threadIdx.x >> 6;
    const Pix q = pix_of(blockIdx.x * 64 + lane, HW, s.W, NPX);
    int off[9];
    const unsigned okb = nine_offsets(off, q, s, -1); // the pixel whose tap lands here: p - (dy-1) W - (dx-1)
    for (int ci = ci0; ci < CIN; ci += 4) {
        float t = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float v = u[(size_t)(ci * 9 + tap) * NPXp + (q.live ? q.p : 0) + off[tap]];
            t += (okb >> tap) & 1 ? v : 0.f;
        }
        if (q.live) dx[((size_t)q.b * s.Cx + ci) * HW + q.r] += t;
    }
}

// ---- the three weight gradients: out[m][n] = sum_p A[m][p] B[n][p] -----------------------------------------------------------
// dW1 = g1t p1t^T [Wd x 9C/2], dW2 = g2t a1t^T [C x Wd], dW3 = g3t p3t^T [C x 9C]: reductions over all P pixels with small
// outputs.  (Library GEMMs take 8-50 us EACH for these shapes -- a third of the configs[2] step.)  One launch for all
// three: a wave owns a 16x16 output tile and a slice of 512 pixels -- bf16 operands on v_mfma_f32_16x16x16_bf16, fp32
// operands on v_mfma_f32_16x16x4_f32, a lane's fragment 8 / 4 bytes straight from the row-major operand matrices (both
// operands are [row][pixel]: the same access pattern) -- and writes its partial tile; k_cond_wreduce adds the slices in
// order (and the per-tile sums of d logs / d b3): deterministic.
struct WJob {
    const void *A, *B;
    int M, N, nt, blk0; // nt: tiles along N; blk0: first (tile) index of this job in the grid
    size_t out;         // offset of the job's output in a slice
};
struct WJobs {
    WJob j[3];
};
static constexpr int WSL = 512; // pixels per slice

typedef short short4_ __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 wfrag_mma(const bf16_t *ar, const bf16_t *br, bool aon, bool bon, int p0, int p1, int lane)
{
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
    // (the reduction index may be permuted as long as both operands agree: within a chunk of 64 pixels a lane takes 16
    // CONSECUTIVE pixels -- 32 / 64 contiguous bytes per lane instead of 8 / 4 bytes at four / sixteen places)
    const int q16 = 16 * (lane >> 4);
    for (int p = p0; p < p1; p += 64) { // (P is a multiple of 64)
        short4_ a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = *(const short4_ *)(ar + p + q16 + 4 * i);
            b[i] = *(const short4_ *)(br + p + q16 + 4 * i);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!aon) a[i] = short4_{0, 0, 0, 0};
            if (!bon) b[i] = short4_{0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[i], b[i], acc, 0, 0, 0);
        }
    }
    return acc;
}
__device__ __forceinline__ f4 wfrag_mma(const float *ar, const float *br, bool aon, bool bon, int p0, int p1, int lane)
{
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
    const int q16 = 16 * (lane >> 4);
    for (int p = p0; p < p1; p += 64) {
        float a[16], b[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            a[i] = ar[p + q16 + i];
            b[i] = br[p + q16 + i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aon ? a[i] : 0.f, bon ? b[i] : 0.f, acc, 0, 0, 0);
    }
    return acc;
}

template <class T>
__global__ __launch_bounds__(64) void k_cond_wgrad(WJobs jobs, float *__restrict__ gpart, int P, int S, size_t per_slice)
{
    const int lane = threadIdx.x, sl = blockIdx.x % S, tile = blockIdx.x / S;
    const int ji = tile >= jobs.j[2].blk0 ? 2 : (tile >= jobs.j[1].blk0 ? 1 : 0);
    const WJob jb = jobs.j[ji];
    const int t = tile - jb.blk0, mt = t / jb.nt, nt = t % jb.nt;
    const int am = 16 * mt + (lane & 15), bn = 16 * nt + (lane & 15);
    const T *ar = (const T *)jb.A + (size_t)min(am, jb.M - 1) * P, *br = (const T *)jb.B + (size_t)min(bn, jb.N - 1) * P;
    const int p0 = sl * WSL, p1 = min(P, p0 + WSL);
    const f4 acc = wfrag_mma(ar, br, am < jb.M, bn < jb.N, p0, p1, lane);
    float *o = gpart + (size_t)sl * per_slice + jb.out;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = 16 * mt + 4 * (lane >> 4) + r, n = 16 * nt + (lane & 15);
        if (m < jb.M && n < jb.N) o[(size_t)m * jb.N + n] = acc[r];
    }
}

// out[i] = sum_s gpart[s][i] (i < per_slice: dW1 | dW2 | dW3), out[per_slice + c] = sum_tiles tpart[tile][c] (c < 2C: d logs | d b3).
// Eight lanes per output: lane k adds the slices k, k + 8, ... in order, the eight partial sums meet in a fixed shuffle tree
// (with one thread per output the launch is a few dozen workgroups each walking 40 dependent loads: 15 us at the configs[2]
// shapes).
__global__ __launch_bounds__(256) void k_cond_wreduce(const float *__restrict__ gpart, const float *__restrict__ tpart,
                                                      float *__restrict__ out, size_t per_slice, int S, int tiles, int C2)
{
    const size_t gid = blockIdx.x * (size_t)256 + threadIdx.x, i = gid >> 3;
    const int part = (int)(gid & 7);
    float t = 0.f;
    if (i < per_slice) {
        for (int sl = part; sl < S; sl += 32) { // (up to four loads in flight)
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = sl + 8 * k < S ? gpart[(size_t)(sl + 8 * k) * per_slice + i] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) t += v[k];
        }
    } else if (i < per_slice + C2) {
        const int c = (int)(i - per_slice);
        for (int k0 = part; k0 < tiles; k0 += 32) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = k0 + 8 * k < tiles ? tpart[(size_t)(k0 + 8 * k) * C2 + c] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) t += v[k];
        }
    }
    t += __shfl_down(t, 4, 64);
    t += __shfl_down(t, 2, 64);
    t += __shfl_down(t, 1, 64);
    if (part == 0 && i < per_slice + C2) out[i] = t;
}

// workspace of the backward (byte offsets, 256-aligned): operand matrices | u | per-tile sums | per-slice partial gradients
struct CondWs {
    size_t ops, u, tpart, gpart, per_slice, total;
};
static CondWs cond_ws(int C, int Wd, int P, size_t elem)
{
    CondWs w;
    const size_t K1 = 9 * (size_t)(C / 2), S = ((size_t)P + WSL - 1) / WSL;
    w.per_slice = (size_t)Wd * K1 + (size_t)C * Wd + (size_t)9 * C * C;
    w.ops = 0;
    w.u = align_up(w.ops + ((size_t)11 * C + 2 * (size_t)Wd + K1) * P * elem, 256);
    w.tpart = align_up(w.u + K1 * P * sizeof(float), 256);
    w.gpart = align_up(w.tpart + (size_t)(P / 64) * 2 * C * sizeof(float), 256);
    w.total = align_up(w.gpart + S * w.per_slice * sizeof(float), 256);
    return w;
}

template <int C> struct CondLaunch {
    static constexpr int K1 = 9 * (C / 2);
    static int forward(const float *x, const float *wt, const float *w1, const float *w2, const float *b3, float *a2, float *h,
                       CondShape s, hipStream_t st)
    {
        static LdsOptIn opt2;
        const int NPX = s.B * s.H * s.W, tiles = (NPX + 63) / 64;
        constexpr int JC = CSplit<C>::JC, ldsc = (JC * C * 64 + JC * STAGE) * (int)sizeof(float);
        if (int rc = lds_opt_in(opt2, (const void *)k_cond_fwd2<C>, 160 * 1024 - 256)) return rc;
        {
            using Hc = HCfg<C>;
            static LdsOptIn optm;
            constexpr int a_m = Hc::P_FL + Hc::A_FL + Hc::T_FL, b_m = 8 * Hc::MT2 * 4 * 64 * 4;
            constexpr int ldsm = (a_m > b_m ? a_m : b_m) * (int)sizeof(float);
            static_assert(ldsm <= 160 * 1024 - 256, "forward 1 fits the LDS");
            if (int rc = lds_opt_in(optm, (const void *)k_cond_fwd1m<C>, 160 * 1024 - 256)) return rc;
            hipLaunchKernelGGL(k_cond_fwd1m<C>, dim3(tiles), dim3(512), ldsm, st, x, w1, w2, a2, s);
        }
        hipLaunchKernelGGL(k_cond_fwd2<C>, dim3(tiles), dim3(64 * JC), ldsc, st, (const float *)a2, wt, b3, h, s);
        IFL_HIP(hipGetLastError());
        return IFL_OK;
    }
    template <class T>
    static int backward(const float *x, const float *dh, const float *h, const float *a2, const float *wt, const float *w1,
                        void *ws, float *grads, float *dx, CondShape s, float logscale, hipStream_t st)
    {
        static LdsOptIn opt1, opt3;
        const int NPX = s.B * s.H * s.W, tiles = (NPX + 63) / 64, P = tiles * 64;
        const CondWs w = cond_ws(C, s.Wd, P, sizeof(T));
        char *b = (char *)ws;
        T *g3t = (T *)(b + w.ops), *p3t = g3t + (size_t)C * P, *g2t = p3t + (size_t)9 * C * P, *a1t = g2t + (size_t)C * P;
        T *g1t = a1t + (size_t)s.Wd * P, *p1t = g1t + (size_t)s.Wd * P;
        float *u = (float *)(b + w.u), *tpart = (float *)(b + w.tpart), *gpart = (float *)(b + w.gpart);
        constexpr int JC = CSplit<C>::JC, ldsc = (JC * C * 64 + JC * STAGE) * (int)sizeof(float);
        constexpr int lds3 = UCfg<C>::LDS_FL * (int)sizeof(float);
        if (int rc = lds_opt_in(opt1, (const void *)k_cond_bwd1<C, T>, 160 * 1024 - 256)) return rc;
        if (int rc = lds_opt_in(opt3, (const void *)k_cond_bwd3u<C, T>, 160 * 1024 - 256)) return rc;
        hipLaunchKernelGGL((k_cond_bwd1<C, T>), dim3(tiles), dim3(64 * JC), ldsc, st, dh, h, a2, wt, g3t, p3t, g2t, tpart, s, P, logscale);
        {
            using Hc = HCfg<C>;
            static LdsOptIn optm;
            constexpr int ldsm = (Hc::P_FL + Hc::A_FL + (C + 3) / 4 * 4 * Hc::PST) * (int)sizeof(float);
            static_assert(ldsm <= 160 * 1024 - 256, "backward 2 fits the LDS");
            if (int rc = lds_opt_in(optm, (const void *)k_cond_bwd2m<C, T>, 160 * 1024 - 256)) return rc;
            hipLaunchKernelGGL((k_cond_bwd2m<C, T>), dim3(tiles), dim3(512), ldsm, st, x, (const T *)g2t, w1, wt + (size_t)K1 * s.Wd, a1t,
                               g1t, p1t, s, P);
        }
        hipLaunchKernelGGL((k_cond_bwd3u<C, T>), dim3(tiles), dim3(512), lds3, st, (const T *)g1t, wt, u, s.Wd, P);
        hipLaunchKernelGGL(k_cond_bwd3g<C>, dim3(tiles), dim3(256), 0, st, (const float *)u, dx, s, P);
        // the three weight gradients
        WJobs jobs;
        const int nt1 = (K1 + 15) / 16, nt2 = s.Wd / 16, nt3 = (9 * C + 15) / 16, mt1 = s.Wd / 16, mt23 = (C + 15) / 16;
        jobs.j[0] = WJob{g1t, p1t, s.Wd, K1, nt1, 0, 0};
        jobs.j[1] = WJob{g2t, a1t, C, s.Wd, nt2, mt1 * nt1, (size_t)s.Wd * K1};
        jobs.j[2] = WJob{g3t, p3t, C, 9 * C, nt3, mt1 * nt1 + mt23 * nt2, (size_t)s.Wd * K1 + (size_t)C * s.Wd};
        const int ntiles = mt1 * nt1 + mt23 * nt2 + mt23 * nt3, S = (P + WSL - 1) / WSL;
        hipLaunchKernelGGL(k_cond_wgrad<T>, dim3(ntiles * S), dim3(64), 0, st, jobs, gpart, P, S, w.per_slice);
        hipLaunchKernelGGL(k_cond_wreduce, dim3((unsigned)((8 * (w.per_slice + 2 * C) + 255) / 256)), dim3(256), 0, st, (const float *)gpart,
                           (const float *)tpart, grads, w.per_slice, S, tiles, 2 * C);
        IFL_HIP(hipGetLastError());
        return IFL_OK;
    }
};

static bool cond_c_ok(int C) { return C == 4 || C == 8 || C == 12 || C == 16 || C == 24 || C == 32 || C == 48; }

static int cond_check(const char *who, int B, int C, int H, int W, int Wd, int Cx)
{
    if (B < 0 || H < 1 || W < 1) IFL_FAIL(IFL_EINVAL, "%s: bad shape B=%d H=%d W=%d", who, B, H, W);
    if (!cond_c_ok(C)) IFL_FAIL(IFL_EUNSUPPORTED, "%s: C=%d is not one of 4, 8, 12, 16, 24, 32, 48", who, C);
    if (Wd < HC || Wd % HC) IFL_FAIL(IFL_EUNSUPPORTED, "%s: width %d must be a multiple of %d", who, Wd, HC);
    if (Cx < C / 2) IFL_FAIL(IFL_EINVAL, "%s: the input has %d channels, the conditioner reads %d", who, Cx, C / 2);
    if ((size_t)B * H * W > (size_t)1 << 30 || W > 4096) IFL_FAIL(IFL_EUNSUPPORTED, "%s: too many pixels", who);
    return IFL_OK;
}

#define IFL_COND_DISPATCH(C, CALL)      \
    switch (C) {                        \
    case 4: return CondLaunch<4>::CALL; \
    case 8: return CondLaunch<8>::CALL; \
    case 12: return CondLaunch<12>::CALL; \
    case 16: return CondLaunch<16>::CALL; \
    case 24: return CondLaunch<24>::CALL; \
    case 32: return CondLaunch<32>::CALL; \
    default: return CondLaunch<48>::CALL; \
    }

} // namespace ifl

using namespace ifl;

extern "C" {

int ifl_cond_supported(int C, int width) { return cond_c_ok(C) && width >= HC && width % HC == 0; }

size_t ifl_cond_weights_floats(int C, int width) { return cond_wt_floats(C, width); }

int ifl_cond_pixels_padded(int B, int H, int W) { return (int)(((size_t)B * H * W + 63) / 64 * 64); }

int ifl_cond_prep_f32(const float *w1, const float *w2, const float *w3, const float *logs, float *wt, int C, int width,
                      float logscale_factor, ifl_stream_t stream)
{
    clear_error();
    if (int rc = cond_check("ifl_cond_prep_f32", 0, C, 1, 1, width, C)) return rc;
    if (!w1 || !w2 || !w3 || !logs || !wt) IFL_FAIL(IFL_EINVAL, "ifl_cond_prep_f32: null pointer");
    const size_t n = cond_wt_floats(C, width);
    hipLaunchKernelGGL(k_cond_prep, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0,
                       (hipStream_t)stream, w1, w2, w3, logs, wt, C, width, logscale_factor);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

int ifl_cond_prep_many_f32(const ifl_cond_prep_job *jobs, int n_jobs, size_t max_weights_floats, ifl_stream_t stream)
{
    clear_error();
    if (n_jobs < 0) IFL_FAIL(IFL_EINVAL, "ifl_cond_prep_many_f32: n_jobs=%d", n_jobs);
    if (n_jobs == 0) return IFL_OK;
    if (!jobs || max_weights_floats == 0) IFL_FAIL(IFL_EINVAL, "ifl_cond_prep_many_f32: null job table or zero size");
    if (n_jobs > 65535) IFL_FAIL(IFL_EUNSUPPORTED, "ifl_cond_prep_many_f32: at most 65535 jobs a call");
    const size_t bl = (max_weights_floats + 255) / 256;
    hipLaunchKernelGGL(k_cond_prep_many, dim3((unsigned)(bl < 64 ? bl : 64), (unsigned)n_jobs), dim3(256), 0, (hipStream_t)stream,
                       (const PrepJob *)jobs);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

int ifl_cond_forward_f32(const float *x, int x_channels, const float *wt, const float *w1, const float *w2, const float *b3, float *a2,
                         float *h, int B, int C, int H, int W, int width, ifl_stream_t stream)
{
    clear_error();
    if (int rc = cond_check("ifl_cond_forward_f32", B, C, H, W, width, x_channels)) return rc;
    if (B == 0) return IFL_OK;
    if (!x || !wt || !w1 || !w2 || !a2 || !h) IFL_FAIL(IFL_EINVAL, "ifl_cond_forward_f32: null pointer");
    const CondShape s{B, H, W, width, x_channels};
    IFL_COND_DISPATCH(C, forward(x, wt, w1, w2, b3, a2, h, s, (hipStream_t)stream));
}

size_t ifl_cond_backward_workspace_bytes(int B, int C, int H, int W, int width, int operands_f32)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || width <= 0) return 0;
    return cond_ws(C, width, ifl_cond_pixels_padded(B, H, W), operands_f32 ? sizeof(float) : sizeof(bf16_t)).total;
}

size_t ifl_cond_grads_floats(int C, int width) { return (size_t)width * 9 * (C / 2) + (size_t)C * width + (size_t)9 * C * C + 2 * (size_t)C; }

int ifl_cond_backward_f32(const float *x, int x_channels, const float *dh, const float *h, const float *a2, const float *wt,
                          const float *w1, int operands_f32, void *ws, size_t ws_bytes, float *grads, float *dx, int B, int C, int H,
                          int W, int width, float logscale_factor, ifl_stream_t stream)
{
    clear_error();
    if (int rc = cond_check("ifl_cond_backward_f32", B, C, H, W, width, x_channels)) return rc;
    if (B == 0) return IFL_OK;
    if (!x || !dh || !h || !a2 || !wt || !w1 || !ws || !grads || !dx) IFL_FAIL(IFL_EINVAL, "ifl_cond_backward_f32: null pointer");
    const size_t need = ifl_cond_backward_workspace_bytes(B, C, H, W, width, operands_f32);
    if (ws_bytes < need || ((uintptr_t)ws & 255)) IFL_FAIL(IFL_EWORKSPACE, "ifl_cond_backward_f32: %zu bytes of 256-aligned workspace needed", need);
    const CondShape s{B, H, W, width, x_channels};
    hipStream_t st = (hipStream_t)stream;
    if (operands_f32) {
        IFL_COND_DISPATCH(C, template backward<float>(x, dh, h, a2, wt, w1, ws, grads, dx, s, logscale_factor, st));
    }
    IFL_COND_DISPATCH(C, template backward<bf16_t>(x, dh, h, a2, wt, w1, ws, grads, dx, s, logscale_factor, st));
}

} // extern "C"
