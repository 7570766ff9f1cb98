// Wide layers (64 < C <= 256, C a multiple of 32: the C = 256 layers of BASELINE configs[4]) at small spatial sizes
// (H <= 16, W <= 32), e.g. a batch shard of 16 images of 256 x 8 x 8.
//
// What binds here is not memory (the whole problem is 1 MB of activations and 2.4 MB of weights) but the chain of
// H + W - 1 dependent anti-diagonals, each a small dense contraction over ALL channels, and the weights of one layer fit
// neither the registers nor the LDS of one compute unit.  So the channels of an image are spread over a TEAM of C/16
// workgroups that stay resident for the whole sweep (one launch instead of one per diagonal):
//
//   workgroup (team, ct) owns output channels 16 ct .. 16 ct + 15 of the team's current tile of 16 columns
//   (column = (image of the tile, image row h); on diagonal d the column's pixel is (h, d - h));
//   wave v of it owns input channels 32 v .. 32 v + 31: its slice of the folded taps, for all taps, stays in registers
//   as split-fp16 MFMA A fragments for the whole launch (2 * KH*KW * 4 registers);
//   per diagonal each wave multiplies its slice (v_mfma_f32_16x16x32_f16, three products per tap), the waves' partial
//   sums meet in LDS, wave 0 rounds the 16 x 16 results, writes them to z and -- already split into fp16 hi/lo, in the
//   layout the consumers' B fragments have -- into the exchange buffer, then raises the workgroup's flag of that
//   diagonal.  The consumers are the waves of the team whose channel slice those 16 channels belong to: each wave waits
//   for two flags only (there is no team-wide barrier) and loads 2 KB.
//
// Left fold (prep.hip):  z_p = Wf_0 x_p - sum_t Wf_t z_{p-t}, Wf_0 = L^-1, Wf_t = L^-1 W_t; the pack holds +Wf_0 and
// -Wf_t.  The fold is a blocked forward substitution L X = [I | W_1 | ...] in fp64 on the fp64 matrix cores, one wave
// per 16 right-hand-side columns; the adjoint (L^T, W_t^T) is the same recurrence on reversed indices.
//
// Flags carry a generation number kept in the caller's scan state (include/invflow.h), so nothing is cleared per
// launch.  A wait is bounded; a team that cannot finish (or a value that leaves the fp16 range) marks the launch and
// the general fp32 scan redoes it from x (gated launch, a no-op otherwise).
//
// The weight gradient of these layers (W = 8: an image row is one 8-wide k-granule) is at the end of this file.
#include "ifl_common.h"
#include "mfma_util.h"
#include <atomic>
#include <type_traits>

namespace ifl {

typedef double doublex4 __attribute__((ext_vector_type(4)));

#ifndef IFL_WIDE_EXP
#define IFL_WIDE_EXP 0 // what-if builds (tools/exp_wide.sh): 1 no flag waits, 2 no exchange loads, 4 no publishing, 8 no MFMAs
#endif
#ifndef IFL_WIDE_SCOPE
#define IFL_WIDE_SCOPE __HIP_MEMORY_SCOPE_SYSTEM
#endif
static constexpr int WIDE_NDMAX = 48;    // diagonals (H <= 16, W <= 32)
static constexpr int WIDE_MAXTEAMS = 64; // teams with a flag block in the scan state
static constexpr int WIDE_NCTMAX = 16;   // workgroups per team (C <= 256)

// ---- the part of the caller's scan state this file uses (behind the duo scan's mailbox) ------------------------------
struct WideState {
    unsigned long long gen;     // flag values of a launch are gen + 1 + sweep
    unsigned int done;          // workgroups of the running launch that have finished
    unsigned int fail;          // != 0: some team gave up or left the fp16 range
    unsigned long long abort_;  // == gen + 1 of a launch whose waits should stop
    unsigned long long voided;  // launches redone by the general scan so far (telemetry; tests read it)
    unsigned long long pad_[12];
    unsigned long long flag[WIDE_MAXTEAMS][WIDE_NDMAX][WIDE_NCTMAX];
};
size_t scan_wide_state_bytes() { return align_up(sizeof(WideState), 256); }
size_t scan_wide_voided_offset() { return offsetof(WideState, voided); }

__device__ __forceinline__ size_t wide_w_index(int co, int ci, int dh, int dw, const Geom &g)
{
    int kh = g.KH - 1 - dh, kw = g.KW - 1 - dw;
    if (g.flipH) kh = g.KH - 1 - kh;
    if (g.flipW) kw = g.KW - 1 - kw;
    return (((size_t)co * g.C + ci) * g.KH + kh) * g.KW + kw;
}
// diagonal-tap matrix in solve order: dir 0 -> L[i][k]; dir 1 (adjoint) -> L^T on reversed indices, again lower triangular
__device__ __forceinline__ float wide_l_entry(const float *w, int i, int k, int dir, const Geom &g)
{
    if (k > i) return 0.f;
    const int r = dir ? g.C - 1 - k : i, c = dir ? g.C - 1 - i : k;
    if (k == i) return g.general_diag ? w[wide_w_index(r, r, 0, 0, g)] : 1.f;
    return w[wide_w_index(r, c, 0, 0, g)];
}

// ---- prep: compact L per direction, inverses of its 16 x 16 diagonal blocks ------------------------------------------
// grid (C/16, ndir, KH*KW), 256 threads; z = 0: lc[dir][i][k permuted inside groups of 16] floats, dinv[dir][bi][r][c] doubles.
// Also rt[dir][t][c][kc] = W_t[c][kc] (operator) / W_t[kc][c] (adjoint): the right-hand sides of the fold, tap-major, so that
// its lanes read consecutive floats (in the layer's layout a tap's entries are 36 bytes apart).
__global__ __launch_bounds__(256) void k_wide_prep(const float *__restrict__ w, float *__restrict__ lc, double *__restrict__ dinv,
                                                   float *__restrict__ rt, Geom g, int dir0, unsigned *zero0, unsigned *zero1)
{
    const int C = g.C, bi = blockIdx.x, dir = dir0 + blockIdx.y, slot = blockIdx.y;
    if (blockIdx.z > 0) { // tap t of rows 16 bi .. + 15, tap-major
        const int NS = g.KH * g.KW, t = blockIdx.z, dh = t / g.KW, dw = t % g.KW;
        float *r = rt + ((size_t)slot * NS + t) * C * C;
        // (sixteen elements per thread: the loads first, all in flight together, then the stores)
        float val[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = threadIdx.x + 256 * u;
            const int kc = idx % C, c = 16 * bi + (idx / C < 16 ? idx / C : 15);
            val[u] = dir ? w[wide_w_index(kc, c, dh, dw, g)] : w[wide_w_index(c, kc, dh, dw, g)];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = threadIdx.x + 256 * u;
            if (idx < 16 * C) r[(size_t)(16 * bi + idx / C) * C + idx % C] = val[u];
        }
        return;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (zero0) *zero0 = 0u;
        if (zero1) *zero1 = 0u;
    }
    float *l = lc + (size_t)slot * C * C;
    // position p = 4 lk + c4 of a group of 16 columns holds column 4 c4 + lk: a lane's float4 at 4 lk is then its
    // A-operand element of the group's four MFMA k-chunks (k = 4 c4 + lk)
    {
        float val[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = threadIdx.x + 256 * u;
            const int i = 16 * bi + (idx / C < 16 ? idx / C : 15), p = idx % C, k = (p & ~15) + 4 * (p & 3) + ((p >> 2) & 3);
            val[u] = wide_l_entry(w, i, k, dir, g);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = threadIdx.x + 256 * u;
            if (idx < 16 * C) l[(size_t)(16 * bi + idx / C) * C + idx % C] = val[u];
        }
    }
    // the diagonal block itself, staged for the sixteen column solves
    __shared__ float blk[16][17];
    blk[threadIdx.x / 16][threadIdx.x % 16] = wide_l_entry(w, 16 * bi + threadIdx.x / 16, 16 * bi + threadIdx.x % 16, dir, g);
    __syncthreads();
    if (threadIdx.x < 16) { // column jj of the inverse of diagonal block bi, by substitution
        const int jj = threadIdx.x;
        double col[16];
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) {
            double a0 = (ii == jj) ? 1.0 : 0.0;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                if (kk < ii && kk >= jj) a0 -= (double)blk[ii][kk] * col[kk];
            col[ii] = ii >= jj ? a0 / (double)blk[ii][ii] : 0.0;
            dinv[(((size_t)slot * (C / 16) + bi) * 16 + ii) * 16 + jj] = col[ii];
        }
    }
}

// ---- fold: X = L^-1 [I | W_1 | ... ] for 16 columns per workgroup -----------------------------------------------------
// grid C/16 column tiles * KH*KW taps * ndir, 256 threads, C*17 doubles of LDS.
// Right-looking blocked substitution on the fp64 matrix cores (16x16x4, 64 cycles each on gfx950, dependent or not:
// tools/mfma_f64_rate_probe.hip): block row bi is finished by the inverse of its diagonal block (the wave that owns it:
// block rows go round the four waves) and published through LDS; every later block row then takes its update
// -L[i][bi] X[bi] in its owner's accumulator.  The update of block row bi + 1 comes first and its owner solves and
// publishes it before the rest of its updates, so that the other waves' updates hide the solve.  One barrier per block row.
// The column blocks of L are requested three steps ahead by loads the compiler does not see (it answers a use of a
// load it tracks with a wait for everything in flight once branches are involved), with hand-counted waits.
// Outputs per direction: wf32[t][kc][c] (fp32 left fold, what the general scan takes) and the scan's register image
//   pack[ct][v][t][hl][lane] (16 B each): lane (m, q) = rows c = 16 ct + m, input channels kc = 32 v + 8 q .. + 7,
//   +Wf_0, -Wf_t, hi = fp16(v), lo = fp16((v - hi) * 2048).
struct WideFoldOut {
    float *wf32[2];
    uintx4 *pack[2];
};
constexpr int fold_loads(int bi) { return bi < 15 ? 4 - (bi + 1) / 4 : 0; } // loads of column block bi per wave
__device__ __forceinline__ void fold_load_f4(floatx4 &v, const float *p)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p));
}
// at most N of the loads above still in flight; redefines the four registers it guards, which keeps their uses behind it
template <int N> __device__ __forceinline__ void fold_wait(floatx4 (&v)[4])
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "n"(N));
}
__global__ __launch_bounds__(256) void k_wide_fold(const float *__restrict__ rt, const float *__restrict__ lc,
                                                   const double *__restrict__ dinv, WideFoldOut out, Geom g, int dir0)
{
    extern __shared__ double xs[]; // [C][17]: row = solve-order index, column = right-hand side
    const int C = g.C, NBK = C / 16, NS = g.KH * g.KW, NW = C / 32;
    // workgroup -> (column tile, tap, direction): the taps with a full right-hand side first, tap 0 (right-hand side =
    // identity: the block rows above the tile's own are zero and skipped) last, so that the compute units that end up with
    // two workgroups (C = 256, 3 x 3, both directions: 288 workgroups) get a short one as the second
    const int ndir = (int)gridDim.x / (NBK * NS), nfull = NBK * (NS - 1) * ndir, wg = blockIdx.x;
    const int jt = wg % NBK;
    const int t = wg < nfull ? 1 + (wg / NBK) % (NS - 1) : 0;
    const int slot = wg < nfull ? wg / (NBK * (NS - 1)) : (wg - nfull) / NBK;
    const int dir = dir0 + slot;
    const int j0 = t == 0 ? (dir ? NBK - 1 - jt : jt) : 0; // first block row of X that is not zero
    const int tid = threadIdx.x, lane = tid & 63, li = lane % 16, lk = lane / 16;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // (in a scalar register: the branches on it are scalar branches)
    const float *l = lc + (size_t)slot * C * C;
    const double *di = dinv + (size_t)slot * NBK * 256;
    const int kc = 16 * jt + li; // this lane's right-hand-side column = input channel of the tap
    const float *rtap = rt + ((size_t)slot * NS + t) * C * C;
    constexpr int XP = 17; // row pitch of xs in doubles (odd: the column reads of the output stage spread over the banks)
    // this wave's block rows: wv, wv + 4, wv + 8, wv + 12 -> accumulator slot i / 4
    doublex4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = doublex4{0.0, 0.0, 0.0, 0.0};
    constexpr int PFD = 3;
    floatx4 lb[PFD + 1][4];
    // column block BI of L for the block rows that still take an update by it: slots (BI + 1) / 4 .. 3, none for the last
    // block (a load whose registers nothing reads would land in whatever the compiler has put there since)
    auto fetch_l = [&](auto bi_c) {
        constexpr int BI = decltype(bi_c)::value;
        const int bi = BI < NBK ? BI : NBK - 1;
#pragma unroll
        for (int a4 = (BI + 1) / 4; a4 < 4 && BI < 15; ++a4) {
            const int i = 4 * a4 + wv < NBK ? 4 * a4 + wv : NBK - 1;
            fold_load_f4(lb[BI % (PFD + 1)][a4], l + (size_t)(16 * i + li) * C + 16 * bi + 4 * lk);
        }
    };
    fetch_l(std::integral_constant<int, 0>{});
    fetch_l(std::integral_constant<int, 1>{});
    fetch_l(std::integral_constant<int, 2>{});
    // right-hand sides and diagonal-block inverses of this wave's (up to) four block rows: loaded once, before the loop
    // (kept as loaded: converting at the load would wait for it on the spot).  Every load is issued by every wave,
    // out-of-range rows clamped to a valid one, and tap 0 (right-hand side = identity) reads its unused slot of rt and
    // masks the bits: a load under a condition becomes a branch with a wait for everything behind it.
    floatx4 rr[4];
    doublex4 dd[4];
    const unsigned keep = t == 0 ? 0u : ~0u;
#pragma unroll
    for (int a4 = 0; a4 < 4; ++a4) {
        const int bi = 4 * a4 + wv < NBK ? 4 * a4 + wv : NBK - 1;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = 16 * bi + 4 * v + lk, c = dir ? C - 1 - i : i;
            const unsigned ld = __float_as_uint(rtap[(size_t)c * C + kc]), id = (c == kc) ? 0x3f800000u : 0u;
            rr[a4][v] = __uint_as_float((ld & keep) | (id & ~keep));
            dd[a4][v] = di[((size_t)bi * 16 + li) * 16 + 4 * v + lk];
        }
    }
    auto solve = [&](auto bi_c) { // block row BI by its owner: X[BI] = Dinv[BI] (R[BI] - acc), into LDS
        constexpr int BI = decltype(bi_c)::value;
        if (BI < NBK && wv == BI % 4) {
            doublex4 sv;
#pragma unroll
            for (int v = 0; v < 4; ++v) sv[v] = (double)rr[BI / 4][v] - acc[BI / 4][v];
            doublex4 res = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) res = __builtin_amdgcn_mfma_f64_16x16x4f64(dd[BI / 4][k4], sv[k4], res, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; ++v) xs[(16 * BI + 4 * v + lk) * XP + li] = res[v];
        }
    };
    auto step = [&](auto bi_c) { // (X[BI] is in LDS) updates by X[BI]; X[BI + 1]
        constexpr int BI = decltype(bi_c)::value;
        if (BI + 1 >= NBK) return; // (uniform)
        if constexpr (BI + PFD < 16) fetch_l(std::integral_constant<int, BI + PFD>{});
        const bool live = BI >= j0; // (uniform; X[BI] = 0 otherwise: nothing to subtract)
        double xb[4] = {0.0, 0.0, 0.0, 0.0};
        if (live) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) xb[c4] = xs[(16 * BI + 4 * c4 + lk) * XP + li];
        }
        fold_wait<fold_loads(BI + 1) + fold_loads(BI + 2) + fold_loads(BI + 3)>(lb[BI % (PFD + 1)]);
        auto update = [&](auto a4_c) {
            constexpr int A4 = decltype(a4_c)::value;
            const int i = 4 * A4 + wv;
            if (A4 < 4 && live && i > BI && i < NBK) {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4)
                    acc[A4 % 4] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)lb[BI % (PFD + 1)][A4 % 4][c4], xb[c4], acc[A4 % 4], 0, 0, 0);
            }
        };
        constexpr int A0 = (BI + 1) / 4; // (block rows of the slots below A0 are done)
        update(std::integral_constant<int, A0>{});
        solve(std::integral_constant<int, BI + 1>{});
        update(std::integral_constant<int, A0 + 1>{});
        update(std::integral_constant<int, A0 + 2>{});
        update(std::integral_constant<int, A0 + 3>{});
        // (a barrier for the LDS alone: __syncthreads() would also wait for the loads in flight)
        if (BI + 1 >= j0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    solve(std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});
    step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{});
    step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{});
    step(std::integral_constant<int, 9>{});
    step(std::integral_constant<int, 10>{});
    step(std::integral_constant<int, 11>{});
    step(std::integral_constant<int, 12>{});
    step(std::integral_constant<int, 13>{});
    step(std::integral_constant<int, 14>{});
    // (C < 256: loads requested for block columns the matrix does not have may still be in flight, into registers that
    // are about to be reused)
    // (the wait names the registers: they stay allocated up to here on every path)
    fold_wait<0>(lb[0]);
    fold_wait<0>(lb[1]);
    fold_wait<0>(lb[2]);
    fold_wait<0>(lb[3]);
    __syncthreads();
    // fp32 left fold: wf32[t][kc][c], c contiguous: a thread per c, the sixteen kc in turn
    float *wf = out.wf32[slot];
    for (int c = tid; c < C; c += 256) {
        const int i = dir ? C - 1 - c : c;
#pragma unroll
        for (int j = 0; j < 16; ++j) wf[((size_t)t * C + 16 * jt + j) * C + c] = (float)xs[i * XP + j];
    }
    // the scan's register image: this workgroup holds input channels 16 jt .. + 15 = half of k-block v = jt / 2
    uintx4 *pk = out.pack[slot];
    const int v = jt / 2, m = tid % 16, qq = (tid / 16) % 2, q = 2 * (jt % 2) + qq;
    const double sgn = t == 0 ? 1.0 : -1.0;
    for (int ct = tid / 32; ct < NBK; ct += 8) {
        const int c = 16 * ct + m, i = dir ? C - 1 - c : c;
        half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float val = (float)(sgn * xs[i * XP + 8 * qq + e]);
            const _Float16 h = (_Float16)val;
            hi[e] = h;
            lo[e] = (_Float16)((val - (float)h) * LO_SCALE);
        }
        const size_t base = ((((size_t)ct * NW + v) * NS + t) * 2) * 64 + 16 * q + m;
        pk[base] = __builtin_bit_cast(uintx4, hi);
        pk[base + 64] = __builtin_bit_cast(uintx4, lo);
    }
}

// ---- the scan ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t wide_pix(int b, int c, int h, int w, const Geom &g, int rh, int rw)
{
    const int hs = rh ? g.H - 1 - h : h;
    const int ws = rw ? g.W - 1 - w : w;
    return (((size_t)b * g.C + c) * g.H + hs) * g.W + ws;
}
// exchange-buffer accesses: system-scope relaxed atomics of 8 bytes (they bypass the non-coherent cache levels; the
// compiler keeps the wait counters)
__device__ __forceinline__ uintx4 load_sys16(const void *p)
{
    const unsigned long long *q = (const unsigned long long *)p;
    const unsigned long long a0 = __hip_atomic_load(q, __ATOMIC_RELAXED, IFL_WIDE_SCOPE);
    const unsigned long long a1 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, IFL_WIDE_SCOPE);
    return uintx4{(unsigned)a0, (unsigned)(a0 >> 32), (unsigned)a1, (unsigned)(a1 >> 32)};
}
__device__ __forceinline__ void store_sys8(void *p, uintx2 v)
{
    __hip_atomic_store((unsigned long long *)p, ((unsigned long long)v[1] << 32) | v[0], __ATOMIC_RELAXED, IFL_WIDE_SCOPE);
}

struct WideScanArgs {
    const float *x;
    const uintx4 *pack;
    float *z;
    unsigned char *exch; // [tile][diagonal][v][hl][column][q] 16 B
    WideState *st;
    int *gate;           // [B]: 1 -> the general scan redoes the batch
    unsigned *amax;      // optional: max|z| as float bits (atomicMax)
    int rh, rw, ntiles, nteams, hp_log2, xcd_map;
};

template <int NW, int KH, int KW>
__global__ __launch_bounds__(NW * 64) void k_scan_team(WideScanArgs a, Geom g)
{
    constexpr int C = 32 * NW, NCT = C / 16, NS = KH * KW;
    __shared__ floatx4 red[2][NW][64];
    __shared__ int s_stop;
    const int tid = threadIdx.x, lane = tid & 63, v = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    int team, ct;
    if (a.xcd_map) { // the workgroups of a team on one XCD (workgroup ids go round the 8 XCDs)
        const int xc = blockIdx.x & 7, k = blockIdx.x >> 3;
        team = xc + 8 * (k / NCT);
        ct = k % NCT;
    } else {
        team = blockIdx.x / NCT;
        ct = blockIdx.x % NCT;
    }
    const int H = g.H, W = g.W, ND = H + W - 1, B = g.B;
    WideState *st = a.st;
    const unsigned long long base = __hip_atomic_load(&st->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) s_stop = 0;

    // this wave's slice of the taps
    half8 Ah[NS], Al[NS];
    {
        const uintx4 *pk = a.pack + ((size_t)(ct * NW + v) * NS * 2) * 64 + lane;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            Ah[t] = __builtin_bit_cast(half8, pk[(size_t)(2 * t) * 64]);
            Al[t] = __builtin_bit_cast(half8, pk[(size_t)(2 * t + 1) * 64]);
        }
    }
    const int hp = 1 << a.hp_log2, h = n & (hp - 1), img = n >> a.hp_log2;
    float zmax = 0.f;
    int bad = 0;
    __syncthreads();

    int sweep = 0;
    for (int tile = team; tile < a.ntiles; tile += a.nteams, ++sweep) {
        const unsigned long long fv = base + 1 + (unsigned long long)sweep;
        const int b = tile * (16 >> a.hp_log2) + img;
        const bool col_ok = h < H && b < B;
        unsigned char *ex = a.exch + (size_t)tile * WIDE_NDMAX * NW * 2 * 1024;
        unsigned long long *fl = &st->flag[team][0][0];

        float xr[8];
        auto load_x = [&](int d) {
            const int wq = d - h;
            const bool ok = col_ok && wq >= 0 && wq < W && d < ND;
#pragma unroll
            for (int e = 0; e < 8; ++e) xr[e] = ok ? a.x[wide_pix(b, 32 * v + 8 * q + e, h, wq, g, a.rh, a.rw)] : 0.f;
        };
        load_x(0);
        // this wave's channels of the last NHIST diagonals (hi and lo halves), newest first; all zero before the image
        constexpr int NHIST = KH + KW - 2;
        uintx4 hh_[NHIST + 1], hl_[NHIST + 1];
#pragma unroll
        for (int j = 0; j <= NHIST; ++j) hh_[j] = hl_[j] = uintx4{0u, 0u, 0u, 0u};
        // column n - DH of a fragment (same image: rows h >= DH), by a row shift across the 16 lanes of a quad group
        auto shifted = [&](const uintx4 &f, auto dh_c) -> half8 {
            constexpr int DH = decltype(dh_c)::value;
            if constexpr (DH == 0) {
                return __builtin_bit_cast(half8, f);
            } else {
                uintx4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned sft = (unsigned)__builtin_amdgcn_update_dpp(0, (int)f[i], 0x110 + DH, 0xf, 0xf, true);
                    o[i] = h >= DH ? sft : 0u;
                }
                return __builtin_bit_cast(half8, o);
            }
        };
        for (int d = 0; d < ND; ++d) {
            half8 bh, bl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const _Float16 hh = (_Float16)xr[e];
                bh[e] = hh;
                bl[e] = (_Float16)((xr[e] - (float)hh) * LO_SCALE);
            }
            load_x(d + 1);
#pragma unroll
            for (int j = NHIST; j >= 2; --j) {
                hh_[j] = hh_[j - 1];
                hl_[j] = hl_[j - 1];
            }
            floatx4 acc = {0.f, 0.f, 0.f, 0.f}, acx = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[0], bh, acc, 0, 0, 0);
            acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[0], bl, acx, 0, 0, 0);
            acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[0], bh, acx, 0, 0, 0);
            // taps that reach two or more diagonals back: everything they need is in registers already
            auto taps_of = [&](auto j_c) {
                constexpr int J = decltype(j_c)::value;
#pragma unroll
                for (int t = 1; t < NS; ++t) {
                    const int dh = t / KW, dw = t % KW;
                    if (dh + dw != J || (IFL_WIDE_EXP & 8)) continue;
                    half8 zh, zl;
                    if (dh == 0) {
                        zh = shifted(hh_[J], std::integral_constant<int, 0>{});
                        zl = shifted(hl_[J], std::integral_constant<int, 0>{});
                    } else if (dh == 1) {
                        zh = shifted(hh_[J], std::integral_constant<int, 1>{});
                        zl = shifted(hl_[J], std::integral_constant<int, 1>{});
                    } else {
                        zh = shifted(hh_[J], std::integral_constant<int, 2>{});
                        zl = shifted(hl_[J], std::integral_constant<int, 2>{});
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t], zh, acc, 0, 0, 0);
                    acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t], zl, acx, 0, 0, 0);
                    acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[t], zh, acx, 0, 0, 0);
                }
            };
            if constexpr (NHIST >= 4) taps_of(std::integral_constant<int, 4>{});
            if constexpr (NHIST >= 3) taps_of(std::integral_constant<int, 3>{});
            if constexpr (NHIST >= 2) taps_of(std::integral_constant<int, 2>{});
            // wait for the two workgroups that own this wave's input channels on diagonal d - 1, then take it in
            int stop = 0;
            hh_[1] = hl_[1] = uintx4{0u, 0u, 0u, 0u};
            if (d > 0 && !(IFL_WIDE_EXP & 1)) {
                const unsigned long long *fp = fl + (size_t)(d - 1) * WIDE_NCTMAX + 2 * v + (lane & 1);
                for (int spins = 0;; ++spins) {
                    const unsigned long long f = __hip_atomic_load(fp, __ATOMIC_RELAXED, IFL_WIDE_SCOPE);
                    if (__all(f >= fv)) break;
                    if ((spins & 63) == 63 &&
                        __hip_atomic_load(&st->abort_, __ATOMIC_RELAXED, IFL_WIDE_SCOPE) == base + 1) {
                        stop = 1;
                        break;
                    }
                    if (spins > 200000) { // ~0.1 s: give the launch up
                        stop = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (d > 0 && !stop && !(IFL_WIDE_EXP & 2)) {
                const unsigned char *src = ex + ((size_t)((d - 1) * NW + v) * 2) * 1024 + (n * 4 + q) * 16;
                hh_[1] = load_sys16(src);
                hl_[1] = load_sys16(src + 1024);
            }
            taps_of(std::integral_constant<int, 1>{});
            floatx4 part;
#pragma unroll
            for (int i = 0; i < 4; ++i) part[i] = acc[i] + acx[i] * LO_INV;
            red[d & 1][v][lane] = part;
            if (stop) s_stop = 1;
            // (a barrier for the LDS alone: __syncthreads() also waits for every load in flight -- the x of the next diagonal)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (s_stop) break; // (uniform: written before the barrier, never cleared)
            if (v == 0) {
                floatx4 zz = red[d & 1][0][lane];
#pragma unroll
                for (int u = 1; u < NW; ++u) {
                    const floatx4 o = red[d & 1][u][lane];
#pragma unroll
                    for (int i = 0; i < 4; ++i) zz[i] += o[i];
                }
                // rows 4 q + i of the tile = channels 16 ct + 4 q + i, column n
                const int wq = d - h;
                const bool ok = col_ok && wq >= 0 && wq < W;
                half4 zh4, zl4;
                float zval[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float val = ok ? zz[i] : 0.f;
                    const float av = fabsf(val);
                    zmax = fmaxf(zmax, av);
                    if (!(av < 60000.f)) bad = 1; // (also NaN)
                    const _Float16 hh = (_Float16)val;
                    zh4[i] = hh;
                    zl4[i] = (_Float16)((val - (float)hh) * LO_SCALE);
                    zval[i] = val;
                }
                // consumers: k-block ct / 2, granule 2 (ct % 2) + q / 2, halves 4 (q % 2) .. + 3.  The exchange first, then the
                // flag, then z itself: the flag waits for nothing but the two exchange stores
                unsigned char *dst = ex + ((size_t)(d * NW + ct / 2) * 2) * 1024 + (n * 4 + 2 * (ct & 1) + (q >> 1)) * 16 + 8 * (q & 1);
                if (!(IFL_WIDE_EXP & 4)) {
                    store_sys8(dst, __builtin_bit_cast(uintx2, zh4));
                    store_sys8(dst + 1024, __builtin_bit_cast(uintx2, zl4));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (lane == 0 && !(IFL_WIDE_EXP & 4))
                    __hip_atomic_store(fl + (size_t)d * WIDE_NCTMAX + ct, fv, __ATOMIC_RELAXED, IFL_WIDE_SCOPE);
                if (ok) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a.z[wide_pix(b, 16 * ct + 4 * q + i, h, wq, g, a.rh, a.rw)] = zval[i];
                }
            }
        }
        if (s_stop) break;
    }
    __syncthreads();
    // verdict of the launch; the last workgroup to finish writes the gate, clears the counters and advances the generation
    if (tid == 0) {
        if (s_stop) {
            __hip_atomic_store(&st->abort_, base + 1, __ATOMIC_RELAXED, IFL_WIDE_SCOPE);
            atomicOr(&st->fail, 1u);
        }
    }
    if (v == 0) {
        bad = __any(bad);
        for (int o = 32; o > 0; o >>= 1) zmax = fmaxf(zmax, __shfl_down(zmax, o, 64));
        if (lane == 0) {
            if (bad) atomicOr(&st->fail, 1u);
            // (a launch that stopped early knows no maximum: +inf sends the weight gradient down its plain fp32 path)
            if (a.amax) atomicMax(a.amax, s_stop ? 0x7f800000u : __float_as_uint(zmax));
            __threadfence();
            const unsigned total = gridDim.x;
            if (atomicAdd(&st->done, 1u) == total - 1) {
                __threadfence();
                const unsigned f = atomicExch(&st->fail, 0u);
                if (f) st->voided += 1;
                for (int i = 0; i < B; ++i) a.gate[i] = f ? 1 : 0;
                st->done = 0;
                const int maxsweeps = (a.ntiles + a.nteams - 1) / a.nteams;
                __hip_atomic_store(&st->gen, base + 1 + (unsigned long long)maxsweeps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence();
            }
        }
    }
}

static int wide_hp_log2(int H)
{
    int l = 0;
    while ((1 << l) < H) ++l;
    return l;
}
static int wide_ntiles(const Geom &g) { return (g.B + (16 >> wide_hp_log2(g.H)) - 1) / (16 >> wide_hp_log2(g.H)); }

bool scan_team_supported(const Geom &g)
{
    return g.C > 64 && g.C <= 256 && g.C % 32 == 0 && g.H <= 16 && g.W <= 32 &&
           ((g.KH == 3 && g.KW == 3) || (g.KH == 2 && g.KW == 2));
}
// workspace of the route: compact L (2 directions), block inverses, tap-major right-hand sides, exchange buffer
static size_t wide_ws_dinv_off(size_t C) { return align_up(2 * C * C * sizeof(float), 256); }
static size_t wide_ws_rt_off(size_t C) { return wide_ws_dinv_off(C) + align_up(2 * C * 16 * sizeof(double), 256); }
static size_t wide_ws_exch_off(size_t C, int ns) { return wide_ws_rt_off(C) + align_up(2 * (size_t)ns * C * C * sizeof(float), 256); }
size_t scan_team_ws_bytes(const Geom &g)
{
    if (!scan_team_supported(g)) return 0;
    const size_t C = (size_t)g.C;
    return wide_ws_exch_off(C, g.KH * g.KW) + align_up((size_t)wide_ntiles(g) * WIDE_NDMAX * (C / 32) * 2 * 1024, 256) + 256;
}

static int device_cu_count()
{
    // (per device, read once: the attribute query costs microseconds of host time per call)
    static std::atomic<int> cache[IFL_MAX_DEVICES];
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < IFL_MAX_DEVICES && (n = cache[dev].load(std::memory_order_relaxed)) > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < IFL_MAX_DEVICES) cache[dev].store(n, std::memory_order_relaxed);
    return n;
}

// fold one or both directions of the operator into pack / wf32 form
int launch_fold_team(const float *w, void *ws, const Geom &g, int dir0, int ndir, void *pack0, float *wf0, void *pack1,
                     float *wf1, unsigned *zero0, unsigned *zero1, hipStream_t s)
{
    const size_t C = (size_t)g.C;
    float *lc = (float *)ws;
    double *dinv = (double *)((char *)ws + wide_ws_dinv_off(C));
    WideFoldOut out;
    out.wf32[0] = wf0;
    out.wf32[1] = wf1;
    out.pack[0] = (uintx4 *)pack0;
    out.pack[1] = (uintx4 *)pack1;
    float *rt = (float *)((char *)ws + wide_ws_rt_off(C));
    hipLaunchKernelGGL(k_wide_prep, dim3(g.C / 16, ndir, g.KH * g.KW), dim3(256), 0, s, w, lc, dinv, rt, g, dir0, zero0, zero1);
    hipLaunchKernelGGL(k_wide_fold, dim3(g.C / 16 * g.KH * g.KW * ndir), dim3(256), C * 17 * sizeof(double), s, rt, lc, dinv, out, g,
                       dir0);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

template <int NW, int KH, int KW> static void launch_team_kernel(const WideScanArgs &a, const Geom &g, int nwg, hipStream_t s)
{
    hipLaunchKernelGGL((k_scan_team<NW, KH, KW>), dim3(nwg), dim3(NW * 64), 0, s, a, g);
}

// z = scan(x) with the packed fold; `gate` (B ints) tells the caller's gated general scan whether to redo the batch
int launch_scan_team(const float *x, const void *pack, float *z, const Geom &g, int rh, int rw, void *ws, void *state,
                     int *gate, unsigned *amax, hipStream_t s)
{
    const int NW = g.C / 32, NCT = g.C / 16;
    const int cus = device_cu_count();
    if (cus < NCT) IFL_FAIL(IFL_EUNSUPPORTED, "wide scan: %d compute units for teams of %d workgroups", cus, NCT);
    WideScanArgs a;
    a.x = x;
    a.pack = (const uintx4 *)pack;
    a.z = z;
    const size_t C = (size_t)g.C;
    a.exch = (unsigned char *)ws + wide_ws_exch_off(C, g.KH * g.KW);
    a.st = (WideState *)state;
    a.gate = gate;
    a.amax = amax;
    a.rh = rh;
    a.rw = rw;
    a.ntiles = wide_ntiles(g);
    a.hp_log2 = wide_hp_log2(g.H);
    int nteams = cus / NCT;
    if (nteams > a.ntiles) nteams = a.ntiles;
    if (nteams > WIDE_MAXTEAMS) nteams = WIDE_MAXTEAMS;
    if (nteams >= 8) nteams -= nteams % 8;
    a.nteams = nteams;
    a.xcd_map = nteams % 8 == 0;
    const int nwg = nteams * NCT;
    const bool k3 = g.KH == 3;
    switch (NW) {
#define IFL_TEAM(N)                                                \
    case N:                                                        \
        if (k3) launch_team_kernel<N, 3, 3>(a, g, nwg, s);         \
        else launch_team_kernel<N, 2, 2>(a, g, nwg, s);            \
        break;
        IFL_TEAM(3) IFL_TEAM(4) IFL_TEAM(5) IFL_TEAM(6) IFL_TEAM(7) IFL_TEAM(8)
#undef IFL_TEAM
    default: IFL_FAIL(IFL_EUNSUPPORTED, "wide scan: C=%d", g.C);
    }
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

// ---- weight gradient, W = 8 -------------------------------------------------------------------------------------------
//     dw[co][ci][kh][kw] = scale * sum_{b,oh,ow} gz[b][co][oh][ow] * x[b][ci][oh - pt + kh][ow - pl + kw]
// One workgroup per 16 x 16 block of (co, ci), all taps; the reduction index of an MFMA k-block is four image rows of
// eight pixels: lane (m, q) of the A operand holds row 4 rg + q of gz[b][16 bco + m], straight from memory (32 B), and
// the B operand is the matching row of x, once per kh, shifted by kw - pl columns in registers (v_alignbit on the packed
// halves, zero fill).  The four waves take every fourth k-block and meet in LDS.  Split fp16 after the power-of-two
// prescale of wgrad_mfma.hip; tensors of extreme range (wgrad_wide_range) are contracted in plain fp32 by the same
// workgroup instead.
template <int S> __device__ __forceinline__ half8 shift_cols(const half8 &f)
{
    // element e <- element e + S, zero outside 0..7
    const uintx4 d = __builtin_bit_cast(uintx4, f);
    auto P = [&](int i) -> unsigned { return (i >= 0 && i < 4) ? d[i < 0 ? 0 : (i > 3 ? 3 : i)] : 0u; };
    uintx4 o;
    if constexpr ((S & 1) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = P(i + S / 2);
    } else {
        constexpr int M = (S - 1) / 2 - ((S - 1) % 2 != 0 && S < 0 ? 1 : 0); // floor((S - 1) / 2)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __builtin_amdgcn_alignbit(P(i + M + 1), P(i + M), 16);
    }
    return __builtin_bit_cast(half8, o);
}

__device__ __forceinline__ void split_row(const floatx4 &v0, const floatx4 &v1, float s, half8 &hi, half8 &lo)
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

template <int KH, int KW, int PL>
__global__ __launch_bounds__(256) void k_wgrad_w8(const float *__restrict__ gz, const float *__restrict__ x, float *__restrict__ dw,
                                                  const unsigned *__restrict__ amax_gz, const unsigned *__restrict__ amax_x, int B,
                                                  int C, int H, int pt, float scale, int mask_mode, int mkh, int mkw)
{
    constexpr int NT = KH * KW, W = 8;
    __shared__ floatx4 red[4][NT][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    const int nb = C / 16, bco = blockIdx.x / nb, bci = blockIdx.x % nb;
    const unsigned ag = *amax_gz, ax = *amax_x;
    if (wgrad_wide_range(ag, ax)) { // plain fp32, one (co, ci) pair per thread
        const int co = 16 * bco + tid / 16, ci = 16 * bci + tid % 16;
        float acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = 0.f;
        for (int b = 0; b < B; ++b)
            for (int oh = 0; oh < H; ++oh)
                for (int ow = 0; ow < W; ++ow) {
                    const float gv = gz[(((size_t)b * C + co) * H + oh) * W + ow];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int ih = oh - pt + t / KW, iw = ow - PL + t % KW;
                        if (ih >= 0 && ih < H && iw >= 0 && iw < W) acc[t] = fmaf(gv, x[(((size_t)b * C + ci) * H + ih) * W + iw], acc[t]);
                    }
                }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float val = acc[t] * scale;
            if (mask_mode && t / KW == mkh && t % KW == mkw) {
                if (mask_mode == 1 && ci >= co) val = 0.f;
                if (mask_mode == 2 && ci > co) val = 0.f;
            }
            dw[((size_t)co * C + ci) * NT + t] = val;
        }
        return;
    }
    const float sa = pow2_scale(ag), sb = pow2_scale(ax);
    floatx4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int rgs = (H + 3) / 4, nkb = B * rgs;
    const floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int kb = wv; kb < nkb; kb += 4) {
        const int b = kb / rgs, r = 4 * (kb % rgs) + q;
        const float *gp = gz + (((size_t)b * C + 16 * bco + m) * H) * W;
        const float *xp = x + (((size_t)b * C + 16 * bci + m) * H) * W;
        floatx4 g0 = zero4, g1 = zero4;
        if (r < H) {
            g0 = *(const floatx4 *)(gp + r * W);
            g1 = *(const floatx4 *)(gp + r * W + 4);
        }
        floatx4 x0[KH], x1[KH];
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            const int xr = r - pt + kh;
            x0[kh] = zero4;
            x1[kh] = zero4;
            if (r < H && xr >= 0 && xr < H) {
                x0[kh] = *(const floatx4 *)(xp + xr * W);
                x1[kh] = *(const floatx4 *)(xp + xr * W + 4);
            }
        }
        half8 ah, al;
        split_row(g0, g1, sa, ah, al);
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            half8 bh, bl;
            split_row(x0[kh], x1[kh], sb, bh, bl);
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                half8 sh, sl;
                if constexpr (true) {
                    // (kw - PL as a compile-time shift: kw is an unrolled constant)
                    switch (kw - PL) {
                    case -2: sh = shift_cols<-2>(bh); sl = shift_cols<-2>(bl); break;
                    case -1: sh = shift_cols<-1>(bh); sl = shift_cols<-1>(bl); break;
                    case 0: sh = bh; sl = bl; break;
                    case 1: sh = shift_cols<1>(bh); sl = shift_cols<1>(bl); break;
                    default: sh = shift_cols<2>(bh); sl = shift_cols<2>(bl); break;
                    }
                }
                floatx4 &ac = acc[kh * KW + kw];
                ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, sh, ac, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, sl, ac, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, sh, ac, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) red[wv][t][lane] = acc[t];
    __syncthreads();
    // thread (lane', i): element (row 4 q' + i, column n') of the block, all taps: nine consecutive floats of dw
    const int l2 = tid & 63, i = tid >> 6, co = 16 * bco + 4 * (l2 >> 4) + i, ci = 16 * bci + (l2 & 15);
    const float inv = scale / (sa * sb);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float val = (red[0][t][l2][i] + red[1][t][l2][i] + red[2][t][l2][i] + red[3][t][l2][i]) * inv;
        if (mask_mode && t / KW == mkh && t % KW == mkw) {
            if (mask_mode == 1 && ci >= co) val = 0.f;
            if (mask_mode == 2 && ci > co) val = 0.f;
        }
        dw[((size_t)co * C + ci) * NT + t] = val;
    }
}

bool wgrad_w8_supported(int B, int C, int H, int W, int KH, int KW, int pt, int pl, const void *gz, const void *x)
{
    if (C <= 64 || C % 16 != 0 || W != 8 || H < 1 || B < 1) return false;
    if (!((KH == 3 && KW == 3) || (KH == 2 && KW == 2))) return false;
    if (!((pt == 0 || pt == KH - 1) && (pl == 0 || pl == KW - 1))) return false;
    if (((uintptr_t)gz | (uintptr_t)x) & 15) return false;
    return true;
}

// ws: 256 bytes (the two maxima when the caller has none)
int launch_wgrad_w8(const float *gz, const float *x, float *dw, void *ws, int B, int C, int H, int KH, int KW, int pt, int pl,
                    float scale, int mask_mode, int mkh, int mkw, const unsigned *amax_gz, const unsigned *amax_x,
                    hipStream_t s)
{
    if (!amax_gz || !amax_x) {
        unsigned *absmax = (unsigned *)ws;
        if (int rc = launch_absmax2(gz, x, (size_t)B * C * H * 8, absmax, s)) return rc;
        amax_gz = absmax;
        amax_x = absmax + 1;
    }
    const dim3 grid((C / 16) * (C / 16));
#define IFL_W8(K, PLV)                                                                                                       \
    hipLaunchKernelGGL((k_wgrad_w8<K, K, PLV>), grid, dim3(256), 0, s, gz, x, dw, amax_gz, amax_x, B, C, H, pt, scale, mask_mode, \
                       mkh, mkw)
    if (KH == 3) {
        if (pl == 0) IFL_W8(3, 0);
        else IFL_W8(3, 2);
    } else {
        if (pl == 0) IFL_W8(2, 0);
        else IFL_W8(2, 1);
    }
#undef IFL_W8
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

} // namespace ifl
