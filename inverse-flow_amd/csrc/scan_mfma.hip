// MFMA wavefront back-substitution scan for C in {32, 64} (gfx950, wave64).
//
// Formulation ("right fold"): with L the diagonal-tap matrix and W_t the other taps,
//     r_p = x_p - sum_t (W_t L^-1) r_{p-t},      z_p = L^-1 r_p
// so the sequential chain carries r only; x enters as the plain fp32 accumulator seed and
// z = L^-1 r is a per-pixel product that is *off* the dependency chain.
//
// Mapping.  One workgroup per image, one wave per 16 output channels (C/16 waves, one per SIMD).
// MFMA tile = 16 channels x 16 image rows: MFMA column n of tile T is image row h = 16T+n for the
// whole kernel and walks along w = d-h as the anti-diagonal index d advances (pixel (h,w) is on
// diagonal h+w; every source (h-dh, w-dw) is on an earlier diagonal: solve_mc.py:88-114 in diagonal
// order, cf. solve_parallel, solve_mc.py:8-50).
//
// Push form.  A step d reads the newest diagonal r_{d-1} from LDS once (three row-shifted fragment sets per
// tile) and pushes it through every tap into the rolling accumulator of the diagonal it lands on (d, d+1,
// d+2); only the taps (0,1) and (1,0) are on the dependent chain, the other MFMAs of the step have their
// operands before or right after the barrier.
//
// LDS (~156 KB at C=64, 32 rows):
//   ring   the last 2 diagonals of r as split fp16 (hi, lo*2^11), [slot][row block][plane][row%16]
//          16-byte pieces, plane = (k-step, hi/lo, k-group): a lane's MFMA B fragment is one
//          ds_read_b128 and the 16 lanes of every hardware lane group hit 16 different 4-bank
//          columns (conflict-free; the naive [row][channel] layout measured 52 % conflict cycles).
//          Row block 0 of a slot is always zero (sources above the image), pixels left of the image are
//          never written and stay zero: exactly the TL zero padding.
//   xs     x quads [row][quad parity][channel][4], filled by LDS-DMA (global_load_lds_dwordx4: no VGPR
//          destination, nothing for the compiler to track) three steps before their first use.
//   zq     z quads in the same layout; a row's quad is stored to NCHW with one 16-byte store per channel
//          the step after its fourth column was staged.
//   dump   where lanes outside the image write their r (branch-free epilogue).
// Both global streams move 16-byte aligned quads of one (channel, row) line; lane = channel.
//
// The only vector-memory operations of a wave are its own DMAs (a fixed number per step) and stores, so the
// wait before each barrier is the immediate s_waitcnt vmcnt(2G): "the DMA issued three steps ago has
// landed", never "everything, including the stores I just issued".
//
// Arithmetic: split-fp16 MFMA with fp32 accumulation.  a*b ~= ah*bh + (ah*bl' + al'*bh) 2^-11 with
// ah = fp16(a), al' = fp16((a-ah) 2^11): three v_mfma_f32_16x16x32_f16 per 32-deep k-step, the
// dropped al*bl term is 2^-22 relative; fp16 denormal operands are kept by the MFMA (checked on
// gfx950 with tools/mfma_f16_denorm_probe.hip).  The folded weights live in AGPRs for the whole scan.
#include <type_traits>

#include "ifl_common.h"
#include "mfma_util.h"
#include "scan_general_body.h"
#include <stdlib.h>

#include <atomic>

namespace ifl {

template <int C, int KH, int KW, int NTILE> struct ScanCfg {
    static constexpr int NW = C / 16;      // 16-channel output groups
    static constexpr int NWAVES = NW;      // one wave per channel group (one per SIMD at C=64), all row tiles: 512 registers
    static constexpr int NQ = C / 32;      // 32-deep k-steps per tap
    static constexpr int NT = KH * KW;     // taps incl. the diagonal one
    static constexpr int NS = NT;          // A slots: NT-1 folded taps + 1 post matrix (L^-1)
    static constexpr int R = 2;            // r-ring depth: the diagonal being written and the previous one (push form)
    static constexpr int NACC = 3;         // rolling accumulators: diagonals d, d+1, d+2
    static_assert(KH + KW - 2 <= 4 && KH <= 3, "push scan: taps reach at most 4 diagonals ahead, 2 rows up");
    static constexpr int NPL = NQ * 8;     // planes per row block: (k-step, hi/lo, k-group)
    static constexpr int RBB = NPL * 256;  // bytes of one row block (16 rows x NPL planes x 16 B)
    static constexpr int SLOTB = (NTILE + 1) * RBB; // one diagonal: an always-zero block (sources above the image) + the tiles
    static constexpr int RINGB = R * SLOTB;
    static constexpr int XROWB = 2 * C * 16 + 16; // quads of one row: [parity][channel][4] + pad (x staging and z staging)
    static constexpr int XSB = 16 * NTILE * XROWB;
    static constexpr int OFF_XS = RINGB, OFF_ZQ = OFF_XS + XSB, OFF_DUMP = OFF_ZQ + XSB;
    static constexpr int LDSB = OFF_DUMP + 4 * 256 + 64 * NWAVES * 8; // dump: where lanes outside the image write their r (branch-free epilogue)
    static constexpr int THREADS = 64 * NWAVES;
    static constexpr int ROWS_PER_ITER = 4 * NTILE; // rows that start/finish a quad each step
    static constexpr int G = ROWS_PER_ITER / NWAVES; // ... per wave: G DMAs and G (possibly masked) stores per step
    static_assert(ROWS_PER_ITER % NWAVES == 0, "rows per step must split evenly over the waves");
    static_assert(C <= 64, "one DMA / store instruction covers one image row of all channels (lane = channel)");
    static_assert(KH <= 16, "a source row is at most one row block up");
};

// "all but the N youngest vector-memory operations of this wave are complete", LDS drained, then the workgroup
// barrier.  N is a constant (see the step body): an immediate, no dispatch on a run-time count.
template <int N> __device__ __forceinline__ void wait_vm_then_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// Development aid (tools/stamps.py; build with HIPCC_EXTRA=-DIFL_STAMPS): per-wave cycle counts of the sections
// of a step, summed over the steps of workgroup 0.  Compiled out of the product build.
#ifdef IFL_STAMPS
__device__ unsigned long long *g_stamps = nullptr;
#define IFL_STAMP(k)                                            \
    do {                                                        \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        st_acc[k] += t_ - st_last;                              \
        st_last = t_;                                           \
    } while (0)
#else
#define IFL_STAMP(k) \
    do {             \
    } while (0)
#endif

// PAD: the layer has fewer channels than the instantiation (geom.C < C; the weights are padded with the identity by
// k_foldpack): lanes beyond geom.C move nothing, their x staging is zero.
template <int C, int KH, int KW, int NTILE, bool PAD>
__device__ __forceinline__ int scan_body(const float *__restrict__ xin, float *__restrict__ zout,
                                         const half8 *__restrict__ apack, const int H, const int W, const int rh,
                                         const int rw, int *__restrict__ flags, const float *__restrict__ wf32,
                                         const Geom &geom, unsigned *__restrict__ amax, const int b)
{
    using Cfg = ScanCfg<C, KH, KW, NTILE>;
    constexpr int NQ = Cfg::NQ, NS = Cfg::NS, RBB = Cfg::RBB, SLOTB = Cfg::SLOTB, G = Cfg::G;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *ring = lds;
    unsigned char *xs = lds + Cfg::OFF_XS;
    unsigned char *zq = lds + Cfg::OFF_ZQ;

    const int tid = threadIdx.x;
#ifdef IFL_STAMPS
    const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index: scalar
    const int wv = wave; // 16-channel output group of this wave
    const int n = lane & 15, g = lane >> 4;
    constexpr int hoff = 0;
    const int Hp = H;
    const int ND = Hp + W - 1;

    // ---- zero the r-ring and the zero block (zero padding of the operator) ---------------------------
    {
        const floatx4 zz = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid * 16; i < Cfg::RINGB; i += Cfg::THREADS * 16) *(floatx4 *)(lds + i) = zz;
        // padded channels are never loaded: their x staging must read as zero (finite times a zero weight)
        if (PAD)
            for (int i = Cfg::OFF_XS + tid * 16; i < Cfg::OFF_ZQ; i += Cfg::THREADS * 16) *(floatx4 *)(lds + i) = zz;
    }

    // ---- folded weights -> registers (A fragments, hi and lo) -----------------------------------
    half8 A[NS][NQ][2];
    {
        // all loads first, then the pins: a pin right behind its load makes every load wait for its own data
        // (36 dependent L2 round trips at C = 64, 3x3)
        half8 Aload[NS][NQ][2];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
                    Aload[s][q][hl] = apack[((((size_t)wv * NS + s) * NQ + q) * 2 + hl) * 64 + lane];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    A[s][q][hl] = Aload[s][q][hl];
                    // pin the fragment in the accumulator half of the register file (MFMA reads A from AGPRs
                    // directly); without this hipcc re-loads the weights from memory inside the scan loop
                    asm volatile("" : "+a"(A[s][q][hl]));
                }
    }

    // ---- per-lane constants: lane (n, g) owns pixel rows h = 16T+n and channels c0..c0+3 (C/D layout) ----
    const int c0 = 16 * wv + 4 * g;
    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    unsigned radr[NTILE][KH]; // LDS address (slot 0) of this lane's B piece for a source dh rows up
    int wadr[NTILE], xadr[NTILE];
    bool hval[NTILE];
#pragma unroll
    for (int T = 0; T < NTILE; ++T) {
        const int h = 16 * T + n;
        hval[T] = h < Hp;
#pragma unroll
        for (int dh = 0; dh < KH; ++dh) {
            const int hs = h - dh + 16; // row block 0 of a slot is the zero block
            radr[T][dh] = ldsbase + (hs / 16) * RBB + g * 256 + (hs % 16) * 16;
        }
        wadr[T] = (T + 1) * RBB + (((c0 / 32) * 2) * 4 + (c0 % 32) / 8) * 256 + n * 16 + ((c0 % 8) / 4) * 8;
        xadr[T] = h * Cfg::XROWB + c0 * 16;
    }

    // DMA / store role: lane = channel, G rows per wave and step
    const int Cr = PAD ? geom.C : C; // channels of the layer
    const int cl = lane < Cr ? lane : 0;
    // lanes that move data, as an EXEC mask (PAD: row operations are issued under it)
    const unsigned long long lmask = PAD ? __builtin_amdgcn_ballot_w64(lane < Cr) : ~0ull;
    const unsigned voff = (unsigned)((size_t)cl * H * W * sizeof(float)); // per-lane byte offset of its channel
    const char *xg = (const char *)xin + (size_t)b * Cr * H * W * sizeof(float);
    char *zg = (char *)zout + (size_t)b * Cr * H * W * sizeof(float);
    // Row operations of a step (G DMAs of x quads, G stores of z quads) are wave-uniform, but scalar arithmetic is
    // the most expensive thing a lone wave can issue (measured: 8.4 cycles per SALU instruction, 5.3 per VALU,
    // tools/issue_rate_probe.hip).  Lane j < 2G therefore computes operation j's addresses with vector
    // instructions, and each operation picks its values up with v_readlane.  Per-lane constants of that stage:
    const bool ro_dma = lane < G;
    const int ro_i = lane < G ? lane : (lane < 2 * G ? lane - G : 0);
    const int ro_hb = 4 * (wave + Cfg::NWAVES * ro_i); // first row of the operation's row class
    const int ro_ko = ro_dma ? 3 : -5;                 // first column of the quad = d + ko - h (rows h = d-1 mod 4 are due)
    const unsigned long long ro_base = ro_dma ? (unsigned long long)xg : (unsigned long long)zg;
    const int ro_lds = ro_dma ? (int)ldsbase + Cfg::OFF_XS : (int)ldsbase + Cfg::OFF_ZQ;
    // byte offset of the quad (row hr, columns wq..wq+3) in a stored channel plane = gbase + hr*grow + wq*gcol
    const int grow = rh ? -4 * W : 4 * W, gcol = rw ? -4 : 4;
    const int gbase = (rh ? (H - 1) * 4 * W : 0) + (rw ? (W - 4) * 4 : 0) + hoff * grow;
    float rmax = 0.f;       // max |r| this lane put into the ring: beyond the fp16 range the image is redone in fp32
    float zmax = 0.f;       // max |z| this lane stored (handed to the weight-gradient kernel as its prescale)
    int qprev = 0;          // in-row staging offset (parity, position in the quad) of the previous step's column
    float xscale = 1.f, zscale = 1.f; // scaled retries of an image whose r left the fp16 range
    // kernel arguments used in the loop, held in scalar registers (a reload would wait on the LDS counter)
    int Hs = Hp, Ws = W;
    asm volatile("" : "+s"(Hs), "+s"(Ws));

    // Rolling accumulators ("push" form): acc[k] collects everything the taps contribute to diagonal d+k; the
    // step that finishes diagonal d only adds the two taps whose source is r_{d-1}.
    floatx4 ahi[NTILE][Cfg::NACC], amid[NTILE][Cfg::NACC];
    // fragments of the source rows two up (dh = 2), carried to the next step.  Single-buffered: the reload is
    // issued after the MFMAs that read them have been issued (operands are read at issue).
    half8 F2h[NTILE][NQ], F2l[NTILE][NQ];
#pragma unroll
    for (int T = 0; T < NTILE; ++T) {
#pragma unroll
        for (int k = 0; k < Cfg::NACC; ++k) {
            ahi[T][k] = floatx4{0.f, 0.f, 0.f, 0.f};
            amid[T][k] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                F2h[T][q][j] = (_Float16)0.f;
                F2l[T][q][j] = (_Float16)0.f;
            }
    }

#ifdef IFL_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
    unsigned long long st_mask[4] = {0, 0, 0, 0}, st_cnt[4] = {0, 0, 0, 0};
    unsigned long long st_rt[4] = {__builtin_amdgcn_s_memrealtime(), 0, 0, 0};
    const unsigned long long st_begin = st_last;
#endif
    __syncthreads();

    // One step, specialised on the tiles in their active window (MASK) and on those that were active in the
    // previous step (PMASK: their z product is still to be written out).  Source diagonal d-1 is read from LDS
    // ONCE (three row-shifted fragment sets per tile); each tap pushes it into the accumulator of the diagonal
    // it lands on:
    //   leading  : the dh=2 taps of r_{d-2} (fragments kept in registers from the previous step) -- they fill
    //              the matrix pipe while this step's fragment reads are in flight,
    //   critical : the two taps whose target is diagonal d itself,
    //   trailing : the other taps of r_{d-1} and the z product.
    // One wave per SIMD: nothing overlaps unless the instruction stream interleaves it.  Everything that is not
    // on the chain is therefore cut into chunks that sit between MFMA groups (an MFMA occupies the matrix pipe
    // for 16 cycles, the wave issues ~3 other instructions meanwhile):
    //   leading  <- z of the previous step -> staging, completed z quads -> global (store role)
    //   critical <- address arithmetic + issue of the x quads needed three steps from now (DMA role)
    //   trailing <- the chain's epilogue r_d = x + acc -> split fp16 -> ring
    // row operations of step d, lane-parallel (see the per-lane constants above); computed one step ahead
    int ro_alo, ro_ahi, ro_loff, ro_okm;
    auto ro_stage = [&](const int d) {
        const int hr = ((d - 1) & 3) + ro_hb;
        const int wq = d + ro_ko - hr;
        const bool ok = hr < Hs && (unsigned)wq < (unsigned)Ws;
        const unsigned goff = (unsigned)(gbase + __mul24(hr, grow) + __mul24(wq, gcol));
        // (no quad due: the buffer's first quad -- a 32-bit select, then one add: the 64-bit select became a branch)
        const unsigned long long a = ro_base + (ok ? goff : 0u);
        ro_alo = (int)(unsigned)a;
        ro_ahi = (int)(unsigned)(a >> 32);
        ro_loff = ro_lds + __mul24(hr, Cfg::XROWB) + ((wq >> 2) & 1) * (C * 16);
        ro_okm = ok ? -1 : 0;
    };
    auto step = [&](auto mask_c, auto scaled_c, const int d) {
        constexpr int MASK = decltype(mask_c)::value;
        constexpr bool SCALED = decltype(scaled_c)::value; // a retry with x scaled down (see the sweep below)
        constexpr int NA = (MASK & 1) + ((MASK >> 1) & 1);
        constexpr int NDH = KH < 2 ? KH : 2;
        constexpr int PER = NA * NQ * 2; // reads per fragment set
        constexpr int NRD2 = (KH > 2 && MASK != 0) ? PER : 0;
        const int srcoff = ((d + 1) & 1) * SLOTB; // ring slot of diagonal d-1 (slot = diagonal mod 2)
        const int dstoff = (d & 1) * SLOTB;       // ring slot of diagonal d
        IFL_STAMP(7); // loop control
#ifdef IFL_STAMPS
        const unsigned long long st_t0 = st_last;
#endif
        auto ro_ptr = [&](int j) -> char * {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane(ro_alo, j), hi = (unsigned)__builtin_amdgcn_readlane(ro_ahi, j);
            return (char *)(((unsigned long long)hi << 32) | lo);
        };

        // The quads this wave DMA'd three steps ago must have landed before anyone reads them below.  Younger
        // vector-memory operations: the G DMAs of each of the steps d-2 and d-1 (always issued) and up to G stores
        // per step (masked off when no quad is due).  "All but the 2G youngest complete" therefore always covers
        // the DMA of step d-3 -- with stores in flight it also asks for a DMA that is two steps old, which has
        // landed long ago -- and never waits for a store of the previous step (a store takes ~2 us to retire).
        // Then the barrier: r of diagonal d-1 and the z staged in step d-1 are complete in LDS.
        wait_vm_then_barrier<2 * G>();
        IFL_STAMP(1); // wait + barrier

        // in-row staging offset of this step's column w = d - h: the same for all tiles (rows 16 apart)
        const int w0 = d - n;
        const int qcur = ((w0 >> 2) & 1) * (C * 16) + (rw ? 3 - (w0 & 3) : (w0 & 3)) * 4;

        // ---- LDS requests, oldest first: staged z quads (store role), x of this step, fragments.  They are
        //      numbered so that they can be issued a few at a time between the first leading MFMAs. ----------------
        // store role: rows h = d-1 (mod 4) completed a quad of z with diagonal d-2 (staged one step ago)
        floatx4_ sv[G];
        floatx2 xq[NTILE][2];
        half8 Fh[NTILE][2][NQ], Fl[NTILE][2][NQ];
        constexpr int NREQ = G + (MASK != 0 ? 2 * NA + NDH * PER : 0);
        auto request = [&](int j) {
            int c = 0;
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (c++ == j) // (any quad: the read stays inside the staging area)
                    lds_read_f32x4(sv[i], (unsigned)__builtin_amdgcn_readlane(ro_loff, G + i) + cl * 16);
            if constexpr (MASK != 0) {
#pragma unroll
                for (int T = 0; T < NTILE; ++T)
                    if (MASK & (1 << T)) {
                        const unsigned xa = ldsbase + Cfg::OFF_XS + xadr[T] + qcur;
                        if (c++ == j) lds_read2_f32<0>(xq[T][0], xa);
                        if (c++ == j) lds_read2_f32<8>(xq[T][1], xa);
                    }
#pragma unroll
                for (int dh = 0; dh < NDH; ++dh)
#pragma unroll
                    for (int T = 0; T < NTILE; ++T)
                        if (MASK & (1 << T)) {
                            const unsigned fa = radr[T][dh] + srcoff;
                            if (c++ == j) lds_read_b128_o<0>(Fh[T][dh][0], fa);
                            if (c++ == j) lds_read_b128_o<4 * 256>(Fl[T][dh][0], fa);
                            if constexpr (NQ == 2) {
                                if (c++ == j) lds_read_b128_o<8 * 256>(Fh[T][dh][1], fa);
                                if (c++ == j) lds_read_b128_o<12 * 256>(Fl[T][dh][1], fa);
                            }
                        }
            }
        };
        constexpr int NYOUNG = (MASK != 0 ? 2 * NA + NDH * PER : 0); // LDS requests younger than the staged quads
        __builtin_amdgcn_sched_barrier(0);

        // A(t) x {hi, lo} fragments of all active tiles -> accumulator tgt; dependent MFMAs kept apart
        // order: per k-step the hi.hi and hi.lo products of all tiles, then the lo.hi products
        // init: the group is the first to touch its accumulator (a diagonal's first contribution): start from zero
        auto mf_one = [&](int t, const auto &fh, const auto &fl, int dh, int tgt, int k, bool init = false) {
            int c = 0;
            const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int T = 0; T < NTILE; ++T)
                    if (MASK & (1 << T))
                        if (c++ == k)
                            ahi[T][tgt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][0], fh(T, dh, q),
                                                                                 (init && q == 0) ? zero : ahi[T][tgt], 0, 0, 0);
#pragma unroll
                for (int T = 0; T < NTILE; ++T)
                    if (MASK & (1 << T))
                        if (c++ == k)
                            amid[T][tgt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][0], fl(T, dh, q),
                                                                                  (init && q == 0) ? zero : amid[T][tgt], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int T = 0; T < NTILE; ++T)
                    if (MASK & (1 << T))
                        if (c++ == k)
                            amid[T][tgt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][1], fh(T, dh, q), amid[T][tgt], 0, 0, 0);
        };
        auto mf = [&](int t, const auto &fh, const auto &fl, int dh, int tgt, bool init = false) {
#pragma unroll
            for (int k = 0; k < 3 * NQ * NA; ++k) mf_one(t, fh, fl, dh, tgt, k, init);
        };
        auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
        // scheduling pattern for the region since the last fence: `lead` MFMAs, then NM x (1 MFMA, NV others)
        // (each region gets its own pipeline id: groups of one id are matched across the whole basic block)
        auto weave = [&](auto id_c, auto lead_c, auto nm_c, auto nv_c) {
            constexpr int ID = decltype(id_c)::value;
            constexpr int LEAD = decltype(lead_c)::value, NM = decltype(nm_c)::value, NV = decltype(nv_c)::value;
            if constexpr (LEAD > 0) __builtin_amdgcn_sched_group_barrier(0x008, LEAD, ID);
#pragma unroll
            for (int k = 0; k < NM; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, ID);
                __builtin_amdgcn_sched_group_barrier(0x296, NV, ID); // VALU | SALU | VMEM | DS
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto f2h = [&](int T, int, int q) -> const half8 & { return F2h[T][q]; };
        auto f2l = [&](int T, int, int q) -> const half8 & { return F2l[T][q]; };
        auto f1h = [&](int T, int dh, int q) -> const half8 & { return Fh[T][dh][q]; };
        auto f1l = [&](int T, int dh, int q) -> const half8 & { return Fl[T][dh][q]; };

        // ---- chunk: z of diagonal d-1 -> staging, at that column's in-row offset (columns outside the image land
        //      in quads that are not live: before a row's first quad, or in the parity its last quad does not use)
        floatx4 zh[NTILE], zm[NTILE];
        auto chunk_zstage = [&]() {
#pragma unroll
            for (int T = 0; T < NTILE; ++T)
                if (MASK & (1 << T)) {
                    unsigned char *zp = zq + xadr[T] + qprev;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *(float *)(zp + r * 16) = SCALED ? (zh[T][r] + zm[T][r] * LO_INV) * zscale : zh[T][r] + zm[T][r] * LO_INV;
                }
        };
        // ---- chunk: store role ----------------------------------------------------------------------------
        // A row without a finished quad issues its store with all lanes masked off (EXEC = 0, no branch: a branch
        // would split the block the MFMAs are woven into).  Whether such a store ticks the vector-memory counter
        // does not matter: the wait before the barrier counts the DMAs only (see there).
        auto chunk_store = [&]() {
            lgkm_wait_n(NYOUNG); // the staged quads have landed (they are the oldest requests)
#pragma unroll
            for (int i = 0; i < G; ++i) {
                char *dstp = ro_ptr(G + i); // wave-uniform: the quad's place in z
                const unsigned okm = (unsigned)__builtin_amdgcn_readlane(ro_okm, G + i);
                const unsigned long long em = (((unsigned long long)okm << 32) | okm) & lmask;
                unsigned long long saved;
                if (C == 64 || PAD || lane < C) {
                    asm volatile("s_mov_b64 %0, exec\n\t"
                                 "s_and_b64 exec, exec, %1\n\t"
                                 "global_store_dwordx4 %2, %3, %4\n\t"
                                 "s_mov_b64 exec, %0\n\t"
                                 "s_nop 0"
                                 : "=&s"(saved)
                                 : "s"(em), "v"(voff), "v"(sv[i]), "s"(dstp)
                                 : "memory", "scc");
                    // (padded lanes hold channel 0's quad again: harmless for the maximum)
                    const float m = fmaxf(fmaxf(fabsf(sv[i][0]), fabsf(sv[i][1])), fmaxf(fabsf(sv[i][2]), fabsf(sv[i][3])));
                    zmax = fmaxf(zmax, __uint_as_float(__float_as_uint(m) & okm));
                }
            }
        };
        // ---- chunk: DMA role, x quads needed three steps from now (unconditional: exact VM operation count) -
        auto chunk_dma = [&](int i) {
            const char *src = ro_ptr(i); // wave-uniform: the quad, or the image's first one when none is due
            const unsigned dst = (unsigned)__builtin_amdgcn_readlane(ro_loff, i); // lane c lands at +16c
            if constexpr (!PAD) {
                if (C == 64 || lane < C)
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + voff),
                                                     (void __attribute__((address_space(3))) *)(size_t)dst, 16, 0, 0);
            } else {
                // the same LDS-DMA under the lane mask (padded channels must stay zero in the staging area)
                unsigned long long saved;
                asm volatile("s_mov_b32 m0, %1\n\t"
                             "s_mov_b64 %0, exec\n\t"
                             "s_and_b64 exec, exec, %2\n\t"
                             "global_load_lds_dwordx4 %3, %4\n\t"
                             "s_mov_b64 exec, %0"
                             : "=&s"(saved)
                             : "s"(dst), "s"(lmask), "v"(voff), "s"(src)
                             : "memory", "scc", "m0");
            }
        };
        // ---- chunk: epilogue of tile T (the chain): r_d = x + acc[0] -> split fp16 -> ring; then the accumulators
        //      rotate: diagonal d+1 becomes the head, a fresh one joins for d+3 -------------------------------
        auto chunk_epilogue = [&](int T) {
            const int w = w0 - 16 * T;
            const bool valid = hval[T] && (unsigned)w < (unsigned)Ws;
            float rv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                rv[r] = (SCALED ? xq[T][r >> 1][r & 1] * xscale : xq[T][r >> 1][r & 1]) + ahi[T][0][r] + amid[T][0][r] * LO_INV;
            half4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const _Float16 h16 = (_Float16)rv[r];
                hi[r] = h16;
                lo[r] = (_Float16)((rv[r] - (float)h16) * LO_SCALE);
            }
            // lanes outside the image write to the dump (no branch: the MFMAs around this chunk keep flowing)
            unsigned char *rp = valid ? ring + dstoff + wadr[T] : lds + Cfg::OFF_DUMP + tid * 8;
            *(half4 *)rp = hi;
            *(half4 *)(rp + 4 * 256) = lo;
            const float m = fmaxf(fmaxf(fabsf(rv[0]), fabsf(rv[1])), fmaxf(fabsf(rv[2]), fabsf(rv[3])));
            rmax = valid ? fmaxf(rmax, m) : rmax;
        };

        using N0 = std::integral_constant<int, 0>;
        constexpr int GM = 3 * NQ * NA; // MFMAs of one tap group
        using NGM = std::integral_constant<int, GM>;
        if constexpr (MASK == 0) {
#pragma unroll
            for (int j = 0; j < NREQ; ++j) request(j);
            fence();
            chunk_store();
#pragma unroll
            for (int i = 0; i < G; ++i) chunk_dma(i);
            ro_stage(d + 1);
            fence();
        } else {
            // ---- leading: taps (2, dw) of r_{d-2}; their targets are diagonals d-2+2+dw = d + dw -----------
            if constexpr (KH > 2) {
                // the first two groups carry the LDS requests (inline asm: placed by hand, one behind each MFMA:
                // measured 21 cycles for MFMA + ds_read_b128, 36.5 for MFMA + two)
                constexpr int NLEADM = (KW > 1 ? 2 : 1) * GM;
                constexpr int RPM = (NREQ + NLEADM - 1) / NLEADM;
#pragma unroll
                for (int k = 0; k < NLEADM; ++k) {
                    if (k < GM) mf_one(2 * KW + 0, f2h, f2l, 0, 0, k);
                    else mf_one(2 * KW + 1, f2h, f2l, 0, 1, k - GM);
                    // (MFMAs are pure: tie the result to an opaque statement, or they sink below the requests)
#pragma unroll
                    for (int T = 0; T < NTILE; ++T)
                        if (MASK & (1 << T)) asm volatile("" : "+a"(ahi[T][k < GM ? 0 : 1]), "+a"(amid[T][k < GM ? 0 : 1]));
                    fence();
#pragma unroll
                    for (int j = k * RPM; j < (k + 1) * RPM && j < NREQ; ++j) request(j);
                    fence();
                }
                if constexpr (KW > 2) mf(2 * KW + 2, f2h, f2l, 0, 2, true); // first contribution to diagonal d+2
                chunk_store();
                if constexpr (KH < 2) {
#pragma unroll
                    for (int i = 0; i < G; ++i) chunk_dma(i);
                }
                weave(std::integral_constant<int, 2>{}, N0{}, NGM{}, std::integral_constant<int, 3>{});
            } else {
#pragma unroll
                for (int j = 0; j < NREQ; ++j) request(j);
                fence();
                chunk_store();
#pragma unroll
                for (int i = 0; i < G; ++i) chunk_dma(i);
                fence();
            }
            IFL_STAMP(2); // read issue + leading MFMAs (+ all reads landed, when stamping)
            // ---- critical: taps (0,1) and (1,0) of r_{d-1} -> diagonal d -----------------------------------------
            // The dh=2 fragments of r_{d-1} (next step's leading operands; single-buffered: every MFMA that reads the
            // old ones has been issued) are requested between the first critical MFMAs.
            auto request2 = [&](int j) {
                int c = 0;
#pragma unroll
                for (int T = 0; T < NTILE; ++T)
                    if (MASK & (1 << T)) {
                        const unsigned fa = radr[T][KH > 2 ? 2 : 0] + srcoff;
                        if (c++ == j) lds_read_b128_o<0>(F2h[T][0], fa);
                        if (c++ == j) lds_read_b128_o<4 * 256>(F2l[T][0], fa);
                        if constexpr (NQ == 2) {
                            if (c++ == j) lds_read_b128_o<8 * 256>(F2h[T][1], fa);
                            if (c++ == j) lds_read_b128_o<12 * 256>(F2l[T][1], fa);
                        }
                    }
            };
            if constexpr (KW > 1) {
                lgkm_wait_n((NDH - 1) * PER); // the dh=0 fragments of all tiles have landed
#pragma unroll
                for (int k = 0; k < GM; ++k) {
                    mf_one(1, f1h, f1l, 0, 0, k);
                    if constexpr (KH > 2) {
#pragma unroll
                        for (int T = 0; T < NTILE; ++T)
                            if (MASK & (1 << T)) asm volatile("" : "+a"(ahi[T][0]), "+a"(amid[T][0]));
                        fence();
                        if (k < PER) request2(k);
                        fence();
                    }
                }
                fence();
            }
            if constexpr (KH > 1) {
                lgkm_wait_n(NRD2); // ... and the dh=1 fragments
                mf(KW, f1h, f1l, 1, 0);
                if constexpr (KH > 2) { // (the DMA issue rides here: the last leading group already carries the stores)
#pragma unroll
                    for (int i = 0; i < G; ++i) chunk_dma(i);
                    weave(std::integral_constant<int, 7>{}, N0{}, NGM{}, std::integral_constant<int, 2>{});
                } else {
                    fence();
                }
            }
            IFL_STAMP(3); // critical MFMAs issued
            // ---- trailing: the remaining taps of r_{d-1} with dh < 2 (targets d+1, d+2) and z_{d-1} = L^-1 r_{d-1},
            //      the epilogue of the chain woven into the first groups.  The accumulators rotate first (diagonal
            //      d+1 becomes the head, a fresh one joins for d+3); the epilogue works on the old head.
            floatx4 head_hi[NTILE], head_mid[NTILE];
#pragma unroll
            for (int T = 0; T < NTILE; ++T)
                if (MASK & (1 << T)) {
                    head_hi[T] = ahi[T][0];
                    head_mid[T] = amid[T][0];
                    ahi[T][0] = ahi[T][1];
                    amid[T][0] = amid[T][1];
                    ahi[T][1] = ahi[T][2];
                    amid[T][1] = amid[T][2]; // ([2] is dead until the group that opens the next diagonal initialises it)
                }
            auto epi = [&](int T) {
                if (MASK & (1 << T)) {
                    const floatx4 sh = ahi[T][0], sm = amid[T][0]; // keep the rotated values out of the chunk's way
                    ahi[T][0] = head_hi[T];
                    amid[T][0] = head_mid[T];
                    chunk_epilogue(T);
                    ahi[T][0] = sh;
                    amid[T][0] = sm;
                }
            };
            // MFMA groups in issue order: the z product first (its results are staged within this step), then the
            // taps.  Riders, one per group: epilogue of tile 0, of tile 1, z staging, next step's row operations;
            // riders without a group run behind the last one.
            int ntr = 0;
            auto rider = [&](int k) {
                if (k == 0) epi(0);
                if (k == 1 && NTILE > 1) epi(1);
                if (k == 2) chunk_zstage();
                if (k == 3) ro_stage(d + 1);
            };
            auto after_group = [&]() {
                constexpr int LEADM = GM < 4 ? GM : 4; // the head's last MFMA needs a few issue slots to finish
                rider(ntr);
                if (ntr == 0)
                    weave(std::integral_constant<int, 3>{}, std::integral_constant<int, LEADM>{},
                          std::integral_constant<int, GM - LEADM>{}, std::integral_constant<int, 5>{});
                else if (ntr == 1)
                    weave(std::integral_constant<int, 4>{}, N0{}, NGM{}, std::integral_constant<int, 4>{});
                else if (ntr == 2)
                    weave(std::integral_constant<int, 5>{}, N0{}, NGM{}, std::integral_constant<int, 3>{});
                else
                    weave(std::integral_constant<int, 6>{}, N0{}, NGM{}, std::integral_constant<int, 3>{});
                ++ntr;
            };
            // z product
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int T = 0; T < NTILE; ++T)
                    if (MASK & (1 << T)) {
                        const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
                        zh[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[NS - 1][q][0], Fh[T][0][q], q ? zh[T] : zero, 0, 0, 0);
                    }
#pragma unroll
                for (int T = 0; T < NTILE; ++T)
                    if (MASK & (1 << T)) {
                        const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
                        zm[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[NS - 1][q][0], Fl[T][0][q], q ? zm[T] : zero, 0, 0, 0);
                    }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int T = 0; T < NTILE; ++T)
                    if (MASK & (1 << T))
                        zm[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[NS - 1][q][1], Fh[T][0][q], zm[T], 0, 0, 0);
            after_group();
#pragma unroll
            for (int dh = 0; dh < NDH; ++dh)
#pragma unroll
                for (int dw = 0; dw < KW; ++dw)
                    if (dh + dw >= 2) {
                        // (without a dh=2 row the farthest tap is this one: it opens its diagonal)
                        mf(dh * KW + dw, f1h, f1l, dh, dh + dw - 2, KH < 3 && dh + dw == KH + KW - 2);
                        after_group();
                    }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k >= ntr) rider(k);
            fence();
            IFL_STAMP(5); // trailing MFMAs issued
        }

        qprev = qcur;
        IFL_STAMP(6); // bookkeeping
#ifdef IFL_STAMPS
        st_mask[MASK] += st_last - st_t0;
        st_cnt[MASK] += 1;
#endif
    };

    // A tile's window runs from two steps before its first pixel (the dh=2 fragments and the early pushes) to
    // the step after its last one (the z product of the last diagonal): the sets of active tiles come in the
    // order {}, {0}, {0,1}, {1}, {} -- one loop per set, so that no control flow merges inside a step.  The last
    // step stores the quads staged by the one before.
    auto sweep = [&](auto scaled_c) {
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        const int last0 = (15 + W - 1 < ND - 1 ? 15 + W - 1 : ND - 1) + 1; // last step of tile 0
        int d = -3;
        ro_stage(d);
        for (; d < -2; ++d) step(I0{}, scaled_c, d);
        if constexpr (NTILE == 2) {
            using I2 = std::integral_constant<int, 2>;
            using I3 = std::integral_constant<int, 3>;
            for (; d < 14; ++d) step(I1{}, scaled_c, d);
            for (; d <= last0; ++d) step(I3{}, scaled_c, d);
            for (; d <= ND; ++d) step(I2{}, scaled_c, d);
        } else {
            for (; d <= last0; ++d) step(I1{}, scaled_c, d);
        }
        for (; d <= ND + 2; ++d) step(I0{}, scaled_c, d);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every DMA and store of the sweep has retired
    };
    sweep(std::false_type{});

#ifdef IFL_STAMPS
    if (g_stamps && b == 0 && lane == 0)
        for (int k = 0; k < 8; ++k) g_stamps[wave * 8 + k] = st_acc[k];
    if (g_stamps && b == 0 && lane == 0 && wave == 0) {
        for (int k = 0; k < 4; ++k) {
            g_stamps[64 + k] = st_mask[k];
            g_stamps[68 + k] = st_cnt[k];
        }
        st_rt[2] = __builtin_amdgcn_s_memrealtime();
        for (int k = 0; k < 4; ++k) g_stamps[72 + k] = st_rt[k];
        g_stamps[78] = st_begin;
        g_stamps[79] = __builtin_amdgcn_s_memtime();
    }
#endif
    // Split fp16 cannot hold |r| >= 65504, and a badly conditioned operator grows r along the sweep.  Such an image
    // is swept once more by the same workgroup with x scaled down by 2^-12 and z scaled back up: the recurrence is
    // linear, inputs of order one still split into a normal fp16 hi and a lo that keeps the rest (22 bits), and r may
    // now reach 2.4e8.  A second, deeper rescale would push the early (small) diagonals into fp16 denormals, whose
    // rounding the growth then amplifies (measured 1e-4 at max|z| = 1e11): beyond the first rescale, or when the input
    // is not finite, the exact fp32 body takes over (general scan body, right-fold form, fp32 copy of the same folded
    // weights): slow, but never a silent Inf/NaN where the exact solver is finite.  flags[] records what happened
    // (diagnostics only: 0, 1 = rescaled, +4 = fp32).
    int redo = __syncthreads_or(rmax < 6.0e4f ? 0 : 1);
    int attempts = 0;
    while (redo && attempts < 1) {
        ++attempts;
        xscale *= 1.0f / 4096.0f;
        zscale *= 4096.0f;
        {
            const floatx4 zz = {0.f, 0.f, 0.f, 0.f};
            for (int i = tid * 16; i < Cfg::RINGB; i += Cfg::THREADS * 16) *(floatx4 *)(lds + i) = zz;
        }
#pragma unroll
        for (int T = 0; T < NTILE; ++T) {
#pragma unroll
            for (int k = 0; k < Cfg::NACC; ++k) {
                ahi[T][k] = floatx4{0.f, 0.f, 0.f, 0.f};
                amid[T][k] = floatx4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    F2h[T][q][j] = (_Float16)0.f;
                    F2l[T][q][j] = (_Float16)0.f;
                }
        }
        rmax = 0.f;
        zmax = 0.f;
        qprev = 0;
        __syncthreads();
        sweep(std::true_type{});
        redo = __syncthreads_or(rmax < 6.0e4f ? 0 : 1);
    }
    if (tid == 0) flags[b] = attempts + (redo ? 4 : 0); // every workgroup owns its word: no clearing pass needed
    if (redo) {
        scan_general_body<Cfg::THREADS>(xin, wf32, zout, geom, rh, rw, 1, (float *)lds, b, tid);
        if (amax) { // the quads stored above are void: take the maximum of what the redo wrote
            __syncthreads();
            const float *zi = zout + (size_t)b * Cr * H * W;
            zmax = 0.f;
            for (int i = tid; i < Cr * H * W; i += Cfg::THREADS) zmax = fmaxf(zmax, fabsf(zi[i]));
        }
    }
    if (amax) {
        for (int o = 32; o > 0; o >>= 1) zmax = fmaxf(zmax, __shfl_down(zmax, o, 64));
        if (lane == 0) atomicMax(amax, __float_as_uint(zmax)); // one atomic per wave; max is order-independent
    }
    return 0;
}

template <int C, int KH, int KW, int NTILE, bool PAD>
__global__ __launch_bounds__(64 * (C / 16)) void k_scan_mfma(const float *__restrict__ xin, float *__restrict__ zout,
                                                             const half8 *__restrict__ apack, int H, int W, int rh, int rw,
                                                             int *__restrict__ flags, const float *__restrict__ wf32,
                                                             Geom geom, unsigned *__restrict__ amax)
{
    scan_body<C, KH, KW, NTILE, PAD>(xin, zout, apack, H, W, rh, rw, flags, wf32, geom, amax, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Fused fold + pack for the MFMA scan: one launch builds the A fragments in the exact per-lane
// register image.
//   slot s < NT-1 : -(W_t L^-1)          (t = s+1; transposed: -(W_t^T L^-T))
//   slot NT-1     :  L^-1                (transposed: L^-T)
// apack[wv][s][q][hl][lane][j] (fp16),  lane = m + 16*gk holds row c = 16wv+m, k = 32q + 8gk + j.
//
// Grid = NT slots x C/16 row groups.  Every workgroup inverts the 16x16 DIAGONAL blocks of L in LDS (fp64, by
// substitution: 16 dependent steps) and then obtains its 16 rows of the slot by a block triangular solve
// X L = W_t on the fp64 matrix cores (C/16 dependent block steps, everything in registers) -- which cuts the
// 64-step serial substitution of the exact solver's channel loop (solve_mc.py:96-109) to ~16 + C/16 short
// dependent stages and never forms the full inverse -- and splits them into fp16 hi / lo*2^11.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t w_index2(int co, int ci, int dh, int dw, int C, int KH, int KW, int flipH,
                                           int flipW)
{
    int kh = KH - 1 - dh, kw = KW - 1 - dw;
    if (flipH) kh = KH - 1 - kh;
    if (flipW) kw = KW - 1 - kw;
    return (((size_t)co * C + ci) * KH + kh) * KW + kw;
}

template <int C> __global__ __launch_bounds__(256) void k_foldpack(FoldJobs jobs)
{
    // blockIdx.z = layer of an inverse-flow block (1 for a single layer)
    const FoldJob &job = jobs.job[blockIdx.z];
    const float *__restrict__ w = job.w;
    _Float16 *apack0 = (_Float16 *)job.out0, *apack1 = (_Float16 *)job.out1;
    float *wf0 = job.wf0, *wf1 = job.wf1;
    const Geom g = job.g;
    const int transposed0 = job.transposed;
    unsigned *zero0 = job.zero0, *zero1 = job.zero1;
    // blockIdx.y = 0: direction `transposed0`; blockIdx.y = 1: the other direction (forward call that also
    // prepares the adjoint for the backward).  Runs ahead of the scans on the stream: clears their absmax words.
    const int transposed = blockIdx.y == 0 ? transposed0 : 1 - transposed0;
    _Float16 *apack = blockIdx.y == 0 ? apack0 : apack1;
    float *wf32 = blockIdx.y == 0 ? wf0 : wf1;
    constexpr int mode = 0;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (zero0) *zero0 = 0u;
        if (zero1) *zero1 = 0u;
    }
#ifdef IFL_STAMPS
    unsigned long long ft[8], fl_ = __builtin_amdgcn_s_memtime();
    int fk = 0;
#define IFL_FSTAMP()                                                   \
    do {                                                               \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            \
        ft[fk++] = t_ - fl_;                                           \
        fl_ = t_;                                                      \
    } while (0)
#else
#define IFL_FSTAMP() \
    do {             \
    } while (0)
#endif
    constexpr int NBK = C / 16, XP = C + 1; // XP: fp64 row pitch (conflict-free row- and column-wise)
    constexpr int LP = C + 1; // row pitch of L: rows a wave reads together fall into different banks
    __shared__ float sL[C * LP];
    __shared__ double sX[C * XP];
    __shared__ __attribute__((aligned(16))) float sW[16 * C];
    const int tid = threadIdx.x;
    const int NT = g.KH * g.KW, RG = C / 16, NQ = C / 32;
    const int s = blockIdx.x / RG, rgrp = blockIdx.x % RG;

    // ---- L (effective diagonal tap) and this workgroup's 16 rows of W_t ----------------------------
    // All gathers (stride KH*KW floats) are issued before any is used: unconditional loads into registers, the
    // triangle / unit diagonal applied afterwards (a conditional load per iteration made this phase a chain of
    // ~20 exposed memory latencies: 15.8k of the kernel's 41k cycles).
    {
        constexpr int NL = C * C / 256, NWL = 16 * C / 256;
        const int Cr = g.C; // channels of the layer (<= C: the rest is padding, identity in L and zero in W_t)
        float vl[NL], vw[NWL > 0 ? NWL : 1];
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int idx = tid + 256 * u, i = idx / C, k = idx % C;
            const bool in = i < Cr && k < Cr;
            vl[u] = w[w_index2(in ? i : 0, in ? k : 0, 0, 0, Cr, g.KH, g.KW, g.flipH, g.flipW)];
        }
        const int t = s < NT - 1 ? s + 1 : 1 % (NT > 1 ? NT : 2), dh = t / g.KW, dw = t % g.KW; // (last slot: a valid tap, unused)
#pragma unroll
        for (int u = 0; u < NWL; ++u) {
            const int idx = tid + 256 * u, cl = idx / C, m = idx % C, c = 16 * rgrp + cl;
            const bool in = c < Cr && m < Cr;
            const int cc = in ? c : 0, mm = in ? m : 0;
            const float v = transposed ? w[w_index2(mm, cc, NT > 1 ? dh : 0, NT > 1 ? dw : 0, Cr, g.KH, g.KW, g.flipH, g.flipW)]
                                       : w[w_index2(cc, mm, NT > 1 ? dh : 0, NT > 1 ? dw : 0, Cr, g.KH, g.KW, g.flipH, g.flipW)];
            vw[u] = in ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int idx = tid + 256 * u, i = idx / C, k = idx % C;
            const bool in = i < Cr && k < Cr;
            sL[i * LP + k] = k < i ? (in ? vl[u] : 0.f) : (k == i ? ((g.general_diag && in) ? vl[u] : 1.f) : 0.f);
        }
        if (s < NT - 1) {
#pragma unroll
            for (int u = 0; u < NWL; ++u) {
                const int idx = tid + 256 * u;
                sW[(idx % C) * 16 + idx / C] = vw[u]; // [m][row]: a thread's four rows are one 16-byte read
            }
        }
    }
    for (int idx = tid; idx < C * XP; idx += 256) sX[idx] = 0.0;
    __syncthreads();
    IFL_FSTAMP(); // 0: loads

    // ---- diagonal blocks of L^-1: column j by forward substitution inside its 16x16 block; the column
    //      lives in registers (fully unrolled), L comes from LDS and does not depend on the chain ------
    if (tid < C) {
        const int j = tid, r0 = (j / 16) * 16, jj = j % 16;
        double col[16];
        // reciprocals of the block's diagonal first: a division inside the substitution chain is ~40 dependent
        // fp64 instructions per row (half of this phase)
        double rdiag[16];
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) rdiag[ii] = 1.0 / (double)sL[(r0 + ii) * LP + r0 + ii];
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) {
            double a0 = (ii == jj) ? 1.0 : 0.0, a1 = 0.0;
#pragma unroll
            for (int kk = 0; kk < 16; kk += 2) {
                if (kk < ii) a0 -= (double)sL[(r0 + ii) * LP + r0 + kk] * (kk >= jj ? col[kk] : 0.0);
                if (kk + 1 < ii) a1 -= (double)sL[(r0 + ii) * LP + r0 + kk + 1] * (kk + 1 >= jj ? col[kk + 1] : 0.0);
            }
            col[ii] = ii >= jj ? (a0 + a1) * rdiag[ii] : 0.0;
            sX[(r0 + ii) * XP + j] = col[ii];
        }
    }
    __syncthreads();
    IFL_FSTAMP(); // 1: diagonal blocks

    // ---- this workgroup's 16 rows of the slot by a block triangular SOLVE (fp64 matrix cores), not by a product with
    //      the full inverse: X L = R (R = the 16 rows of W_t, or of the identity for the last slot), block by block:
    //          normal      X_j = (R_j - sum_{k>j} X_k L_kj) L_jj^-1       j = NBK-1 .. 0
    //          transposed  X_j = (R_j - sum_{k<j} X_k L_jk^T) L_jj^-T     j = 0 .. NBK-1     (X = W_t^T L^-T)
    //      Only the diagonal blocks of L^-1 (above) are needed.  The solve runs on the TRANSPOSES Y_j = X_j^T, so that
    //      every intermediate is a right-hand operand: v_mfma_f64_16x16x4_f64 layouts (tools/mfma_f64_layout_probe.hip):
    //      A[i = l%16][k = l/16], B[k = l/16][j = l%16], D[i = 4v + l/16][j = l%16] in register v -- register kc of a
    //      result IS the B operand of k-chunk kc of the next product, nothing leaves the registers:
    //          Y_j = Dinv_jj^T (R_j^T - sum_k L_kj^T Y_k)   resp.   Y_j = Dinv_jj (R_j^T - sum_k L_jk Y_k)
    //      One wave: 40 dependent-ish MFMAs instead of the three barrier-separated levels of the off-diagonal inverse plus
    //      a 16-deep product chain per wave.  Y_j[4v + lk][li] = X[row li][column 16j + 4v + lk].
    typedef double doublex4 __attribute__((ext_vector_type(4)));
    const int wvf = tid / 64, lf = tid % 64, li = lf % 16, lk = lf / 16;
    if (wvf == 0) {
        doublex4 Ys[NBK]; // by solve step (compile-time index); step p solved block (transposed ? p : NBK-1-p)
        // every A operand (entries of L and of the diagonal blocks of L^-1) up front: none depends on the chain, and
        // an LDS load + conversion in front of each MFMA would sit on it
        double aL[NBK][NBK][4], aD[NBK][4];
#pragma unroll
        for (int step = 0; step < NBK; ++step) {
            const int j = transposed ? step : NBK - 1 - step;
#pragma unroll
            for (int p = 0; p < NBK; ++p)
                if (p < step) {
                    const int k = transposed ? p : NBK - 1 - p;
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc)
                        aL[step][p][kc] = transposed ? -(double)sL[(16 * j + li) * LP + 16 * k + 4 * kc + lk]
                                                     : -(double)sL[(16 * k + 4 * kc + lk) * LP + 16 * j + li];
                }
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                aD[step][kc] = transposed ? sX[(16 * j + li) * XP + 16 * j + 4 * kc + lk]
                                          : sX[(16 * j + 4 * kc + lk) * XP + 16 * j + li];
        }
#pragma unroll
        for (int step = 0; step < NBK; ++step) {
            const int j = transposed ? step : NBK - 1 - step;
            // acc = R_j^T (D layout: element (4v + lk, li) = R[row li][column 16j + 4v + lk])
            doublex4 acc;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int col = 16 * j + 4 * v + lk;
                acc[v] = s == NT - 1 ? (col == 16 * rgrp + li ? 1.0 : 0.0) : (double)sW[col * 16 + li];
            }
            // acc -= sum_k (L_kj^T | L_jk) Y_k over the blocks solved so far
            doublex4 parts[NBK];
#pragma unroll
            for (int p = 0; p < NBK; ++p) {
                if (p < step) {
                    // A[i = li][kk = 4kc + lk] = -(L_kj^T)[i][kk] = -L[16k + kk][16j + i]   (normal)
                    //                          = -L_jk[i][kk]     = -L[16j + i][16k + kk]   (transposed)
                    // (its own accumulator per solved block: the products of different blocks are independent chains)
                    {
                        doublex4 part = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int kc = 0; kc < 4; ++kc) part = __builtin_amdgcn_mfma_f64_16x16x4f64(aL[step][p][kc], Ys[p][kc], part, 0, 0, 0);
                        parts[p] = part;
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < NBK; ++p)
                if (p < step) acc += parts[p];
            // Y_j = (Dinv_jj^T | Dinv_jj) acc
            doublex4 res = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) res = __builtin_amdgcn_mfma_f64_16x16x4f64(aD[step][kc], acc[kc], res, 0, 0, 0);
            Ys[step] = res;
        }
        IFL_FSTAMP(); // 2: block solve
        // the 16 x C result (fp32, signed) goes to LDS (sW is free again) so that all four waves can pack it
#pragma unroll
        for (int p = 0; p < NBK; ++p)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int j = transposed ? p : NBK - 1 - p;
                sW[li * C + 16 * j + 4 * v + lk] = (s == NT - 1 ? 1.f : -1.f) * (float)Ys[p][v];
            }
    }
    __syncthreads();
    // ---- split, pack: a thread owns (row cl, 8 consecutive columns) = one 16-byte piece of the hi plane and one of the
    //      lo plane of the A-fragment image; the fp32 copy for the fallback scan is written row by row
    if (tid < 16 * (C / 8)) {
        const int cl = tid % 16, k8 = tid / 16; // columns 8 k8 .. 8 k8 + 7
        const int q = k8 / 4, gk = k8 % 4;
        half8 hi8, lo8;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float val = sW[cl * C + 8 * k8 + jj];
            const _Float16 hi = (_Float16)val;
            hi8[jj] = hi;
            lo8[jj] = (_Float16)((val - (float)hi) * LO_SCALE);
        }
        const size_t base = ((((size_t)rgrp * NT + s) * NQ + q) * 2) * 64 * 8;
        *(half8 *)(apack + base + (size_t)(cl + 16 * gk) * 8) = hi8;
        *(half8 *)(apack + base + (size_t)64 * 8 + (size_t)(cl + 16 * gk) * 8) = lo8;
    }
    if (wf32)
        for (int idx = tid; idx < 16 * C; idx += 256) {
            const int cl = idx % 16, kc = idx / 16, c = 16 * rgrp + cl;
            // fp32 copy [slot][kc][c] of the layer's own channels for the fp32 fallback scan
            if (kc < g.C && c < g.C) wf32[((size_t)s * g.C + kc) * g.C + c] = sW[cl * C + kc];
        }
    IFL_FSTAMP(); // 3: pack
    IFL_FSTAMP(); // 3: product + pack
#ifdef IFL_STAMPS
    if (g_stamps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0)
        for (int k = 0; k < 4; ++k) g_stamps[160 + k] = ft[k];
#endif
}

// channel count the MFMA kernels are instantiated for: the layer's own, padded up to 32 or 64
int mfma_padded_channels(int C) { return C <= 32 ? 32 : 64; }

size_t scan_mfma_pack_bytes(const Geom &g)
{
    const size_t ct = (size_t)mfma_padded_channels(g.C);
    return (size_t)g.KH * g.KW * ct * ct * 2 * sizeof(_Float16);
}

bool scan_mfma_supported(const Geom &g, const void *x, const void *z)
{
    if (g.C < 9 || g.C > 64) return false; // (a handful of channels: the VALU scan wins)
    if (!((g.KH == 3 && g.KW == 3) || (g.KH == 2 && g.KW == 2))) return false;
    if (g.W % 4 != 0 || g.H > 32 || g.H < 1) return false;
    if (((uintptr_t)x | (uintptr_t)z) & 15) return false;
    return true;
}

int launch_foldpack_jobs(const FoldJobs &jobs, int njobs, int ndir, hipStream_t s)
{
    const Geom &g = jobs.job[0].g; // (all layers of a block share C, KH, KW)
    const int ct = mfma_padded_channels(g.C);
    const dim3 grid(g.KH * g.KW * (ct / 16), ndir, njobs);
    if (g.C > 64 || g.C < 1) IFL_FAIL(IFL_EUNSUPPORTED, "launch_foldpack_mfma: C=%d", g.C);
    if (ct == 64)
        hipLaunchKernelGGL(k_foldpack<64>, grid, dim3(256), 0, s, jobs);
    else
        hipLaunchKernelGGL(k_foldpack<32>, grid, dim3(256), 0, s, jobs);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

int launch_foldpack_mfma(const float *w, void *out0, float *wf0, void *out1, float *wf1, const Geom &g, int transposed,
                         int ndir, unsigned *zero0, unsigned *zero1, hipStream_t s)
{
    FoldJobs jobs;
    jobs.job[0] = FoldJob{w, out0, wf0, out1, wf1, g, transposed, zero0, zero1};
    return launch_foldpack_jobs(jobs, 1, ndir, s);
}

// compute units of the current device (a cache of a device property: written once per device, with its final value)
static int device_cus()
{
    static std::atomic<int> cus[IFL_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= IFL_MAX_DEVICES) return 0;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (!v) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

template <int C, int KH, int KW, int NTILE, bool PAD>
static int launch_one(const float *x, float *z, const void *apack, const Geom &g, int rh, int rw, int *flags,
                      const float *wf32, unsigned *amax, hipStream_t s)
{
    using Cfg = ScanCfg<C, KH, KW, NTILE>;
    static_assert(Cfg::LDSB <= 160 * 1024, "ring + x staging must fit the CU's LDS");
    static LdsOptIn opt_in;
    if (int rc = lds_opt_in(opt_in, (const void *)k_scan_mfma<C, KH, KW, NTILE, PAD>, Cfg::LDSB)) return rc;
    if (scan_general_lds_bytes(g) > (size_t)Cfg::LDSB)
        IFL_FAIL(IFL_EUNSUPPORTED, "launch_scan_mfma: fp32 fallback does not fit the kernel's LDS");
#ifdef IFL_STAMPS
    if (const char *e = getenv("IFL_STAMPS")) {
        unsigned long long *ptr = (unsigned long long *)strtoull(e, nullptr, 0);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &ptr, sizeof(ptr));
    }
#endif
    // (two workgroups per image: the duo form, scan_duo.hip -- launch_scan_mfma routes the shapes it takes there)
    hipLaunchKernelGGL((k_scan_mfma<C, KH, KW, NTILE, PAD>), dim3(g.B), dim3(Cfg::THREADS), Cfg::LDSB, s, x, z,
                       (const half8 *)apack, g.H, g.W, rh, rw, flags, wf32, g, amax);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

bool scan_duo_route(const Geom &g, const void *state, bool whole_image)
{
    if (!scan_duo_supported(g)) return false;
    if (g.H <= 16) return true;
    return state && g.B <= scan_duo_max_images() && (whole_image || 2 * g.B <= device_cus());
}

int launch_scan_mfma(const ScanIO &io, const void *apack, const Geom &g, int rh, int rw, int *flags, const float *wf32,
                     unsigned *amax, void *state, bool whole_image, hipStream_t s)
{
    const int nt = g.H <= 16 ? 1 : 2;
    const int ct = mfma_padded_channels(g.C);
    // Layers of exactly 32 or 64 channels on 32-pixel rows: the duo form (scan_duo.hip), one workgroup per 16-row tile --
    // an image of more than 16 rows is two workgroups, which needs the caller's scan-state block and a compute unit per
    // workgroup.  Everything else: one workgroup per image, below.
    if (scan_duo_route(g, state, whole_image)) {
        if (nt == 1) return launch_scan_duo(io, apack, g, rh, rw, flags, wf32, amax, nullptr, false, s);
        // (whole_image: the same two sweeps by one workgroup per image, bit-identical to the two-workgroup form)
        if ((uintptr_t)state & 255) IFL_FAIL(IFL_EINVAL, "scan_state must be 256-byte aligned");
        return launch_scan_duo(io, apack, g, rh, rw, flags, wf32, amax, state, whole_image, s);
    }
    if (io.x16 || io.z16 || !io.x32 || !io.z32) IFL_FAIL(IFL_EUNSUPPORTED, "launch_scan_mfma: bf16 storage is the duo scan's");
    const float *x = io.x32;
    float *z = io.z32;
#define IFL_CASE(CC, KK, NN) \
    if (ct == CC && g.KH == KK && g.KW == KK && nt == NN)                                                                   \
        return g.C == CC ? launch_one<CC, KK, KK, NN, false>(x, z, apack, g, rh, rw, flags, wf32, amax, s)                  \
                         : launch_one<CC, KK, KK, NN, true>(x, z, apack, g, rh, rw, flags, wf32, amax, s);
    IFL_CASE(64, 3, 1)
    IFL_CASE(64, 3, 2)
    IFL_CASE(32, 3, 1)
    IFL_CASE(32, 3, 2)
    IFL_CASE(64, 2, 1)
    IFL_CASE(64, 2, 2)
    IFL_CASE(32, 2, 1)
    IFL_CASE(32, 2, 2)
#undef IFL_CASE
    IFL_FAIL(IFL_EUNSUPPORTED, "launch_scan_mfma: no instantiation for C=%d K=%dx%d H=%d", g.C, g.KH, g.KW, g.H);
}

} // namespace ifl
