// MFMA wavefront back-substitution scan, "duo" form: two waves per SIMD with different jobs, the tile resident in LDS
// (gfx950, wave64).
//
// Same mathematics, same lattice mapping, same r ring and the same split-fp16 arithmetic as scan_mfma.hip (right fold
// r_p = x_p - sum_t (W_t L^-1) r_{p-t}, z_p = L^-1 r_p; push form; MFMA column n = image row, walking along w = d - h) --
// read the header of that file first.  A workgroup owns ONE row tile (16 image rows; an image of 17..32 rows is two
// workgroups, see "hand-off") and runs 2 x C/16 waves, two per SIMD:
//
//   chain waves  (wave w < C/16): the dependent chain.  Step d: fragment reads of r_{d-1}, the taps one diagonal back
//                ((0,1), (1,0): the only products ON the chain) into diagonal d, the epilogue r_d = seed + acc -> split
//                fp16 -> ring, and -- woven around the epilogue -- the taps two diagonals back ((0,2), (1,1), (2,0)) into
//                diagonal d+1.  30 MFMAs at C = 64, 3x3; no vector-memory instruction.
//   helper waves (wave C/16 + w, the same SIMD): everything with slack.  The taps three and four diagonals back ((1,2),
//                (2,1), (2,2)) and z = L^-1 r run on the fragments the helper read in the PREVIOUS step (r_{d-2}): 24 MFMAs
//                whose operands are in registers when the barrier opens, so they fill the matrix pipe while the chain
//                wave waits for its fragments.  What they sum up for a pixel reaches the chain as that pixel's SEED (below).
//                Plus all data movement and the hand-off mailbox.
//
// The tile lives in LDS.  x of the tile's 16 rows is fetched by LDS-DMA as whole 128-byte (channel, row) lines -- a row PF
// steps before its first pixel -- into [row][8-channel chunk][1 KiB], the sixteen-byte pieces of a chunk permuted so that
// the per-diagonal accesses (lane = (row, 4 channels), one pixel each) are bank-conflict free.  Every (channel, pixel)
// word goes through three states IN PLACE: x -> seed = x + (helper's partial sums of that pixel), written by the helper
// one step before the chain needs it -> z, written by the helper two steps after the chain produced r.  When a row's last z
// is in, the row leaves as whole lines (ds_read_b128 + global_store_dwordx4 of the same 1-KiB pieces).  No staging
// buffers, no per-quad duties, no row registers: a helper's step is 12 fragment reads, 24 MFMAs, two read2 + four
// write2 of tile words, and now and then two DMAs or two stores (round 2 kept rows in hand-numbered accumulator
// registers and moved quads through two staging areas; its helpers needed ~1800 cycles a step and the chain waves waited
// 640 of them at the barrier).
//
// One s_barrier per step joins all waves: after barrier d the ring holds r_{d-1} and the tile holds the seeds of
// diagonal d.
//
// Hand-off (images of 17..32 rows, two workgroups i and i+8 of the grid): the upper part's helper 0 publishes rows 14
// and 15 of every finished diagonal as one 1 KiB mailbox line of 8-byte {value, tag} granules (write-through); the
// lower part's helper 0 prefetches the line by LDS-DMA, checks the tags and writes the rows into row block 0 of its ring
// (the block that is the zero padding of a whole image) one step before its waves read them.  Tags are launch
// generations kept in the caller's scan state (an argument of the entry points): no cleaning, valid under graph replay.  The upper
// part waits for nobody; every wait of the lower part is bounded; an image whose hand-off failed or whose r left the
// fp16 range is redone by the lower part's workgroup alone (both tiles in turn through the same mailbox, x scaled by
// 2^-12), and beyond that by the exact fp32 body.  IFL_FLAG_WHOLE_IMAGE runs the same two sweeps in ONE workgroup per
// image (nparts = 1 with H > 16): the same instruction sequence on the same operands, bit-identical results.
#include <stdlib.h>
#include <type_traits>

#include "bf16_util.h"
#include "ifl_common.h"
#include "mfma_util.h"
#include "scan_general_body.h"

namespace ifl {

template <int C, int KH, int KW> struct DuoCfg {
    static constexpr int NW = C / 16;      // chain waves = 16-channel output groups; as many helper waves
    static constexpr int NQ = C / 32;      // 32-deep k-steps per tap
    static constexpr int NT = KH * KW;     // taps incl. the diagonal one
    static constexpr int NS = NT;          // A slots of the packed weights: NT-1 folded taps + L^-1
    static_assert(KH + KW - 2 <= 4 && KH <= 3 && KW <= 3, "push scan: taps reach at most 4 diagonals ahead, 2 rows up");
    static constexpr int NPL = NQ * 8;     // planes per row block: (k-step, hi/lo, k-group)
    static constexpr int RBB = NPL * 256;  // one row block (16 rows x NPL planes x 16 B)
    static constexpr int SLOTB = 2 * RBB;  // one diagonal: row block 0 (rows above the tile: zero or the hand-off) + the tile
    static constexpr int RINGB = 2 * SLOTB;
    static constexpr int NCH = C / 8;      // 8-channel chunks of a row: 1 KiB each (8 lines of 32 pixels)
    static constexpr int TROWB = NCH * 1024;
    static constexpr int TILEB = 16 * TROWB;
    static constexpr int OFF_T = RINGB;
    static constexpr int OFF_DUMP = OFF_T + TILEB; // where lanes outside the image write (branch-free epilogues)
    static constexpr int DUMPB = 3072;             // chain: [0, 512) + [1024, 1536); helpers: [2048, 2048 + 256 + 384 + 4)
    static constexpr int NHL = 8;                  // landing slots of mailbox lines (lower part)
    static constexpr int OFF_HALO = OFF_DUMP + DUMPB;
    static constexpr int OFF_DMY = OFF_HALO + NHL * 1024; // landing of the mailbox prefetches no line is due for
    static constexpr int LDSB = OFF_DMY + 1024;
    static constexpr int THREADS = 128 * NW;
    static constexpr int CPH = NCH / NW;   // chunks of a row that one helper moves (its own 16 channels)
    static_assert(CPH == 2, "a helper's 16 channels are two chunks");
    static constexpr int PF = 8;           // a row is requested PF steps before its first pixel
    static constexpr int LEAD = 1;         // steps before the first pixel of a tile that takes no hand-off (its first row is waited for there)
    static constexpr int ZLAG = 2;         // z of diagonal d - ZLAG goes into the tile in step d (formed at the end of step d - 1)
    static constexpr int SLAG = ZLAG + 32; // row r leaves in step r + SLAG (one step after its last z went into the tile)
#ifndef IFL_PFH
#define IFL_PFH 2
#endif
    static constexpr int PFH = IFL_PFH;    // mailbox lines are requested PFH steps before they are delivered
#ifndef IFL_GATE
#define IFL_GATE 2
#endif
    static constexpr int GATE = IFL_GATE;  // the lower part asks for its first line once the upper part's diagonal 14 + GATE is visible
    static_assert(PFH < NHL && PF >= PFH + 2, "the sweep's lead-in covers both prefetches");
    // which wave multiplies tap (dh, dw): by the number of diagonals it reaches back
    // which wave multiplies tap (dh, dw).  IFL_SPLIT_TAPS (an experiment that is kept compilable): the taps three and four
    // diagonals back go to the helper, which hands their sums over as a seed written over x in the tile.  Measured: the
    // helper's products then contend with the chain's for the matrix pipe at the wrong moments; the step got longer.
#ifndef IFL_SPLIT_TAPS
#define IFL_SPLIT_TAPS 0
#endif
    static constexpr bool chain_tap(int dh, int dw) { return dh + dw >= 1 && (!IFL_SPLIT_TAPS || dh + dw <= 2); }
    static constexpr bool helper_tap(int dh, int dw) { return IFL_SPLIT_TAPS && dh + dw >= 3; }
    static constexpr bool HAS_HT = IFL_SPLIT_TAPS && KH + KW - 2 >= 3; // the helper contributes partial sums
};

#define IFL_STR2(x) #x
#define IFL_STR(x) IFL_STR2(x)
#ifndef IFL_LOWER_SLEEP
#define IFL_LOWER_SLEEP 127 // x 64 cycles: what a lower part waits before it requests its rows
#endif
#ifndef IFL_A_VS
#define IFL_A_VS 4
#endif
#ifndef IFL_WEAVE
#define IFL_WEAVE 4 // instructions of the chain's epilogue per trailing MFMA
#endif
#ifndef IFL_HELPER_SLEEP
#define IFL_HELPER_SLEEP 0 // x 64 cycles: what a helper waits behind the barrier before it touches the LDS (the chain waves' requests first)
#endif
#ifndef IFL_PRIO_CHAIN
#define IFL_PRIO_CHAIN 2
#endif
#ifndef IFL_PRIO_HELPER
#define IFL_PRIO_HELPER 0
#endif
// Development aid (tools/exp_scan.sh, tools/time_scan.py): what-if builds that drop one kind of work (results are then
// garbage) to see what a step is waiting for.  1: no seed update, 2: no z write, 4: no helper products, 8: no seed read
// (chain), 16: no helper fragment reads, 32: no trailing products (chain), 64: no row loads, 128: no ring write, 256: no row stores.
// Any bit forces the verdict good.  Never defined in the product.
#ifndef IFL_EXP
#define IFL_EXP 0
#endif
// Development aid: every image takes the redo path (scaled sweeps), so that ordinary inputs test it.  Never defined in the product.
#ifndef IFL_FORCE_REDO
#define IFL_FORCE_REDO 0
#endif
// Development aid (tools/duo_stamps.py; build with HIPCC_EXTRA=-DIFL_STAMPS): timeline of image 0's two workgroups.
#ifdef IFL_STAMPS
__device__ unsigned long long *g_stamps = nullptr;
#endif

// mailbox geometry (bytes): [image][DUO_LINES][1 KiB]; line u holds rows 14, 15 of the upper part's diagonal u, the last
// line is the verdict.  Independent of the channel count (a narrower layer leaves part of a line unused).
static constexpr int DUO_LINES = 80;
static constexpr int DUO_LINEB = 1024;
static constexpr int DUO_MAX_IMAGES = 128;

// ---- LDS / memory statements the compiler does not track (counted waits by hand; mfma_util.h has the fragment reads) ----
// four words 128 bytes apart (a lane's four channels of one pixel in the tile): two read2 / write2
__device__ __forceinline__ void tile_read4(floatx2 &a, floatx2 &b, unsigned addr)
{
    asm volatile("ds_read2_b32 %0, %2 offset0:0 offset1:32\n\tds_read2_b32 %1, %2 offset0:64 offset1:96" : "=&v"(a), "=&v"(b) : "v"(addr));
}
__device__ __forceinline__ void tile_write4(unsigned addr, float v0, float v1, float v2, float v3)
{
    asm volatile("ds_write2_b32 %0, %1, %2 offset0:0 offset1:32\n\tds_write2_b32 %0, %3, %4 offset0:64 offset1:96" ::"v"(addr), "v"(v0),
                 "v"(v1), "v"(v2), "v"(v3)
                 : "memory");
}
// a fragment set (hi and lo planes of all k-steps) into the accumulator half of the register file: the helper waves carry two
// sets across steps, and an MFMA reads its B operand from there just as well
template <int NQ, int OFF> __device__ __forceinline__ void lds_read_set_a(half8 (&h)[NQ], half8 (&l)[NQ], unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(h[0]) : "v"(addr), "n"(OFF));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(l[0]) : "v"(addr), "n"(OFF + 1024));
    if constexpr (NQ == 2) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(h[1]) : "v"(addr), "n"(OFF + 2048));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(l[1]) : "v"(addr), "n"(OFF + 3072));
    }
}
// one 1-KiB piece: global -> LDS at lds_dst + 16 lane (LDS-DMA; s_nop: m0 and a scalar written by a vector instruction -- a
// spill reload -- are read late by a vector-memory instruction, and nobody inserts wait states in front of an asm statement)
__device__ __forceinline__ void dma_piece(unsigned lds_dst, unsigned voff, const char *src)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(src) : "memory", "m0");
}
// Cache policy of the z rows: 0 default, 1 nt, 2 sc1, 3 sc0 sc1.  With the default policy the 33.5 MB of z that a launch
// writes in its last third sit dirty in the L2s when it ends and the launch is not over before they are written back
// (+4 us measured); written through (sc1) they leave while the sweep is still running.  The weight-gradient kernel that
// reads z next takes the same time either way.
#ifndef IFL_ST_POL
#define IFL_ST_POL 2
#endif
#define IFL_POL_0
#define IFL_POL_1 nt
#define IFL_POL_2 sc1
#define IFL_POL_3 sc0 sc1
#define IFL_CAT2(a, b) a##b
#define IFL_CAT(a, b) IFL_CAT2(a, b)
__device__ __forceinline__ void store_piece(unsigned voff, const floatx4 &v, char *dst)
{
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 " IFL_STR(IFL_CAT(IFL_POL_, IFL_ST_POL)) "\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(dst) : "memory");
}

// BF: x is stored as bf16 (the *_bf16 entry points): its rows arrive as half-width lines and are widened inside the tile; z
// leaves as fp32 (zout), as bf16 rounded to nearest even (zout16), or both, whichever is not NULL -- bit for bit what the
// f32 launch on the widened input returns, rounded.
__device__ __forceinline__ void store_piece16(unsigned voff, const uintx2 &v, char *dst)
{
    asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2 " IFL_STR(IFL_CAT(IFL_POL_, IFL_ST_POL)) ::"v"(voff), "v"(v), "s"(dst) : "memory");
}

template <int C, int KH, int KW, bool PAD, bool BF>
__global__ __launch_bounds__(128 * (C / 16)) void k_scan_duo(const void *__restrict__ xin_, float *__restrict__ zout,
                                                             bf16_t *__restrict__ zout16,
                                                             const half8 *__restrict__ apack, const int H, const int W,
                                                             const int rh, const int rw, int *__restrict__ flags,
                                                             const float *__restrict__ wf32, const Geom geom,
                                                             unsigned *__restrict__ amax, const SplitState sp, const int nparts)
{
    using Cfg = DuoCfg<C, KH, KW>;
    constexpr int NW = Cfg::NW, NQ = Cfg::NQ, NS = Cfg::NS, RBB = Cfg::RBB, SLOTB = Cfg::SLOTB;
    constexpr int PF = Cfg::PF, PFH = Cfg::PFH, LEAD = Cfg::LEAD, ZLAG = Cfg::ZLAG, SLAG = Cfg::SLAG, CPH = Cfg::CPH;
    constexpr int NDMA = BF ? 1 : Cfg::CPH; // vector-memory operations of a wave per requested row
    constexpr int PER = NQ * 2; // LDS reads of one fragment set
    constexpr int GM = 3 * NQ;  // MFMAs of one tap
    static_assert(!PAD, "the duo scan takes layers of exactly 32 or 64 channels (launch_scan_mfma routes the others)");
    using XT = std::conditional_t<BF, bf16_t, float>;
    const XT *__restrict__ xin = (const XT *)xin_;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

#ifdef IFL_PAD
    asm volatile(".rept " IFL_STR(IFL_PAD) "\n\ts_nop 0\n\t.endr"); // (development aid: shifts the code behind it by 4 IFL_PAD bytes)
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_chain = wave < NW;
    const int wv = is_chain ? wave : wave - NW; // 16-channel group of this wave (chain: its outputs; helper: its products and rows)
    const int n = lane & 15, g = lane >> 4;
    const bool split = nparts == 2;
    // solo: one workgroup sweeps both tiles of an image of more than 16 rows in turn (IFL_FLAG_WHOLE_IMAGE)
    const bool solo = !split && H > 16;
    const int b = split ? (int)((blockIdx.x >> 4) * 8 + (blockIdx.x & 7)) : (int)blockIdx.x;
    const int my_part = split ? (int)((blockIdx.x >> 3) & 1) : -1;
    if (b >= geom.B) return;

    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const int c0 = 16 * wv + 4 * g; // C/D layout: lane (n, g) owns channels c0..c0+3 of pixel row n

    // hand-off state (split and solo sweeps use the mailbox)
    const bool twotile = split || solo;
    unsigned gen0 = 0;
    char *mb = nullptr;
    if (twotile) {
        gen0 = __builtin_amdgcn_readfirstlane(sp.gen[b]);
        mb = (char *)sp.mbox + (size_t)b * DUO_LINES * DUO_LINEB;
    }
    unsigned long long *const verdict = (unsigned long long *)(mb + (size_t)(DUO_LINES - 1) * DUO_LINEB);

    float rmax = 0.f; // chain: max |r| this lane put into the ring (beyond the fp16 range the image is redone)
    float zmax = 0.f; // helper: max |z| this lane put into the tile (the weight-gradient kernel's prescale)
    int dead = 0;     // helper 0 of a lower part: the upper part never showed up (bounded spin ran out)

    auto zero_ring = [&]() {
        const floatx4 zz = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid * 16; i < Cfg::RINGB; i += Cfg::THREADS * 16) *(floatx4 *)(lds + i) = zz;
    };
    zero_ring();

    // ---- addresses shared by both kinds of waves --------------------------------------------------------------------------
    unsigned radr[KH]; // LDS address (slot 0) of this lane's B piece for a source dh rows up
#pragma unroll
    for (int dh = 0; dh < KH; ++dh) {
        const int hs = n - dh + 16; // row block 0 of a slot holds the rows above the tile
        radr[dh] = ldsbase + (hs / 16) * RBB + g * 256 + (hs % 16) * 16;
    }
    // The tile: [row][chunk = channel / 8][1 KiB].  Inside a chunk the sixteen-byte piece (cc = channel % 8, q = quad of the
    // stored line) sits at position 8 cc + ((q + 4 (cc >> 2)) & 7).  Lane (n, g) touches its four channels c0 + i of ONE pixel
    // per access (i = 0..3: 128 bytes apart, one read2 / write2 pair): within a 32-lane group g & 1 takes both values and
    // the 16 rows' pixels are 16 consecutive columns, so bank = (column + 16 (g & 1)) mod 32 -- all different.
    const unsigned tbase = ldsbase + Cfg::OFF_T + n * Cfg::TROWB + (2 * wv + (g >> 1)) * 1024 + (g & 1) * 512;
    const int tq4 = 4 * (g & 1);
    const unsigned tidle = tbase + (lane & 31) * 4; // what a lane without a pixel reads: a word of its own bank
    // the same address in two instructions: byte (4 wp + 64 (g & 1)) mod 128 of the lane's 128 bytes, wp = the stored column
    // of logical column t - n (reflected or not): (crel + srel t) mod 128 with a per-lane constant
    const int crel = rw ? 4 * (31 + n) + 64 * (g & 1) : 64 * (g & 1) - 4 * n;
    const int srel = rw ? -4 : 4;
    int Ws = W;
    asm volatile("" : "+s"(Ws));
    [[maybe_unused]] auto taddr = [&](const int w) { // tile address of (row n, channels c0.., logical column w), 0 <= w < 32
        const int wp = rw ? 31 - w : w;
        return tbase + ((((wp >> 2) + tq4) & 7) << 4) + ((wp & 3) << 2);
    };
    // ---- the tile's rows move as whole 1-KiB pieces: the two chunks of a wave's own 16 channels (helper w and chain wave w share
    //      them).  Lane l of a piece holds the sixteen bytes at LDS position l: channel cc = l >> 3 of the chunk, quad
    //      ((l & 7) - 4 (cc >> 2)) & 7 of its line: 8 consecutive lanes cover one 128-byte (channel, row) line, loads and stores
    const int pcc = lane >> 3, pq = ((lane & 7) - 4 * (pcc >> 2)) & 7;
    const unsigned pgo = (unsigned)(pcc * H * W * 4 + pq * 16); // byte offset inside a chunk's 8 channel planes (fp32)
    const char *xg = (const char *)xin + ((size_t)b * C + 16 * wv) * H * W * sizeof(XT);
    const unsigned chunkB = (unsigned)(8 * H * W * 4);
    const unsigned tchunk = ldsbase + Cfg::OFF_T + (2 * wv) * 1024; // (+ row TROWB, + 1024 for the second chunk)
    // bf16 rows: a (channel, row) line is 64 bytes, a wave's 16 channels ONE 1-KiB piece: lane l brings the sixteen bytes
    // (channel l >> 2, pixels 8 (l & 3) ...).  It lands in the second chunk's place and is widened from there (cvt_row).
    const unsigned bgo = (unsigned)((lane >> 2) * H * W * 2 + (lane & 3) * 16);
    // rows [r0, r1) of the tile at image row hoff, this wave's pieces: global -> tile (rows may be reflected)
    auto load_rows = [&](const int hoff, const int r0, const int r1) {
        for (int r = r0; r < r1; ++r) {
            const char *src = xg + (rh ? H - 1 - (hoff + r) : hoff + r) * W * (int)sizeof(XT);
            const unsigned dst = __builtin_amdgcn_readfirstlane(tchunk + r * Cfg::TROWB);
            if constexpr (BF) {
                dma_piece(dst + 1024, bgo, src);
            } else {
                dma_piece(dst, pgo, src);
                dma_piece(dst + 1024, pgo + chunkB, src);
            }
        }
    };
    // a landed bf16 row of this wave -> fp32, in place: every lane reads its sixteen bytes (8 pixels of one channel), the
    // wave waits, every lane writes the two quads where the fp32 layout wants them (all reads are back before the first
    // write goes out: the landing place and the destinations overlap)
    auto cvt_row = [&](const int r) {
        if constexpr (BF) {
            const unsigned base = tchunk + r * Cfg::TROWB;
            uintx4 v;
            asm volatile("ds_read_b128 %0, %1 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(base + lane * 16) : "memory");
            const int c16 = lane >> 2, cc = c16 & 7, q0 = 2 * (lane & 3);
            const unsigned d0 = base + (c16 >> 3) * 1024 + (8 * cc + ((q0 + 4 * (cc >> 2)) & 7)) * 16;
            const unsigned d1 = base + (c16 >> 3) * 1024 + (8 * cc + ((q0 + 1 + 4 * (cc >> 2)) & 7)) * 16;
            const uintx4 a = {v[0] << 16, v[0] & 0xffff0000u, v[1] << 16, v[1] & 0xffff0000u};
            const uintx4 c = {v[2] << 16, v[2] & 0xffff0000u, v[3] << 16, v[3] & 0xffff0000u};
            asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %2, %3" ::"v"(d0), "v"(a), "v"(d1), "v"(c) : "memory");
        }
    };
    // The launched tile's rows are requested at kernel entry, in front of everything else: the latency of the first rows
    // overlaps the weight loads.  The helper takes the rows it will wait for one by one, [0, HSPLIT); the chain wave of the
    // same channels takes the rest and waits for all of them at once, two steps before the first of them is due.
    constexpr int HSPLIT = 8;

#ifdef IFL_STAMPS
    unsigned long long st_rt[4] = {__builtin_amdgcn_s_memrealtime(), 0, 0, 0}, st_mt[2] = {0, 0}, st_bar = 0;
    unsigned long long st_c[5] = {0, 0, 0, 0, 0}, st_cl = __builtin_amdgcn_s_memtime();
#endif

    // The sweeps a workgroup runs, in order.  Launched tile(s) first: a split launch's workgroup has one (its part), a
    // solo launch both in turn, a single-tile image one; then, if the verdict is bad, the redo of the whole image with x
    // scaled by 2^-12 (both tiles in turn), by the workgroup of the lower part.
    struct Sweep {
        int hoff, Hp;
        float xscale, zscale;
        bool publish, consume;
        unsigned tag;
        int dfirst;
        bool first; // the launched pass's first sweep: its rows were requested at kernel entry
    };
    const unsigned tag1 = gen0 + 1, tag2 = gen0 + 2;
    auto sweep_of = [&](const int redo, const int t) {
        Sweep s;
        const bool two = twotile;
        const int part = redo ? t : (split ? my_part : t);
        s.hoff = two && part == 1 ? 16 : 0;
        s.Hp = two ? (part == 1 ? H - 16 : 16) : H;
        s.xscale = redo ? 1.0f / 4096.0f : 1.0f;
        s.zscale = redo ? 4096.0f : 1.0f;
        s.publish = two && part == 0;
        s.consume = two && part == 1;
        s.tag = redo ? tag2 : tag1;
        s.dfirst = s.consume ? -PF : -LEAD;
        s.first = !redo && t == 0;
        return s;
    };
    const int nfirst = solo ? 2 : 1; // sweeps of the launched pass
    const int nredo = twotile ? 2 : 1;

    if (is_chain) {
        // =================================== chain waves ===================================================
        if (!(IFL_EXP & 64)) {
            const Sweep s0 = sweep_of(0, 0);
            if (my_part == 1) __builtin_amdgcn_s_sleep(IFL_LOWER_SLEEP);
            load_rows(s0.hoff, HSPLIT, s0.Hp);
        }
        half8 A[NS - 1][NQ][2]; // this wave's folded taps as A fragments (hi, lo), in registers for the whole kernel
        {
            half8 Aload[NS - 1][NQ][2]; // all loads first, then the pins (a pin behind its load serialises the round trips)
#pragma unroll
            for (int s = 0; s < NS - 1; ++s)
                if (Cfg::chain_tap((s + 1) / KW, (s + 1) % KW))
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int hl = 0; hl < 2; ++hl)
                            Aload[s][q][hl] = apack[((((size_t)wv * NS + s) * NQ + q) * 2 + hl) * 64 + lane];
#pragma unroll
            for (int s = 0; s < NS - 1; ++s)
                if (Cfg::chain_tap((s + 1) / KW, (s + 1) % KW))
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int hl = 0; hl < 2; ++hl) {
                            A[s][q][hl] = Aload[s][q][hl];
                            // (pinned, or hipcc re-loads the weights inside the loop.  An MFMA reads them from the accumulator half
                            // of the register file, which holds 128 registers per lane at two waves per SIMD and the accumulators as
                            // well: the low parts of the first IFL_A_VS taps stay in ordinary registers -- left to itself the
                            // compiler parks them there anyway and copies them into scratch registers in front of every use)
                            if (NW == 4 && hl == 1 && s < IFL_A_VS) asm volatile("" : "+v"(A[s][q][hl]));
                            else asm volatile("" : "+a"(A[s][q][hl]));
                        }
        }
        const int wadr = RBB + (((c0 / 32) * 2) * 4 + (c0 % 32) / 8) * 256 + n * 16 + ((c0 % 8) / 4) * 8;

        auto chain_sweep = [&](const Sweep sw) {
            const int Hp = sw.Hp;
            const float xscale = sw.xscale;
            const bool hval = n < Hp;
            const int ND = Hp + W - 1;       // diagonals 0 .. ND-1
            const int DEND = ND + SLAG - 32; // the helpers' last step (the last row leaves)
            // Rolling accumulators ("push" form): the accumulator of diagonal t is [(t + 1) mod 3]; the step loop is unrolled by
            // three, so that every step names its accumulators at compile time (no register copies when they rotate)
            floatx4 ahi[3], amid[3];
            half8 F2h[NQ], F2l[NQ]; // fragments of the source rows two up, carried to the next step
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ahi[k] = floatx4{0.f, 0.f, 0.f, 0.f};
                amid[k] = floatx4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    F2h[q][j] = (_Float16)0.f;
                    F2l[q][j] = (_Float16)0.f;
                }

#ifdef IFL_STAMPS
#define IFL_CSTAMP(k)                                                 \
    do {                                                              \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
        st_c[k] += t_ - st_cl;                                        \
        st_cl = t_;                                                   \
    } while (0)
#else
#define IFL_CSTAMP(k) \
    do {              \
    } while (0)
#endif
            auto step = [&](auto r_c, const int d) {
                constexpr int R = decltype(r_c)::value; // (d + 1) mod 3
                constexpr int A0 = R, A1 = (R + 1) % 3, A2 = (R + 2) % 3; // accumulators of the diagonals d, d+1, d+2
                constexpr int NDH = KH < 2 ? KH : 2;
                constexpr int NREQ = 2 + NDH * PER;
                constexpr int NRD2 = KH > 2 ? PER : 0;
                const int srcoff = ((d + 1) & 1) * SLOTB; // ring slot of diagonal d-1
                const int dstoff = (d & 1) * SLOTB;       // ring slot of diagonal d
                // r of diagonal d-1 complete in the ring (this wave's ring writes drained: lgkmcnt), x of this step in the tile
#ifdef IFL_STAMPS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
                st_c[4] += tb0 - st_cl; // (section 4: the trailing products and the epilogue, up to the drained ring write)
                st_cl = tb0;
#endif
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef IFL_STAMPS
                {
                    const unsigned long long tb1 = __builtin_amdgcn_s_memtime();
                    st_bar += tb1 - tb0;
                    st_c[0] += tb1 - st_cl; // (section 0: at the barrier)
                    st_cl = tb1;
                    // per-step timeline of image 0: when this wave passed barrier d, and how long it had waited there
                    if (g_stamps && b == 0 && tid == 0 && d - sw.dfirst < 64) {
                        g_stamps[128 + (my_part == 1 ? 128 : 0) + 2 * (d - sw.dfirst)] = tb1;
                        g_stamps[128 + (my_part == 1 ? 128 : 0) + 2 * (d - sw.dfirst) + 1] = tb1 - tb0;
                    }
                }
#endif
                // (the fragments requested during the previous step have landed: see `landed`)
#pragma unroll
                for (int q = 0; q < NQ; ++q) asm volatile("" : "+v"(F2h[q]), "+v"(F2l[q]));
                const int w0 = d - n;
                const bool valid = hval && (unsigned)w0 < (unsigned)Ws;
                // (a lane without a pixel reads a word of its own bank)
                const unsigned xa = valid ? tbase + ((crel + srel * d) & 127) : tidle;
                floatx2 xq[2];
                half8 Fh[2][NQ], Fl[2][NQ];
                // (the fragments first, in the order of their use; x -- wanted by the epilogue only -- last)
                auto request = [&](int j) {
                    int c = 0;
#pragma unroll
                    for (int dh = 0; dh < NDH; ++dh) {
                        const unsigned fa = radr[dh] + srcoff;
                        if (c++ == j) lds_read_b128_o<0>(Fh[dh][0], fa);
                        if (c++ == j) lds_read_b128_o<4 * 256>(Fl[dh][0], fa);
                        if constexpr (NQ == 2) {
                            if (c++ == j) lds_read_b128_o<8 * 256>(Fh[dh][1], fa);
                            if (c++ == j) lds_read_b128_o<12 * 256>(Fl[dh][1], fa);
                        }
                    }
                    if (c++ == j) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:32" : "=v"(xq[0]) : "v"(xa));
                    if (c++ == j) asm volatile("ds_read2_b32 %0, %1 offset0:64 offset1:96" : "=v"(xq[1]) : "v"(xa));
                };
                __builtin_amdgcn_sched_barrier(0);
                // The destination of an asynchronous LDS read must stay allocated until the wait that covers it: as an operand of
                // this statement behind the wait it does.  (A destination nothing reads later -- the last step's fragments for
                // taps that feed a diagonal beyond the image -- is otherwise handed out again at once, and the data that lands
                // later goes on top of whatever lives there by then: tests/test_build_checks.py walks the ISA for exactly that.)
                auto landed = [&](int dh) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) asm volatile("" : "+v"(Fh[dh][q]), "+v"(Fl[dh][q]));
                };
                // A(t) x {hi, lo} fragments -> accumulator tgt (order: per k-step hi.hi and hi.lo, then the lo.hi products)
                auto mf_one = [&](int t, const half8 *fh, const half8 *fl, int tgt, int k, bool init = false) {
                    int c = 0;
                    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        if (c++ == k)
                            ahi[tgt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][0], fh[q], (init && q == 0) ? zero : ahi[tgt], 0, 0, 0);
                        if (c++ == k)
                            amid[tgt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][0], fl[q], (init && q == 0) ? zero : amid[tgt], 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        if (c++ == k) amid[tgt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][1], fh[q], amid[tgt], 0, 0, 0);
                };
                auto mf = [&](int t, const half8 *fh, const half8 *fl, int tgt, bool init = false) {
#pragma unroll
                    for (int k = 0; k < GM; ++k) mf_one(t, fh, fl, tgt, k, init);
                };
                auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };

                // ---- leading: taps (2, dw) of r_{d-2} (fragments kept from the previous step) -> diagonals d + dw; they
                //      fill the matrix pipe while this step's LDS requests (one behind each MFMA) are in flight
                if constexpr (KH > 2) {
                    constexpr int NLEADM = (KW > 1 ? 2 : 1) * GM;
                    constexpr int RPM = (NREQ + NLEADM - 1) / NLEADM;
#pragma unroll
                    for (int k = 0; k < NLEADM; ++k) {
                        if (k < GM) mf_one(2 * KW + 0, F2h, F2l, A0, k);
                        else mf_one(2 * KW + 1, F2h, F2l, A1, k - GM);
                        // (MFMAs are pure: tie the result to an opaque statement, or they sink below the requests)
                        asm volatile("" : "+a"(ahi[k < GM ? A0 : A1]), "+a"(amid[k < GM ? A0 : A1]));
                        fence();
#pragma unroll
                        for (int j = k * RPM; j < (k + 1) * RPM && j < NREQ; ++j) request(j);
                        fence();
                    }
                    if constexpr (KW > 2) mf(2 * KW + 2, F2h, F2l, A2, true); // first contribution to diagonal d+2
                    fence();
                } else {
#pragma unroll
                    for (int j = 0; j < NREQ; ++j) request(j);
                    fence();
                }
                IFL_CSTAMP(1); // leading products + requests (the stamp waits for the requests: they are due at the next wait anyway)
                // ---- critical: taps (0,1) and (1,0) of r_{d-1} -> diagonal d.  The dh=2 fragments of r_{d-1} (next
                //      step's leading operands; every MFMA that reads the old ones has been issued) ride behind the first
                auto request2 = [&](int j) {
                    int c = 0;
                    const unsigned fa = radr[KH > 2 ? 2 : 0] + srcoff;
                    if (c++ == j) lds_read_b128_o<0>(F2h[0], fa);
                    if (c++ == j) lds_read_b128_o<4 * 256>(F2l[0], fa);
                    if constexpr (NQ == 2) {
                        if (c++ == j) lds_read_b128_o<8 * 256>(F2h[1], fa);
                        if (c++ == j) lds_read_b128_o<12 * 256>(F2l[1], fa);
                    }
                };
                if constexpr (KW > 1) {
                    lgkm_wait_n((NDH - 1) * PER + 2); // the dh=0 fragments have landed
                    landed(0);
#pragma unroll
                    for (int k = 0; k < GM; ++k) {
                        mf_one(1, Fh[0], Fl[0], A0, k);
                        if constexpr (KH > 2) {
                            asm volatile("" : "+a"(ahi[A0]), "+a"(amid[A0]));
                            fence();
                            if (k < PER) request2(k);
                            fence();
                        }
                    }
                    fence();
                }
                if constexpr (KH > 1) {
                    lgkm_wait_n(NRD2 + 2); // ... and the dh=1 fragments (behind them: x, and the dh=2 fragments of the next step)
                    landed(1);
                    mf(KW, Fh[1], Fl[1], A0);
                    fence();
                }
                // x has landed (for the epilogue; the next step's dh=2 fragments may still be on their way)
                lgkm_wait_n(NRD2);
                asm volatile("" : "+v"(xq[0]), "+v"(xq[1]));
                IFL_CSTAMP(2); // critical products
                // ---- trailing: the remaining taps of r_{d-1} (targets d+1, d+2) with the chain's epilogue woven in
                const floatx4 head_hi = ahi[A0], head_mid = amid[A0];
#pragma unroll
                for (int dh = 0; dh < NDH; ++dh)
#pragma unroll
                    for (int dw = 0; dw < KW; ++dw)
                        if (dh + dw >= 2) {
                            // (without a dh=2 row the farthest tap is this one: it opens its diagonal)
                            mf(dh * KW + dw, Fh[dh], Fl[dh], dh + dw == 2 ? A1 : A2, KH < 3 && dh + dw == KH + KW - 2);
                        }
                // epilogue: r_d = x + acc -> split fp16 -> ring (lanes outside the image write to the dump: no branch)
                {
                    float rv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) rv[r] = xq[r >> 1][r & 1] * xscale + head_hi[r] + head_mid[r] * LO_INV;
                    half4 hi, lo;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const _Float16 h16 = (_Float16)rv[r];
                        hi[r] = h16;
                        lo[r] = (_Float16)((rv[r] - (float)h16) * LO_SCALE);
                    }
                    unsigned char *rp = (valid && !(IFL_EXP & 128)) ? lds + dstoff + wadr : lds + Cfg::OFF_DUMP + lane * 8;
                    *(half4 *)rp = hi;
                    *(half4 *)(rp + 4 * 256) = lo;
                    const float m = fmaxf(fmaxf(fabsf(rv[0]), fabsf(rv[1])), fmaxf(fabsf(rv[2]), fabsf(rv[3])));
                    rmax = fmaxf(rmax, valid ? m : 0.f);
                }
                // scheduling pattern for the region since the last fence: 1 MFMA, then (1 MFMA, IFL_WEAVE others) until the others
                // run out -- the ring write goes out several MFMAs before the step's last one, whose time covers its latency
                constexpr int NTR = GM * (KH * KW - 1 - (KW > 1 ? 1 : 0) - (KH > 1 ? 1 : 0) - (KH > 2 ? KW : 0));
                static_assert(NTR >= 0, "trailing taps");
                if constexpr (NTR > 4) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 3);
#pragma unroll
                    for (int k = 0; k < NTR - 1; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 3);
                        __builtin_amdgcn_sched_group_barrier(0x296, IFL_WEAVE, 3); // VALU | SALU | VMEM | DS
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };

            // steps dfirst .. DEND, one barrier each (the helpers mirror them); the chain works in steps -1 .. ND-1
            // (step -1: r_{-2} of a lower tile -- rows 14, 15 of the part above -- becomes next step's fragments two rows up)
            int d = sw.dfirst;
            for (; d < -1; ++d) asm volatile("s_barrier" ::: "memory");
            for (; d + 2 <= ND - 1; d += 3) { // ((d + 1) mod 3 = 0 here)
                // (the rows this wave requested at kernel entry -- HSPLIT and up -- have landed: the only vector-memory operations
                // it has in flight; two steps and a barrier before anybody touches the first of them)
                if (d == HSPLIT - 3) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if constexpr (BF)
                        if (sw.first) // (the rows this wave requested: the launched pass's first sweep)
                            for (int r = HSPLIT; r < Hp; ++r) cvt_row(r);
                }
                step(std::integral_constant<int, 0>{}, d);
                step(std::integral_constant<int, 1>{}, d + 1);
                step(std::integral_constant<int, 2>{}, d + 2);
            }
            if (d <= ND - 1) step(std::integral_constant<int, 0>{}, d++);
            if (d <= ND - 1) step(std::integral_constant<int, 1>{}, d++);
            // (the last step's fragment requests for a diagonal that does not exist are covered by this wait)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < NQ; ++q) asm volatile("" : "+v"(F2h[q]), "+v"(F2l[q]));
            for (; d <= DEND; ++d) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };

        // ---- the chain waves' view of the kernel's control flow (the helpers mirror it barrier for barrier) ----
        __syncthreads();
        asm volatile("s_setprio " IFL_STR(IFL_PRIO_CHAIN));
#ifdef IFL_STAMPS
        st_rt[1] = __builtin_amdgcn_s_memrealtime();
        st_mt[0] = __builtin_amdgcn_s_memtime();
#endif
        // One instance of the sweep, in a loop over the workgroup's sweeps: k < nfirst the launched pass, then -- behind a
        // bad verdict -- the redo (x scaled by 2^-12: both tiles in turn, or the single tile), then, beyond that, exact fp32
        for (int k = 0, redo = 0;; ++k) {
            if (k) {
                zero_ring();
                __syncthreads();
            }
            chain_sweep(sweep_of(redo, redo ? k - nfirst : k));
            if (redo || k + 1 < nfirst) __syncthreads();
            if (redo) {
                if (k + 1 == nfirst + nredo) break;
                continue;
            }
            if (k + 1 < nfirst) continue;
#ifdef IFL_STAMPS
            st_mt[1] = __builtin_amdgcn_s_memtime();
            st_rt[2] = __builtin_amdgcn_s_memrealtime();
            if (g_stamps && b == 0 && tid == 0) {
                unsigned long long *o = g_stamps + (my_part == 1 ? 32 : 0);
                const Sweep s0 = sweep_of(0, 0);
                o[0] = st_rt[0], o[1] = st_rt[1], o[2] = st_rt[2], o[3] = st_mt[1] - st_mt[0];
                o[4] = (unsigned long long)(s0.Hp + W - 1 + SLAG - 32 - s0.dfirst + 1);
                o[5] = st_bar;
                for (int j = 0; j < 5; ++j) o[24 + j] = st_c[j];
            }
#endif
            int bad = __syncthreads_or(rmax < 6.0e4f ? 0 : 1);
            if (my_part == 0) return; // (the verdict is the helpers' business)
            if (my_part == 1) bad = __syncthreads_or(0); // helper 0 adds the upper part's verdict
            if (IFL_FORCE_REDO) bad = 1;
            if (IFL_EXP) bad = 0;
            if (!bad) return;
            rmax = 0.f;
            redo = 1;
        }
        const int bad2 = __syncthreads_or(rmax < 6.0e4f ? 0 : 1);
        if (bad2) {
            scan_general_body<Cfg::THREADS, XT>(xin, wf32, zout, geom, rh, rw, 1, (float *)lds, b, tid, zout16);
            __syncthreads();
        }
        return;
    }

    // ======================================= helper waves ==================================================
    {
        half8 Z[NQ][2]; // L^-1 as A fragments (slot NS-1 of the packed weights)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) Z[q][hl] = apack[((((size_t)wv * NS + (NS - 1)) * NQ + q) * 2 + hl) * 64 + lane];
        // (used here, so that the compiler's wait for these loads sits here and not -- as vmcnt(0) -- inside the step loop,
        // where it would drain the DMAs and stores in flight)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) asm volatile("" : "+v"(Z[q][hl]));
        char *zg = (char *)zout + ((size_t)b * C + 16 * wv) * H * W * sizeof(float);
        char *zg16 = (char *)zout16 + ((size_t)b * C + 16 * wv) * H * W * sizeof(bf16_t);
        const unsigned pgo16 = pgo / 2, chunkB16 = chunkB / 2; // (a lane's quad as four bf16: eight bytes of a 64-byte line)
        const unsigned dmy = __builtin_amdgcn_readfirstlane(ldsbase + Cfg::OFF_DMY);
        const unsigned hdump = ldsbase + Cfg::OFF_DUMP + 2048 + lane * 4;
        // mailbox role of helper 0: lane l carries the 8-byte piece (row 14 + (l & 1), plane (l >> 1) & 15, half l >> 5)
        const int mrow = 14 + (lane & 1), mpl = (lane >> 1) & 15, mhalf = lane >> 5;
        const bool mlane = mpl < Cfg::NPL;
        const unsigned madr = ldsbase + (mlane ? mpl : 0) * 256 + mrow * 16 + mhalf * 8; // (+ slot, + RBB for the tile's own rows)
        const bool w_mbox = wv == 0;

#ifdef IFL_STAMPS
        unsigned long long st_slow = 0, st_spins = 0, st_gate = 0, st_h[7] = {0, 0, 0, 0, 0, 0, 0}, st_hl = __builtin_amdgcn_s_memtime();
#define IFL_HSTAMP(k)                                                  \
    do {                                                               \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
        st_h[k] += t_ - st_hl;                                         \
        st_hl = t_;                                                    \
    } while (0)
#else
#define IFL_HSTAMP(k) \
    do {              \
    } while (0)
#endif
        // "all but the n youngest vector-memory operations are complete" for a run-time n (an immediate in the instruction):
        // exact up to 3, above that rounded DOWN to a multiple of four (a few operations more are waited for: rows further
        // down the tile, requested at the same time) -- a short dispatch.  The counter holds 63.
        auto wait_vm = [&](int n) {
            if (n < 4) {
                switch (n) {
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                }
                return;
            }
#define IFL_V(N) \
    case N / 4: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
            switch (n >> 2) {
                IFL_V(4) IFL_V(8) IFL_V(12) IFL_V(16) IFL_V(20) IFL_V(24) IFL_V(28) IFL_V(32) IFL_V(36) IFL_V(40) IFL_V(44) IFL_V(48)
            default: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
            }
#undef IFL_V
        };
        auto reduce_amax = [&]() {
            if (amax) {
                for (int o = 32; o > 0; o >>= 1) zmax = fmaxf(zmax, __shfl_down(zmax, o, 64));
                if (lane == 0) atomicMax(amax, __float_as_uint(zmax)); // one atomic per wave; max is order-independent
            }
        };
        // the next launch uses other tags (both parts have read this one long ago).  Close to the wrap the image's lines
        // are cleaned, so that a tag of 2^31 launches ago cannot pass for a fresh one.
        auto advance_generation = [&]() {
            if (!twotile) return;
            if (gen0 >= 0xFFFFFFF0u) {
                const floatx4 zz = {0.f, 0.f, 0.f, 0.f};
                for (int i = (tid - NW * 64) * 16; i < DUO_LINES * DUO_LINEB; i += NW * 64 * 16) *(floatx4 *)(mb + i) = zz;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            }
            if (tid == NW * 64) sp.gen[b] = gen0 >= 0xFFFFFFF0u ? 0u : gen0 + 2;
        };

        if (!(IFL_EXP & 64)) {
            const Sweep s0 = sweep_of(0, 0);
            // (a lower part's first pixel is sixteen steps away: it stays out of the way of the upper parts' requests, which
            // every compute unit of the chip issues at this very moment)
            if (my_part == 1) __builtin_amdgcn_s_sleep(IFL_LOWER_SLEEP);
            load_rows(s0.hoff, 0, s0.Hp < HSPLIT ? s0.Hp : HSPLIT);
        }

        auto helper_sweep = [&](const Sweep sw, const bool last_of_pass, const bool preloaded) {
            const int Hp = sw.Hp, hoff = sw.hoff;
            const float zscale = sw.zscale;
            const unsigned tag = sw.tag;
            const int dfirst = sw.dfirst;
            const int ND = Hp + W - 1;
            const int DEND = ND + SLAG - 32;
            const int u_last = W + 14;  // last upper diagonal with a pixel in row 15
            const int dl_last = W - 2;  // ... as a diagonal of the lower tile
            const bool mbox = w_mbox && (sw.publish || sw.consume);
            const bool mb_in = sw.consume && mbox, mb_out = sw.publish && mbox;
            const int M = mbox ? 1 : 0; // mailbox operations per step (exactly one: a spare line takes the steps without one)
            const bool hval = n < Hp;
            // byte offset of tile row r inside a channel plane (rows may be reflected)
            auto row_off = [&](int r) { return (rh ? H - 1 - (hoff + r) : hoff + r) * W * 4; };

            auto poll_line = [&](const int u, uintx4 &q) {
                // the line prefetched PFH steps ago was not complete: poll it (bounded) with agent-scope loads
                const unsigned long long *hp = (const unsigned long long *)(mb + (size_t)u * DUO_LINEB + lane * 16);
#ifdef IFL_STAMPS
                st_slow += 1;
#endif
                for (int spins = 0;; ++spins) {
#ifdef IFL_STAMPS
                    st_spins += 1;
#endif
                    const unsigned long long a0 = __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long a1 = __hip_atomic_load(hp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    q = uintx4{(unsigned)a0, (unsigned)(a0 >> 32), (unsigned)a1, (unsigned)(a1 >> 32)};
                    if (__all(q[1] == tag && q[3] == tag)) return;
                    if (spins > 20000) { // ~tens of ms: the image is void and redone whole
                        dead = 1;
                        return;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            };
            // All rows of the tile are requested at once, in row order, before the sweep's first step (the launched tile's: at
            // kernel entry, see below): the requests cost the helper its issue slots while nothing else is going on, and a row is
            // waited for one step before its first pixel.
            if (!preloaded && !(IFL_EXP & 64)) load_rows(hoff, 0, Hp);
            const int HR = preloaded ? (Hp < HSPLIT ? Hp : HSPLIT) : Hp; // the rows this wave requested

            // Every per-step condition is a range of d: one unsigned compare each, "(unsigned)(d - lo) < n" with n = 0
            // when the sweep does not have that duty (the conditions are wave-uniform)
            const unsigned n_hin = mb_in ? (unsigned)(dl_last + 3) : 0u;   // d in [-2, dl_last]
            const unsigned n_hout = mb_out ? (unsigned)(u_last - 13) : 0u; // d - 1 in [14, u_last]

            floatx4 zh = {0.f, 0.f, 0.f, 0.f}, zm = {0.f, 0.f, 0.f, 0.f}; // z product of the previous step (diagonal d - ZLAG)

            // ---- one step.  LD: the sweep's first steps, in which rows are requested and waited for; ST: its last ones, in
            //      which rows leave (the step in between has neither: compile-time, so that it carries no test for them).
            // The chain wave of this SIMD is the pole of the step and every instruction issued here takes an issue slot from
            // it: a step is the z of the previous step into the tile, four fragment reads, six MFMAs, and little else.
            // Vector-memory operations of a sweep, in this order (the counted waits rely on it): all rows; then per step the mailbox
            // operation (exactly one per step of a mailbox helper: a spare line takes the steps without one) and -- long after
            // the last row has landed -- a row store.
            auto step = [&](auto ld_c, auto st_c, const int d) {
                constexpr bool LD = decltype(ld_c)::value, ST = decltype(st_c)::value;
                IFL_HSTAMP(6); // (the wait at the end of the previous step)
                asm volatile("s_barrier" ::: "memory");
                IFL_HSTAMP(0); // barrier
                const int srcoff = ((d + 1) & 1) * SLOTB; // ring slot of diagonal d-1
                if (IFL_HELPER_SLEEP) __builtin_amdgcn_s_sleep(IFL_HELPER_SLEEP);
                // ---- LDS requests: the fragments of r_{d-1} for this step's z product FIRST -- the wave's longest dependent
                //      path is barrier -> fragments -> six MFMAs -> barrier, and the z of the previous step below (some twenty
                //      vector instructions) runs while they are on their way ...
                half8 Fh[NQ], Fl[NQ];
                if (!(IFL_EXP & 16)) lds_read_set<NQ, 0>(Fh, Fl, radr[0] + srcoff);
                // ---- z of diagonal d - ZLAG (formed at the end of the previous step) -> the tile, over the x the chain consumed
                //      ZLAG steps ago.  A column outside the image repeats an older pixel of its row (the ring keeps it) or is
                //      zero: the maximum over everything formed is the maximum over the image.
                const bool z_due = (unsigned)(d - ZLAG) < (unsigned)ND && !(IFL_EXP & 2);
                if (z_due) {
                    const int t = d - ZLAG;
                    const bool vz = hval && (unsigned)(t - n) < 32u;
                    const unsigned za = tbase + ((crel + srel * t) & 127);
                    float zv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) zv[r] = (zh[r] + zm[r] * LO_INV) * zscale;
                    tile_write4(vz ? za : hdump, zv[0], zv[1], zv[2], zv[3]);
                    zmax = fmaxf(fmaxf(zmax, fmaxf(fabsf(zv[0]), fabsf(zv[1]))), fmaxf(fabsf(zv[2]), fabsf(zv[3])));
                }
                // ... the mailbox line that landed (lower part): it was the vector-memory operation of step d - PFH; younger: the
                // lines of the steps in between; rows 14, 15 of the diagonal the chain waves finished in the previous step (upper
                // part) ...
                bool h_in = false, h_out = false;
                floatx4_ hq;
                uintx2 pv;
                if (mbox) {
                    h_in = (unsigned)(d + 2) < n_hin && !dead;
                    h_out = (unsigned)(d - 15) < n_hout;
                    if (h_in) {
                        wait_vm(PFH - 1);
                        lds_read_f32x4(hq, ldsbase + Cfg::OFF_HALO + (d & (Cfg::NHL - 1)) * 1024 + lane * 16);
                    }
                    if (h_out) asm volatile("ds_read_b64 %0, %1" : "=v"(pv) : "v"(madr + RBB + srcoff) : "memory");
                }
                // ... the row whose last z went into the tile a step ago (this wave's own channels: its own LDS writes, complete
                // since the end of that step) on its way out
                const int rs = d - SLAG;
                const bool st_due = ST && (unsigned)rs < (unsigned)Hp && !(IFL_EXP & 256);
                floatx4 sv[CPH];
                if (st_due) {
                    const unsigned la = tchunk + rs * Cfg::TROWB + lane * 16;
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(sv[0]), "=&v"(sv[1]) : "v"(la) : "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                IFL_HSTAMP(1); // z, requests
                // ---- vector-memory requests (see the order above)
                if constexpr (LD) {
                    // Row d+1 -- its first pixel is on diagonal d+1, which the chain reads next step -- must have landed: all but
                    // the operations behind its DMAs, which are the rows after it and the mailbox operations of the sweep's steps
                    // so far (this step's comes behind this wait)
                    if ((unsigned)(d + 1) < (unsigned)HR) {
                        wait_vm(NDMA * (HR - d - 2) + M * (d - dfirst));
                        cvt_row(d + 1);
                    }
                }
                if (mb_in) {
                    if (d == -2 - PFH) {
                        // gate: the first line is requested once the upper part's diagonal 14 + GATE is visible, so that every
                        // later request (one per step, like the upper part's lines) finds its line; the steps before this one
                        // (this tile's first rows are on their way) did not have to wait for the upper part
                        uintx4 q;
#ifdef IFL_STAMPS
                        const unsigned long long g0 = __builtin_amdgcn_s_memrealtime();
#endif
                        poll_line(14 + Cfg::GATE < u_last ? 14 + Cfg::GATE : u_last, q);
#ifdef IFL_STAMPS
                        st_gate = __builtin_amdgcn_s_memrealtime() - g0;
                        st_slow = 0;
#endif
                    }
                    // the line to be delivered PFH steps from now
                    const int dl = d + PFH;
                    const bool ok = (unsigned)(dl + 2) < n_hin;
                    const char *line = mb + (size_t)(ok ? dl + 16 : 0) * DUO_LINEB;
                    const unsigned dst = __builtin_amdgcn_readfirstlane(ok ? ldsbase + Cfg::OFF_HALO + (dl & (Cfg::NHL - 1)) * 1024 : dmy);
                    asm volatile("s_mov_b32 m0, %0\n\t"
                                 "s_nop 4\n\t"
                                 "global_load_lds_dwordx4 %1, %2 sc0 sc1" ::"s"(dst), "v"(lane * 16), "s"(line)
                                 : "memory", "m0");
                }
                // the fragments have landed (LDS operations complete in order: behind them only the two writes of z may still be on
                // their way, unless a mailbox or store read -- wanted right below -- is younger still)
                if (mbox || st_due) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else lgkm_wait_n(z_due ? 2 : 0);
#pragma unroll
                for (int q = 0; q < NQ; ++q) asm volatile("" : "+v"(Fh[q]), "+v"(Fl[q]));
                if (mb_out) {
                    const int u = d - 1;
                    asm volatile("" : "+v"(pv));
                    // (zero outside the image: the operator's padding; the ring keeps older pixels there)
                    const bool in = h_out && mlane && (unsigned)(u - mrow) < (unsigned)W;
                    const uintx4 q = {in ? pv[0] : 0u, tag, in ? pv[1] : 0u, tag};
                    char *line = mb + (size_t)(h_out ? u : DUO_LINES - 2) * DUO_LINEB;
                    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(lane * 16), "v"(q), "s"(line) : "memory");
                }
                if (st_due) { // the row on its way out, second half
                    asm volatile("" : "+v"(sv[0]), "+v"(sv[1]));
                    if (zout) {
                        char *dst = zg + row_off(rs);
                        store_piece(pgo, sv[0], dst);
                        store_piece(pgo + chunkB, sv[1], dst);
                    }
                    if (zout16) {
                        char *dst = zg16 + row_off(rs) / 2;
                        const uintx2 h0 = {(unsigned)narrow_bf16(sv[0][0]) | ((unsigned)narrow_bf16(sv[0][1]) << 16),
                                           (unsigned)narrow_bf16(sv[0][2]) | ((unsigned)narrow_bf16(sv[0][3]) << 16)};
                        const uintx2 h1 = {(unsigned)narrow_bf16(sv[1][0]) | ((unsigned)narrow_bf16(sv[1][1]) << 16),
                                           (unsigned)narrow_bf16(sv[1][2]) | ((unsigned)narrow_bf16(sv[1][3]) << 16)};
                        store_piece16(pgo16, h0, dst);
                        store_piece16(pgo16 + chunkB16, h1, dst);
                    }
                }
                // hand-off in: rows 14, 15 of the upper part's diagonal d + 16 join diagonal d of this tile's ring
                if (h_in) {
                    asm volatile("" : "+v"(hq));
                    uintx4 q = __builtin_bit_cast(uintx4, hq);
                    if (!__all(q[1] == tag && q[3] == tag)) poll_line(d + 16, q);
                    if (mlane && !dead) {
                        const uintx2 v = {q[0], q[2]};
                        asm volatile("ds_write_b64 %0, %1" ::"v"(madr + (d & 1) * SLOTB), "v"(v) : "memory");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                IFL_HSTAMP(2); // memory requests, hand-off in
                // ---- z of diagonal d-1 = L^-1 r_{d-1}, for the next step to put into the tile (same products in the same order
                //      as the whole-image kernel's z product)
                if (!(IFL_EXP & 4)) {
                    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
                    zh = zero, zm = zero;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        zh = __builtin_amdgcn_mfma_f32_16x16x32_f16(Z[q][0], Fh[q], zh, 0, 0, 0);
                        zm = __builtin_amdgcn_mfma_f32_16x16x32_f16(Z[q][0], Fl[q], zm, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) zm = __builtin_amdgcn_mfma_f32_16x16x32_f16(Z[q][1], Fh[q], zm, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                IFL_HSTAMP(3); // z product
                // (a hand-off write must be in the ring before the barrier)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            };
            {
                using T = std::true_type;
                using F = std::false_type;
                int d = dfirst;
                for (; d <= Hp - 2 && d <= DEND; ++d) step(T{}, F{}, d);
                for (; d < SLAG && d <= DEND; ++d) step(F{}, F{}, d);
                for (; d <= DEND; ++d) step(F{}, T{}, d);
            }
            // (the last rows' stores and the last mailbox operations are still on their way.  Another sweep reuses the
            // tile and waits for them; the last sweep of the launched pass lets the wave run on to the verdict)
            if (!last_of_pass) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };

        __syncthreads();
        asm volatile("s_setprio " IFL_STR(IFL_PRIO_HELPER));
        // ---- one instance of the sweep, in a loop over the workgroup's sweeps (the chain waves' loop, barrier for barrier) ----
        for (int k = 0, redo = 0;; ++k) {
            if (k) {
                zero_ring();
                __syncthreads();
            }
            helper_sweep(sweep_of(redo, redo ? k - nfirst : k), !redo && k + 1 == nfirst, k == 0);
            if (redo || k + 1 < nfirst) __syncthreads();
            if (redo) {
                if (k + 1 == nfirst + nredo) break;
                continue;
            }
            if (k + 1 < nfirst) continue;
#ifdef IFL_STAMPS
            if (g_stamps && b == 0 && tid == NW * 64) {
                unsigned long long *o = g_stamps + (my_part == 1 ? 32 : 0);
                o[8] = st_slow, o[9] = st_spins, o[10] = st_gate;
                for (int j = 0; j < 7; ++j) o[16 + j] = st_h[j];
            }
#endif
            // (an upper part's last row stores are on their way: they must have reached this XCD's L2 before the release
            // fence below can send them on)
            if (my_part == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int bad = __syncthreads_or(dead << 1);
            if (my_part == 0) {
                // The verdict tells the lower part whether this tile is good.  This L2's dirty lines of z go back first
                // (agent-scope release): should the lower part redo the image -- for its own tile's sake, too -- they must
                // not land on top of the redone rows later (the two workgroups may sit on XCDs with separate L2s).
                if (!bad) reduce_amax();
                if (tid == NW * 64) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(verdict, ((unsigned long long)tag1 << 32) | (unsigned)(bad ? 2 : 1), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
            if (my_part == 1) {
                // (the upper part finished some twenty steps ago: one poll in practice; bounded all the same)
                unsigned pv = 0;
                for (int spins = 0; spins < 20000; ++spins) {
                    const unsigned long long v = __hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(v >> 32) == tag1) {
                        pv = (unsigned)v;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                bad = __syncthreads_or(bad | (pv == 1 ? 0 : 1)); // (no verdict in time counts as a failed hand-off)
            }
            if (IFL_FORCE_REDO) bad = 1;
            if (IFL_EXP) bad = 0;
            if (!bad) {
                reduce_amax();
                if (tid == NW * 64) flags[b] = 0;
                advance_generation();
                return;
            }
            // ---- redo: scaled sweeps of both tiles through the mailbox, under the second tag ------------------------------------
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            zmax = 0.f;
            dead = 0;
            redo = 1;
        }
        const int bad2 = __syncthreads_or(dead);
        if (tid == NW * 64) flags[b] = 1 + (bad2 ? 4 : 0);
        if (bad2) {
            scan_general_body<Cfg::THREADS, XT>(xin, wf32, zout, geom, rh, rw, 1, (float *)lds, b, tid, zout16);
            __syncthreads();
            if (amax) { // the rows stored above are void: take the maximum of what the redo wrote
                zmax = 0.f;
                if (zout) {
                    const float *zi = zout + (size_t)b * C * H * W;
                    for (int i = tid - NW * 64; i < C * H * W; i += NW * 64) zmax = fmaxf(zmax, fabsf(zi[i]));
                } else {
                    // (bf16 output only: the rounded values bound the maximum to half an ulp of bf16 -- the prescale it feeds
                    // is a power of two with a factor of two of headroom)
                    const bf16_t *zi = zout16 + (size_t)b * C * H * W;
                    for (int i = tid - NW * 64; i < C * H * W; i += NW * 64) zmax = fmaxf(zmax, fabsf(widen(zi[i])));
                }
            }
        }
        reduce_amax();
        advance_generation();
    }
}

// ---- launch -------------------------------------------------------------------------------------------------------
// state block: [generation per image: 128 words][mailbox: 128 images x 80 lines x 1 KiB]
static constexpr size_t DUO_MBOX_OFF = DUO_MAX_IMAGES * sizeof(unsigned);
size_t scan_duo_state_bytes() { return DUO_MBOX_OFF + (size_t)DUO_MAX_IMAGES * DUO_LINES * DUO_LINEB; }
int scan_duo_max_images() { return DUO_MAX_IMAGES; }

bool scan_duo_supported(const Geom &g)
{
    // layers of exactly 32 or 64 channels on 32-pixel rows: the helper waves move whole 128-byte lines, 8 lanes per line
    // (every other shape the MFMA scan covers runs on the whole-image kernel of scan_mfma.hip)
    return (g.C == 32 || g.C == 64) && g.W == 32 && g.H <= 32 && ((g.KH == 3 && g.KW == 3) || (g.KH == 2 && g.KW == 2));
}

template <int C, int KH, int KW, bool PAD, bool BF>
static int launch_duo(const void *x, float *z, bf16_t *z16, const void *apack, const Geom &g, int rh, int rw, int *flags,
                      const float *wf32, unsigned *amax, char *state, bool whole_image, hipStream_t s)
{
    using Cfg = DuoCfg<C, KH, KW>;
    static_assert(Cfg::LDSB <= 160 * 1024, "ring + tile must fit the CU's LDS");
    static LdsOptIn opt_in;
    if (int rc = lds_opt_in(opt_in, (const void *)k_scan_duo<C, KH, KW, PAD, BF>, Cfg::LDSB)) return rc;
    if (scan_general_lds_bytes(g) > (size_t)Cfg::LDSB)
        IFL_FAIL(IFL_EUNSUPPORTED, "launch_scan_duo: fp32 fallback does not fit the kernel's LDS");
    // an image of more than 16 rows is two tiles: two workgroups (nparts = 2), or -- whole_image -- one that sweeps both
    const int nparts = (g.H > 16 && !whole_image) ? 2 : 1;
    if (g.H > 16 && (!state || g.B > DUO_MAX_IMAGES)) IFL_FAIL(IFL_EINVAL, "launch_scan_duo: a two-tile scan needs the state block");
    SplitState sp{nullptr, nullptr};
    if (state) sp = SplitState{(unsigned long long *)(state + DUO_MBOX_OFF), (unsigned *)state};
#ifdef IFL_STAMPS
    if (const char *e = getenv("IFL_STAMPS")) {
        unsigned long long *ptr = (unsigned long long *)strtoull(e, nullptr, 0);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &ptr, sizeof(ptr));
    }
#endif
    const dim3 grid(nparts == 2 ? 16 * ((g.B + 7) / 8) : g.B);
    hipLaunchKernelGGL((k_scan_duo<C, KH, KW, PAD, BF>), grid, dim3(Cfg::THREADS), Cfg::LDSB, s, x, z, z16, (const half8 *)apack, g.H,
                       g.W, rh, rw, flags, wf32, g, amax, sp, nparts);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

int launch_scan_duo(const ScanIO &io, const void *apack, const Geom &g, int rh, int rw, int *flags, const float *wf32,
                    unsigned *amax, void *state, bool whole_image, hipStream_t s)
{
    if (!io.z32 && !io.z16) IFL_FAIL(IFL_EINVAL, "launch_scan_duo: no output");
#define IFL_CASE(CC, KK)                                                                                               \
    if (g.C == CC && g.KH == KK && g.KW == KK)                                                                         \
        return io.x16 ? launch_duo<CC, KK, KK, false, true>(io.x16, io.z32, io.z16, apack, g, rh, rw, flags, wf32, amax,      \
                                                            (char *)state, whole_image, s)                                \
                      : launch_duo<CC, KK, KK, false, false>(io.x32, io.z32, io.z16, apack, g, rh, rw, flags, wf32, amax,     \
                                                             (char *)state, whole_image, s);
    IFL_CASE(64, 3)
    IFL_CASE(32, 3)
    IFL_CASE(64, 2)
    IFL_CASE(32, 2)
#undef IFL_CASE
    IFL_FAIL(IFL_EUNSUPPORTED, "launch_scan_duo: no instantiation for C=%d K=%dx%d", g.C, g.KH, g.KW);
}

} // namespace ifl
