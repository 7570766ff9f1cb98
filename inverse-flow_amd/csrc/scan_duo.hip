// MFMA wavefront back-substitution scan, "duo" form: two waves per SIMD with different jobs (gfx950, wave64).
//
// Same mathematics, same lattice mapping, same LDS ring and the same split-fp16 arithmetic as scan_mfma.hip (right
// fold r_p = x_p - sum_t (W_t L^-1) r_{p-t}, z_p = L^-1 r_p; push form; MFMA column n = image row, walking along
// w = d - h) -- read the header of that file first.  What changes is who issues what.  A step of the round-1 kernel
// was bound by the *sum* of the issue costs of everything one wave per SIMD had to issue (54 MFMAs + ~170 vector +
// ~35 scalar + ~36 LDS instructions: ~2000 cycles for 860 cycles of matrix work).  Here a workgroup owns ONE row tile
// (16 image rows; an image of 17..32 rows is two workgroups, see "hand-off") and runs 2 x C/16 waves:
//
//   chain waves  (wave w < C/16, one per SIMD): the dependent chain and nothing else -- fragment reads of r_{d-1},
//                the 8 tap products (48 MFMAs at C = 64, 3x3), the epilogue r_d = x + acc -> split fp16 -> ring.
//                No vector-memory instruction, no scalar address arithmetic, no staging of results.
//   helper waves (wave C/16 + w, the same SIMD as chain wave w at C = 64): everything off the chain -- the z product
//                z_{d-1} = L^-1 r_{d-1} (6 MFMAs) and its staging, the image rows (loaded whole into registers PF steps
//                ahead, x quads handed to the chain waves through LDS, finished z quads taken back, rows stored as whole
//                lines: see the helper section), and the hand-off mailbox.  Their scalar, LDS and memory instructions issue
//                through ports the chain wave does not use; measured (tools/two_wave_probe.hip): 1782 cycles per step
//                for one wave doing both jobs, 1102 for the pair.
//
// One s_barrier per step joins all waves: after barrier d the ring holds r_{d-1} (chain waves), the staging holds the x
// quads of step d (helpers waited for their DMAs) and the z staged in step d-1.
//
// Hand-off (images of 17..32 rows, two workgroups i and i+8 of the grid): the upper part's helper 0 publishes rows 14
// and 15 of every finished diagonal as one 1 KiB mailbox line of 8-byte {value, tag} granules (write-through); the
// lower part's helper 0 prefetches the line by LDS-DMA, checks the tags and writes the rows into row block 0 of its ring
// (the block that is the zero padding of a whole image) one step before its chain waves read them.  Tags are launch
// generations kept in the caller's scan state (an argument of the entry points): no cleaning, valid under graph replay.  The upper
// part waits for nobody; every wait of the lower part is bounded; an image whose hand-off failed or whose r left the
// fp16 range is redone by the lower part's workgroup alone (both tiles in turn through the same mailbox, x scaled by
// 2^-12), and beyond that by the exact fp32 body.  Results are bit-identical to the whole-image kernel of scan_mfma.hip.
#include <stdlib.h>
#include <type_traits>

#include "ifl_common.h"
#include "mfma_util.h"
#include "scan_general_body.h"

namespace ifl {

template <int C, int KH, int KW> struct DuoCfg {
    static constexpr int NW = C / 16;      // chain waves = 16-channel output groups; as many helper waves
    static constexpr int NQ = C / 32;      // 32-deep k-steps per tap
    static constexpr int NT = KH * KW;     // taps incl. the diagonal one
    static constexpr int NS = NT;          // A slots of the packed weights: NT-1 folded taps + L^-1
    static constexpr int NACC = 3;         // rolling accumulators: diagonals d, d+1, d+2
    static_assert(KH + KW - 2 <= 4 && KH <= 3, "push scan: taps reach at most 4 diagonals ahead, 2 rows up");
    static constexpr int NPL = NQ * 8;     // planes per row block: (k-step, hi/lo, k-group)
    static constexpr int RBB = NPL * 256;  // one row block (16 rows x NPL planes x 16 B)
    static constexpr int SLOTB = 2 * RBB;  // one diagonal: row block 0 (rows above the tile: zero or the hand-off) + the tile
    static constexpr int RINGB = 2 * SLOTB;
    static constexpr int QG = 2;           // quads of a row that one x / z duty moves between registers and staging
    static constexpr int NXS = 2 * QG;     // x quads a row keeps staged ([row][quad % NXS][channel][4])
    static constexpr int XROWB = NXS * C * 16 + 16;
    static constexpr int XSB = 16 * XROWB;
    static constexpr int NZS = 2 * QG;     // z quads: [row][quad % NZS][channel][4]
    static constexpr int ZROWB = NZS * C * 16 + 16;
    static constexpr int ZQB = 16 * ZROWB;
    static constexpr int OFF_XS = RINGB, OFF_ZQ = OFF_XS + XSB, OFF_DUMP = OFF_ZQ + ZQB;
    static constexpr int DUMPB = 4 * 256 + 64 * NW * 8; // where chain lanes outside the image write their r
    static constexpr int NHL = 8;                        // landing slots of mailbox lines (lower part)
    static constexpr int OFF_HALO = OFF_DUMP + DUMPB;
    static constexpr int OFF_DMY = OFF_HALO + NHL * 1024; // landing of the mailbox prefetches no line is due for
    static constexpr int LDSB = OFF_DMY + NW * 1024;
    static constexpr int THREADS = 128 * NW;
    static constexpr int G = 4 / NW;       // rows per helper wave that start / finish a quad each step
    static constexpr int PF = 8;           // x quads are requested PF steps before their first use
    static constexpr int LEAD = 4;         // steps before the first pixel of a tile that takes no hand-off (its first rows: one burst)
#ifndef IFL_PFH
#define IFL_PFH 2
#endif
    static constexpr int PFH = IFL_PFH;    // mailbox lines are requested PFH steps before they are delivered
#ifndef IFL_A_VS
#define IFL_A_VS 4
#endif
#ifndef IFL_GATE
#define IFL_GATE 2
#endif
    static constexpr int GATE = IFL_GATE;  // the lower part asks for its first line once the upper part's diagonal 14 + GATE is visible
    static_assert(PFH < NHL && PF >= PFH + 2, "the sweep's lead-in covers both prefetches");
    static_assert(4 % NW == 0 && C <= 64, "one DMA / store instruction covers one image row of all channels");
};

// Development aid (tools/exp_scan.sh): what-if builds that drop one kind of work (results are then garbage) to see what
// a step is waiting for.  1: no x loads, 2: no z stores, 4: no z product, 8: no hand-off, 16: no mailbox prefetch, 32: no x
// duty, 64: no z duty, 128: no wait at a row's first use.  Any bit also compiles the redo passes out (the verdict is forced
// good), which alone is worth 1.8 us: compare what-if builds with each other (a bit without effect, e.g. 8192, is the
// baseline), not with the product.  Never defined in the product.
#ifndef IFL_EXP
#define IFL_EXP 0
#endif
#ifndef IFL_PRIO_CHAIN
#define IFL_PRIO_CHAIN 0
#endif
#ifndef IFL_PRIO_HELPER
#define IFL_PRIO_HELPER 1
#endif
#ifndef IFL_PRIO_ZPROD
#define IFL_PRIO_ZPROD 3
#endif
// cache policy of the z stores / x DMAs by number: 0 default, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0
#ifndef IFL_ST_POL
#define IFL_ST_POL 0
#endif
#ifndef IFL_LD_POL
#define IFL_LD_POL 0
#endif
#define IFL_POL_0
#define IFL_POL_1 nt
#define IFL_POL_2 sc1
#define IFL_POL_3 sc0 sc1
#define IFL_POL_4 sc1 nt
#define IFL_POL_5 sc0
#define IFL_CAT2(a, b) a##b
#define IFL_CAT(a, b) IFL_CAT2(a, b)
#define IFL_ST_POLICY IFL_CAT(IFL_POL_, IFL_ST_POL)
#define IFL_LD_POLICY IFL_CAT(IFL_POL_, IFL_LD_POL)
#define IFL_STR2(x) #x
#define IFL_STR(x) IFL_STR2(x)
// Development aid (tools/duo_stamps.py; build with HIPCC_EXTRA=-DIFL_STAMPS): timeline of image 0's two workgroups.
#ifdef IFL_STAMPS
__device__ unsigned long long *g_stamps = nullptr;
#endif

// mailbox geometry (bytes): [image][DUO_LINES][1 KiB]; line u holds rows 14, 15 of the upper part's diagonal u, the last
// line is the verdict.  Independent of the channel count (a narrower layer leaves part of a line unused).
static constexpr int DUO_LINES = 80;
static constexpr int DUO_LINEB = 1024;
static constexpr int DUO_MAX_IMAGES = 128;

// ---- the helper waves' row buffers: accumulator registers a0 .. a127, addressed by NUMBER ----------------------------------
// Buffer k, register i (16 bytes per lane) is a[4 (k NIM + i) : +3].  They are not C++ variables: the compiler knows them
// only as registers that every statement below clobbers.  As variables they were moved around between statements -- through
// scratch registers while loads were in flight when they were targets of asm loads, or by the hundred moves per step when
// nothing asynchronous targeted them.  Nothing the compiler emits in the helper waves' code uses an accumulator register
// (their MFMAs are asm with ordinary registers; tests/test_build_checks.py looks at the ISA).
#define IFL_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
// buffer <- bounce buffer (NIM reads of 1 KiB apart, complete on return)
template <int R0, int NIM> __device__ __forceinline__ void rb_take(unsigned la)
{
    if constexpr (NIM == 8)
        asm volatile("ds_read_b128 a[%1:%2], %0\n\tds_read_b128 a[%3:%4], %0 offset:1024\n\tds_read_b128 a[%5:%6], %0 offset:2048\n\t"
                     "ds_read_b128 a[%7:%8], %0 offset:3072\n\tds_read_b128 a[%9:%10], %0 offset:4096\n\tds_read_b128 a[%11:%12], %0 offset:5120\n\t"
                     "ds_read_b128 a[%13:%14], %0 offset:6144\n\tds_read_b128 a[%15:%16], %0 offset:7168\n\ts_waitcnt lgkmcnt(0)" ::"v"(la),
                     "n"(R0), "n"(R0 + 3), "n"(R0 + 4), "n"(R0 + 7), "n"(R0 + 8), "n"(R0 + 11), "n"(R0 + 12), "n"(R0 + 15), "n"(R0 + 16),
                     "n"(R0 + 19), "n"(R0 + 20), "n"(R0 + 23), "n"(R0 + 24), "n"(R0 + 27), "n"(R0 + 28), "n"(R0 + 31)
                     : "memory", IFL_AGPRS);
    else
        asm volatile("ds_read_b128 a[%1:%2], %0\n\tds_read_b128 a[%3:%4], %0 offset:1024\n\tds_read_b128 a[%5:%6], %0 offset:2048\n\t"
                     "ds_read_b128 a[%7:%8], %0 offset:3072\n\ts_waitcnt lgkmcnt(0)" ::"v"(la),
                     "n"(R0), "n"(R0 + 3), "n"(R0 + 4), "n"(R0 + 7), "n"(R0 + 8), "n"(R0 + 11), "n"(R0 + 12), "n"(R0 + 15)
                     : "memory", IFL_AGPRS);
}
// the lanes m of the buffer -> LDS, register i at la + 128 i (8 channels per instruction: a row of 32 pixels)
template <int R0, int NIM> __device__ __forceinline__ void rb_write_all(unsigned long long m, unsigned la)
{
    unsigned long long sv;
    if constexpr (NIM == 8)
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\t"
                     "ds_write_b128 %2, a[%3:%4]\n\tds_write_b128 %2, a[%5:%6] offset:128\n\tds_write_b128 %2, a[%7:%8] offset:256\n\t"
                     "ds_write_b128 %2, a[%9:%10] offset:384\n\tds_write_b128 %2, a[%11:%12] offset:512\n\tds_write_b128 %2, a[%13:%14] offset:640\n\t"
                     "ds_write_b128 %2, a[%15:%16] offset:768\n\tds_write_b128 %2, a[%17:%18] offset:896\n\ts_mov_b64 exec, %0"
                     : "=&s"(sv)
                     : "s"(m), "v"(la), "n"(R0), "n"(R0 + 3), "n"(R0 + 4), "n"(R0 + 7), "n"(R0 + 8), "n"(R0 + 11), "n"(R0 + 12), "n"(R0 + 15),
                       "n"(R0 + 16), "n"(R0 + 19), "n"(R0 + 20), "n"(R0 + 23), "n"(R0 + 24), "n"(R0 + 27), "n"(R0 + 28), "n"(R0 + 31)
                     : "memory", IFL_AGPRS);
    else
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\t"
                     "ds_write_b128 %2, a[%3:%4]\n\tds_write_b128 %2, a[%5:%6] offset:128\n\tds_write_b128 %2, a[%7:%8] offset:256\n\t"
                     "ds_write_b128 %2, a[%9:%10] offset:384\n\ts_mov_b64 exec, %0"
                     : "=&s"(sv)
                     : "s"(m), "v"(la), "n"(R0), "n"(R0 + 3), "n"(R0 + 4), "n"(R0 + 7), "n"(R0 + 8), "n"(R0 + 11), "n"(R0 + 12), "n"(R0 + 15)
                     : "memory", IFL_AGPRS);
}
// ... and back: LDS -> the lanes m of the buffer (asynchronous: s_waitcnt lgkmcnt before the registers are used)
template <int R0, int NIM> __device__ __forceinline__ void rb_read_all(unsigned long long m, unsigned la)
{
    unsigned long long sv;
    if constexpr (NIM == 8)
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\t"
                     "ds_read_b128 a[%3:%4], %2\n\tds_read_b128 a[%5:%6], %2 offset:128\n\tds_read_b128 a[%7:%8], %2 offset:256\n\t"
                     "ds_read_b128 a[%9:%10], %2 offset:384\n\tds_read_b128 a[%11:%12], %2 offset:512\n\tds_read_b128 a[%13:%14], %2 offset:640\n\t"
                     "ds_read_b128 a[%15:%16], %2 offset:768\n\tds_read_b128 a[%17:%18], %2 offset:896\n\ts_mov_b64 exec, %0"
                     : "=&s"(sv)
                     : "s"(m), "v"(la), "n"(R0), "n"(R0 + 3), "n"(R0 + 4), "n"(R0 + 7), "n"(R0 + 8), "n"(R0 + 11), "n"(R0 + 12), "n"(R0 + 15),
                       "n"(R0 + 16), "n"(R0 + 19), "n"(R0 + 20), "n"(R0 + 23), "n"(R0 + 24), "n"(R0 + 27), "n"(R0 + 28), "n"(R0 + 31)
                     : "memory", IFL_AGPRS);
    else
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\t"
                     "ds_read_b128 a[%3:%4], %2\n\tds_read_b128 a[%5:%6], %2 offset:128\n\tds_read_b128 a[%7:%8], %2 offset:256\n\t"
                     "ds_read_b128 a[%9:%10], %2 offset:384\n\ts_mov_b64 exec, %0"
                     : "=&s"(sv)
                     : "s"(m), "v"(la), "n"(R0), "n"(R0 + 3), "n"(R0 + 4), "n"(R0 + 7), "n"(R0 + 8), "n"(R0 + 11), "n"(R0 + 12), "n"(R0 + 15)
                     : "memory", IFL_AGPRS);
}
// a whole image row: global memory -> buffer, NIM full-wave loads at the lane offsets go[i] (asynchronous: vmcnt), and back
// (s_nop 4: a scalar register written by a vector instruction -- a spill reload -- is read 5 states late by a vector-memory
// instruction, and nobody inserts wait states in front of an asm statement)
template <int R0, int NIM> __device__ __forceinline__ void rb_load_row(const unsigned (&go)[NIM], const char *src)
{
    if constexpr (NIM == 8)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 a[%9:%10], %0, %8\n\tglobal_load_dwordx4 a[%11:%12], %1, %8\n\t"
                     "global_load_dwordx4 a[%13:%14], %2, %8\n\tglobal_load_dwordx4 a[%15:%16], %3, %8\n\t"
                     "global_load_dwordx4 a[%17:%18], %4, %8\n\tglobal_load_dwordx4 a[%19:%20], %5, %8\n\t"
                     "global_load_dwordx4 a[%21:%22], %6, %8\n\tglobal_load_dwordx4 a[%23:%24], %7, %8" ::"v"(go[0]),
                     "v"(go[1]), "v"(go[2]), "v"(go[3]), "v"(go[4]), "v"(go[5]), "v"(go[6]), "v"(go[7]), "s"(src), "n"(R0), "n"(R0 + 3),
                     "n"(R0 + 4), "n"(R0 + 7), "n"(R0 + 8), "n"(R0 + 11), "n"(R0 + 12), "n"(R0 + 15), "n"(R0 + 16), "n"(R0 + 19), "n"(R0 + 20),
                     "n"(R0 + 23), "n"(R0 + 24), "n"(R0 + 27), "n"(R0 + 28), "n"(R0 + 31)
                     : "memory", IFL_AGPRS);
    else
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 a[%5:%6], %0, %4\n\tglobal_load_dwordx4 a[%7:%8], %1, %4\n\t"
                     "global_load_dwordx4 a[%9:%10], %2, %4\n\tglobal_load_dwordx4 a[%11:%12], %3, %4" ::"v"(go[0]),
                     "v"(go[1]), "v"(go[2]), "v"(go[3]), "s"(src), "n"(R0), "n"(R0 + 3), "n"(R0 + 4), "n"(R0 + 7), "n"(R0 + 8), "n"(R0 + 11),
                     "n"(R0 + 12), "n"(R0 + 15)
                     : "memory", IFL_AGPRS);
}
template <int R0, int NIM> __device__ __forceinline__ void rb_store_row(const unsigned (&go)[NIM], char *dst)
{
    if constexpr (NIM == 8)
        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, a[%9:%10], %8\n\tglobal_store_dwordx4 %1, a[%11:%12], %8\n\t"
                     "global_store_dwordx4 %2, a[%13:%14], %8\n\tglobal_store_dwordx4 %3, a[%15:%16], %8\n\t"
                     "global_store_dwordx4 %4, a[%17:%18], %8\n\tglobal_store_dwordx4 %5, a[%19:%20], %8\n\t"
                     "global_store_dwordx4 %6, a[%21:%22], %8\n\tglobal_store_dwordx4 %7, a[%23:%24], %8" ::"v"(go[0]),
                     "v"(go[1]), "v"(go[2]), "v"(go[3]), "v"(go[4]), "v"(go[5]), "v"(go[6]), "v"(go[7]), "s"(dst), "n"(R0), "n"(R0 + 3),
                     "n"(R0 + 4), "n"(R0 + 7), "n"(R0 + 8), "n"(R0 + 11), "n"(R0 + 12), "n"(R0 + 15), "n"(R0 + 16), "n"(R0 + 19), "n"(R0 + 20),
                     "n"(R0 + 23), "n"(R0 + 24), "n"(R0 + 27), "n"(R0 + 28), "n"(R0 + 31)
                     : "memory", IFL_AGPRS);
    else
        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, a[%5:%6], %4\n\tglobal_store_dwordx4 %1, a[%7:%8], %4\n\t"
                     "global_store_dwordx4 %2, a[%9:%10], %4\n\tglobal_store_dwordx4 %3, a[%11:%12], %4" ::"v"(go[0]),
                     "v"(go[1]), "v"(go[2]), "v"(go[3]), "s"(dst), "n"(R0), "n"(R0 + 3), "n"(R0 + 4), "n"(R0 + 7), "n"(R0 + 8), "n"(R0 + 11),
                     "n"(R0 + 12), "n"(R0 + 15)
                     : "memory", IFL_AGPRS);
}
// one register of a buffer under a lane mask: -> LDS, <- LDS (asynchronous), -> global memory
template <int R0> __device__ __forceinline__ void rb_write_one(unsigned long long m, unsigned la)
{
    unsigned long long sv;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b128 %2, a[%3:%4]\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv)
                 : "s"(m), "v"(la), "n"(R0), "n"(R0 + 3)
                 : "memory", IFL_AGPRS);
}
template <int R0> __device__ __forceinline__ void rb_read_one(unsigned long long m, unsigned la)
{
    unsigned long long sv;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_read_b128 a[%3:%4], %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv)
                 : "s"(m), "v"(la), "n"(R0), "n"(R0 + 3)
                 : "memory", IFL_AGPRS);
}
template <int R0> __device__ __forceinline__ void rb_store_one(unsigned long long m, unsigned go, char *dst)
{
    // (s_nop 4: a scalar register written by a vector instruction -- a spill reload -- is read 5 states late by a
    // vector-memory instruction, and nobody inserts wait states in front of an asm statement)
    unsigned long long sv;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_nop 4\n\tglobal_store_dwordx4 %2, a[%4:%5], %3 " IFL_STR(IFL_ST_POLICY) "\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(sv)
                 : "s"(m), "v"(go), "s"(dst), "n"(R0), "n"(R0 + 3)
                 : "memory", IFL_AGPRS);
}

template <int C, int KH, int KW, bool PAD>
__global__ __launch_bounds__(128 * (C / 16)) void k_scan_duo(const float *__restrict__ xin, float *__restrict__ zout,
                                                             const half8 *__restrict__ apack, const int H, const int W,
                                                             const int rh, const int rw, int *__restrict__ flags,
                                                             const float *__restrict__ wf32, const Geom geom,
                                                             unsigned *__restrict__ amax, const SplitState sp, const int nparts)
{
    using Cfg = DuoCfg<C, KH, KW>;
    constexpr int NW = Cfg::NW, NQ = Cfg::NQ, NS = Cfg::NS, RBB = Cfg::RBB, SLOTB = Cfg::SLOTB;
    constexpr int PF = Cfg::PF, PFH = Cfg::PFH, NXS = Cfg::NXS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_chain = wave < NW;
    const int wv = is_chain ? wave : wave - NW; // 16-channel group of this wave (chain: its outputs; helper: its z product)
    const int n = lane & 15, g = lane >> 4;
    const bool split = nparts == 2;
    const int b = split ? (int)((blockIdx.x >> 4) * 8 + (blockIdx.x & 7)) : (int)blockIdx.x;
    const int my_part = split ? (int)((blockIdx.x >> 3) & 1) : -1;
    if (b >= geom.B) return;

    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const int Cr = PAD ? geom.C : C;
    const int c0 = 16 * wv + 4 * g; // C/D layout: lane (n, g) owns channels c0..c0+3 of pixel row n

    // hand-off state
    unsigned gen0 = 0;
    char *mb = nullptr;
    if (split) {
        gen0 = __builtin_amdgcn_readfirstlane(sp.gen[b]);
        mb = (char *)sp.mbox + (size_t)b * DUO_LINES * DUO_LINEB;
    }
    unsigned long long *const verdict = (unsigned long long *)(mb + (size_t)(DUO_LINES - 1) * DUO_LINEB);

    float rmax = 0.f; // chain: max |r| this lane put into the ring (beyond the fp16 range the image is redone)
    float zmax = 0.f; // helper: max |z| this lane stored (the weight-gradient kernel's prescale)
    int dead = 0;     // helper 0 of a lower part: the upper part never showed up (bounded spin ran out)

    auto zero_ring = [&]() {
        const floatx4 zz = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid * 16; i < Cfg::RINGB; i += Cfg::THREADS * 16) *(floatx4 *)(lds + i) = zz;
    };
    zero_ring();
    if (PAD) { // padded channels are never loaded: their x staging must read as zero (finite times a zero weight)
        const floatx4 zz = {0.f, 0.f, 0.f, 0.f};
        for (int i = Cfg::OFF_XS + tid * 16; i < Cfg::OFF_ZQ; i += Cfg::THREADS * 16) *(floatx4 *)(lds + i) = zz;
    }

#ifdef IFL_STAMPS
    unsigned long long st_rt[4] = {__builtin_amdgcn_s_memrealtime(), 0, 0, 0}, st_mt[2] = {0, 0}, st_bar = 0;
#endif

    // One sweep over the row tile starting at image row hoff (Hp rows): steps d = -PF .. Hp + W, one barrier each.
    //   publish / consume: this tile hands its last two rows down / receives the two rows above it (tag: this launch's)
    if (is_chain) {
        // =================================== chain waves ===================================================
        half8 A[NS - 1][NQ][2]; // folded taps as A fragments (hi, lo), in AGPRs for the whole kernel
        {
            half8 Aload[NS - 1][NQ][2]; // all loads first, then the pins (a pin behind its load serialises the round trips)
#pragma unroll
            for (int s = 0; s < NS - 1; ++s)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl)
                        Aload[s][q][hl] = apack[((((size_t)wv * NS + s) * NQ + q) * 2 + hl) * 64 + lane];
#pragma unroll
            for (int s = 0; s < NS - 1; ++s)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl) {
                        A[s][q][hl] = Aload[s][q][hl];
                        // (pinned, or hipcc re-loads the weights inside the loop.  The accumulator half holds 128 registers
                        // per lane at two waves per SIMD: the low parts of the first IFL_A_VS taps stay in ordinary registers,
                        // where an MFMA reads them just as well -- left to itself the compiler parks them there anyway and
                        // copies them into a scratch register in front of every use)
                        if (NW == 4 && hl == 1 && s < IFL_A_VS) asm volatile("" : "+v"(A[s][q][hl]));
                        else asm volatile("" : "+a"(A[s][q][hl]));
                    }
        }
        unsigned radr[KH]; // LDS address (slot 0) of this lane's B piece for a source dh rows up
#pragma unroll
        for (int dh = 0; dh < KH; ++dh) {
            const int hs = n - dh + 16; // row block 0 of a slot holds the rows above the tile
            radr[dh] = ldsbase + (hs / 16) * RBB + g * 256 + (hs % 16) * 16;
        }
        const int wadr = RBB + (((c0 / 32) * 2) * 4 + (c0 % 32) / 8) * 256 + n * 16 + ((c0 % 8) / 4) * 8;
        const unsigned xadr = ldsbase + Cfg::OFF_XS + n * Cfg::XROWB + c0 * 16;
        int Ws = W;
        asm volatile("" : "+s"(Ws));

        auto chain_sweep = [&](const int Hp, const float xscale, const int dfirst) {
            const bool hval = n < Hp;
            const int ND = Hp + W - 1;
            floatx4 ahi[Cfg::NACC], amid[Cfg::NACC];
            half8 F2h[NQ], F2l[NQ]; // fragments of the source rows two up, carried to the next step
#pragma unroll
            for (int k = 0; k < Cfg::NACC; ++k) {
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

            auto step = [&](const int d) {
                constexpr int NDH = KH < 2 ? KH : 2;
                constexpr int PER = NQ * 2; // reads per fragment set
                constexpr int GM = 3 * NQ;  // MFMAs of one tap
                constexpr int NREQ = 2 + NDH * PER;
                constexpr int NRD2 = KH > 2 ? PER : 0;
                const int srcoff = ((d + 1) & 1) * SLOTB; // ring slot of diagonal d-1
                const int dstoff = (d & 1) * SLOTB;       // ring slot of diagonal d
                // r of diagonal d-1 complete in LDS (this wave's ring writes drained: lgkmcnt), x of this step landed
#ifdef IFL_STAMPS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
#endif
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef IFL_STAMPS
                st_bar += __builtin_amdgcn_s_memtime() - tb0;
#endif
                const int w0 = d - n;
                const unsigned xa = xadr + ((w0 >> 2) & (NXS - 1)) * (C * 16) + (rw ? 3 - (w0 & 3) : (w0 & 3)) * 4;
                floatx2 xq[2];
                half8 Fh[2][NQ], Fl[2][NQ];
                auto request = [&](int j) {
                    int c = 0;
                    if (c++ == j) lds_read2_f32<0>(xq[0], xa);
                    if (c++ == j) lds_read2_f32<8>(xq[1], xa);
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
                };
                __builtin_amdgcn_sched_barrier(0);
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
                // scheduling pattern for the region since the last fence: `lead` MFMAs, then NM x (1 MFMA, NV others)
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

                // ---- leading: taps (2, dw) of r_{d-2} (fragments kept from the previous step) -> diagonals d + dw; they
                //      fill the matrix pipe while this step's LDS requests (one behind each MFMA) are in flight
                if constexpr (KH > 2) {
                    constexpr int NLEADM = (KW > 1 ? 2 : 1) * GM;
                    constexpr int RPM = (NREQ + NLEADM - 1) / NLEADM;
#pragma unroll
                    for (int k = 0; k < NLEADM; ++k) {
                        if (k < GM) mf_one(2 * KW + 0, F2h, F2l, 0, k);
                        else mf_one(2 * KW + 1, F2h, F2l, 1, k - GM);
                        // (MFMAs are pure: tie the result to an opaque statement, or they sink below the requests)
                        asm volatile("" : "+a"(ahi[k < GM ? 0 : 1]), "+a"(amid[k < GM ? 0 : 1]));
                        fence();
#pragma unroll
                        for (int j = k * RPM; j < (k + 1) * RPM && j < NREQ; ++j) request(j);
                        fence();
                    }
                    if constexpr (KW > 2) mf(2 * KW + 2, F2h, F2l, 2, true); // first contribution to diagonal d+2
                    fence();
                } else {
#pragma unroll
                    for (int j = 0; j < NREQ; ++j) request(j);
                    fence();
                }
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
                    lgkm_wait_n((NDH - 1) * PER); // x and the dh=0 fragments have landed
#pragma unroll
                    for (int k = 0; k < GM; ++k) {
                        mf_one(1, Fh[0], Fl[0], 0, k);
                        if constexpr (KH > 2) {
                            asm volatile("" : "+a"(ahi[0]), "+a"(amid[0]));
                            fence();
                            if (k < PER) request2(k);
                            fence();
                        }
                    }
                    fence();
                }
                if constexpr (KH > 1) {
                    lgkm_wait_n(NRD2); // ... and the dh=1 fragments
                    mf(KW, Fh[1], Fl[1], 0);
                    fence();
                }
                // ---- trailing: the remaining taps of r_{d-1} (targets d+1, d+2) with the chain's epilogue woven in.
                //      The accumulators rotate first (d+1 becomes the head, a fresh one joins for d+3).
                const floatx4 head_hi = ahi[0], head_mid = amid[0];
                ahi[0] = ahi[1];
                amid[0] = amid[1];
                ahi[1] = ahi[2];
                amid[1] = amid[2]; // ([2] is dead until the tap that opens the next diagonal initialises it)
                int ntap = 0;
#pragma unroll
                for (int dh = 0; dh < NDH; ++dh)
#pragma unroll
                    for (int dw = 0; dw < KW; ++dw)
                        if (dh + dw >= 2) {
                            // (without a dh=2 row the farthest tap is this one: it opens its diagonal)
                            mf(dh * KW + dw, Fh[dh], Fl[dh], dh + dw - 2, KH < 3 && dh + dw == KH + KW - 2);
                            ++ntap;
                        }
                // epilogue: r_d = x + acc -> split fp16 -> ring (lanes outside the image write to the dump: no branch)
                {
                    const bool valid = hval && (unsigned)w0 < (unsigned)Ws;
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
                    unsigned char *rp = valid ? lds + dstoff + wadr : lds + Cfg::OFF_DUMP + tid * 8;
                    *(half4 *)rp = hi;
                    *(half4 *)(rp + 4 * 256) = lo;
                    const float m = fmaxf(fmaxf(fabsf(rv[0]), fabsf(rv[1])), fmaxf(fabsf(rv[2]), fabsf(rv[3])));
                    rmax = valid ? fmaxf(rmax, m) : rmax;
                }
                constexpr int NTR = GM * (KH * KW - 1 - (KW > 1 ? 1 : 0) - (KH > 1 ? 1 : 0) - (KH > 2 ? KW : 0));
                static_assert(NTR >= 0, "trailing taps");
                if constexpr (NTR > 4)
                    weave(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, NTR - 2>{},
                          std::integral_constant<int, 3>{});
                else
                    fence();
                (void)ntap;
            };

            int d = dfirst;
            for (; d < -2; ++d) asm volatile("s_barrier" ::: "memory"); // (the helpers' lead-in: first x quads, first lines)
            for (; d <= ND; ++d) step(d);
            for (; d <= ND + 1; ++d) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };

        // ---- the chain waves' view of the kernel's control flow (the helpers mirror it barrier for barrier) ----
        __syncthreads();
        asm volatile("s_setprio " IFL_STR(IFL_PRIO_CHAIN));
#ifdef IFL_STAMPS
        st_rt[1] = __builtin_amdgcn_s_memrealtime();
        st_mt[0] = __builtin_amdgcn_s_memtime();
#endif
        // (a tile that takes no hand-off has a short lead-in: its helpers start the first rows' loads in one burst)
        chain_sweep(my_part == 1 ? H - 16 : (H < 16 ? H : 16), 1.0f, my_part == 1 ? -PF : -Cfg::LEAD);
#ifdef IFL_STAMPS
        st_mt[1] = __builtin_amdgcn_s_memtime();
        st_rt[2] = __builtin_amdgcn_s_memrealtime();
        if (g_stamps && b == 0 && tid == 0) {
            unsigned long long *o = g_stamps + (my_part == 1 ? 32 : 0);
            o[0] = st_rt[0], o[1] = st_rt[1], o[2] = st_rt[2], o[3] = st_mt[1] - st_mt[0];
            o[4] = (unsigned long long)((my_part == 1 ? H - 16 : (H < 16 ? H : 16)) + W + 1 + PF);
            o[5] = st_bar;
        }
#endif
        int bad = __syncthreads_or(rmax < 6.0e4f ? 0 : 1);
        if (my_part == 0) return; // (the verdict is the helpers' business)
        if (my_part == 1) bad = __syncthreads_or(0); // helper 0 adds the upper part's verdict
        if (IFL_EXP) bad = 0;
        if (!bad) return;
        // redo, x scaled by 2^-12: both tiles in turn (or the single tile), then -- beyond that -- exact fp32
        const int ntile = split ? 2 : 1;
        rmax = 0.f;
        for (int t = 0; t < ntile; ++t) {
            zero_ring();
            __syncthreads();
            chain_sweep(split ? (t ? H - 16 : 16) : H, 1.0f / 4096.0f, (split && t) ? -PF : -Cfg::LEAD);
            __syncthreads();
        }
        const int bad2 = __syncthreads_or(rmax < 6.0e4f ? 0 : 1);
        if (bad2) {
            scan_general_body<Cfg::THREADS>(xin, wf32, zout, geom, rh, rw, 1, (float *)lds, b, tid);
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
        const unsigned fadr = ldsbase + RBB + g * 256 + n * 16; // this lane's B piece of the tile's row n (slot 0)
        const unsigned zadr = ldsbase + Cfg::OFF_ZQ + n * Cfg::ZROWB + c0 * 16;
        // ---- x and z of this wave's image rows live in REGISTERS, one full 128-byte line per (channel, row) ------------
        // Helper w owns the tile rows RPH w .. RPH w + RPH - 1, as buffers k = 0 .. RPH-1 of NIM 16-byte registers per lane
        // (the accumulator registers a0 .. a127, by number: see rb_*).  A row is loaded whole, PFR steps before its first
        // pixel is due, by NIM instructions in which 8 consecutive lanes cover one (channel, row) line of 32 pixels: every
        // request is a full line, fetched once (lane = 8 channels x 8 quads; the round-1 kernel asked for 16 bytes of a line
        // every fourth step and the L2 had lost the line by then: 5.6 x the algorithmic fetch, and its partial-line stores
        // were the most expensive thing in the kernel).  Every fourth step the lanes that hold the row's next quad write it to
        // the x staging, PFX steps before the chain waves need it; PFX + 7 steps later the same lanes and registers receive
        // the finished z quad from the z staging, and when the last quad is in, the row goes out as whole lines.  No LDS-DMA,
        // no partial line ever moves.  The step loop is unrolled by four: the row that is due depends on the step modulo 4
        // only, so the registers are named at compile time.
        static_assert(!PAD, "the duo scan takes layers of exactly 32 or 64 channels (launch_scan_mfma routes the others)");
        constexpr int RPH = 16 / NW, NIM = C / 8, PFR = Cfg::PF, PFX = 2, NQL = 8; // (W = 32: 8 quads per row)
        const int lq = lane & 7, lc = lane >> 3; // this lane's quad of the line and channel within the instruction
        constexpr int QG = Cfg::QG;
        static_assert(Cfg::NXS == Cfg::NZS && NQL % QG == 0, "one staging offset per lane serves both duties");
        constexpr unsigned long long QMASK = 0x0101010101010101ull * ((1u << QG) - 1u); // QG neighbouring lanes of every 8
        // where this lane's quad goes inside a row's staging: slot = its (logical) quad index modulo the slots
        const unsigned qs_lane = (unsigned)(((rw ? NQL - 1 - lq : lq) & (Cfg::NXS - 1)) * (C * 16) + lc * 16);
        unsigned go[NIM];                        // byte offset inside an image of what this lane moves in instruction i
#pragma unroll
        for (int i = 0; i < NIM; ++i) go[i] = (unsigned)((8 * i + lc) * H * W * 4 + lq * 16);
        const char *xg = (const char *)xin + (size_t)b * Cr * H * W * sizeof(float);
        char *zg = (char *)zout + (size_t)b * Cr * H * W * sizeof(float);
        const unsigned dmy = __builtin_amdgcn_readfirstlane(ldsbase + Cfg::OFF_DMY + wv * 1024);
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
        // "all but the n youngest vector-memory operations are complete" for a run-time n (an immediate in the instruction).
        // The counts that occur are small (see `behind`): a short switch, anything else waits for everything.
        auto wait_vm = [&](int n) {
#define IFL_V(N) \
    case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
            if constexpr (16 / NW == 4) {
                switch (n) {
                    IFL_V(1) IFL_V(2) IFL_V(3) IFL_V(4) IFL_V(5) IFL_V(6) IFL_V(7)
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                }
            } else {
                switch (n) {
                    IFL_V(1) IFL_V(2) IFL_V(3) IFL_V(4) IFL_V(5) IFL_V(6) IFL_V(7) IFL_V(8) IFL_V(9) IFL_V(10) IFL_V(11) IFL_V(12)
                    IFL_V(13) IFL_V(14) IFL_V(15) IFL_V(16) IFL_V(17) IFL_V(18) IFL_V(19)
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                }
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
            if (!split) return;
            if (gen0 >= 0xFFFFFFF0u) {
                const floatx4 zz = {0.f, 0.f, 0.f, 0.f};
                for (int i = (tid - NW * 64) * 16; i < DUO_LINES * DUO_LINEB; i += NW * 64 * 16) *(floatx4 *)(mb + i) = zz;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            }
            if (tid == NW * 64) sp.gen[b] = gen0 >= 0xFFFFFFF0u ? 0u : gen0 + 2;
        };

        __syncthreads();
        asm volatile("s_setprio " IFL_STR(IFL_PRIO_HELPER));
        const unsigned tag1 = gen0 + 1, tag2 = gen0 + 2;
        const int ntile = split ? 2 : 1;
        // pass 0: the tile this workgroup was launched for; passes 1, 2: the redo of the whole image (see the chain waves)
        for (int pass = 0; pass < 3; ++pass) {
            int hoff, Hp;
            float zscale;
            bool publish, consume;
            unsigned tag;
            if (pass == 0) {
                hoff = my_part == 1 ? 16 : 0;
                Hp = my_part == 1 ? H - 16 : (H < 16 ? H : 16);
                zscale = 1.0f;
                publish = my_part == 0;
                consume = my_part == 1;
                tag = tag1;
            } else {
                const int t = pass - 1;
                if (t >= ntile) break;
                zero_ring();
                __syncthreads();
                hoff = t ? 16 : 0;
                Hp = split ? (t ? H - 16 : 16) : H;
                zscale = 4096.0f;
                publish = split && t == 0;
                consume = split && t == 1;
                tag = tag2;
            }
            // ================================ one sweep over the tile ==================================================
            {
                const int ND = Hp + W - 1;
                const int u_last = W + 14;  // last upper diagonal with a pixel in row 15
                const int dl_last = W - 2;  // ... as a diagonal of the lower tile
                const bool mbox = w_mbox && (publish || consume) && !(IFL_EXP & 8);
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

                // What the first use of buffer k (PFR - PFX steps after its loads went out) does NOT wait for: the operations
                // this wave issued in the last three steps -- one mailbox operation per step of the mailbox helper's
                // PFR - PFX steps in between and of the step itself, and the loads of its rows k + 4 .. k + 6 (only a wave with eight rows has
                // such).  Rows k + 1 .. k + 3 went out 5, 4, 3 steps ago: they are waited for too, which costs nothing
                // (they have landed) and keeps the set of counts small.
                int behind[RPH];
#pragma unroll
                for (int k = 0; k < RPH; ++k) {
                    int n = (mbox ? PFR - PFX + 1 : 0); // (the x duty comes behind its own step's mailbox operation)
#pragma unroll
                    for (int k2 = k + 4; k2 < RPH; ++k2)
                        if (k2 - k <= PFR - PFX && RPH * wv + k2 < Hp) n += NIM;
                    behind[k] = n;
                }
                // A tile that takes no hand-off starts LEAD steps before its first pixel instead of PFR: the rows the
                // skipped steps would have loaded (rows 0 .. PFR - LEAD - 1, all of them helper 0's) go out in one burst here.
                constexpr int LEAD = Cfg::LEAD, NBURST = PFR - LEAD;
                static_assert(NBURST == 4 && LEAD % 4 == 0 && RPH >= NBURST, "burst = buffers 0..3 of helper 0");
                const int dfirst = consume ? -PFR : -LEAD;
                if (!consume && wv == 0 && !(IFL_EXP & 1)) {
                    if (0 < Hp) rb_load_row<0 * 4 * NIM, NIM>(go, xg + row_off(0));
                    if (1 < Hp) rb_load_row<1 * 4 * NIM, NIM>(go, xg + row_off(1));
                    if (2 < Hp) rb_load_row<2 * 4 * NIM, NIM>(go, xg + row_off(2));
                    if (3 < Hp) rb_load_row<3 * 4 * NIM, NIM>(go, xg + row_off(3));
#pragma unroll
                    for (int k = 0; k < NBURST; ++k) {
                        // not waited for at row k's first use (step k - PFX): the rows the loop loaded in the last three steps
                        // (r = k + 4 .. k + 6, this wave's if it has eight rows) and one mailbox store per finished step
                        int n = 0;
                        if (RPH > NBURST)
                            for (int r = (k + 4 > NBURST ? k + 4 : NBURST); r <= k + PFR - PFX && r < RPH; ++r)
                                if (r < Hp) n += NIM;
                        if (publish && mbox) n += k + LEAD - PFX + 1;
                        behind[k] = n;
                    }
                }

                // Every per-step condition is a range of d: one unsigned compare each, "(unsigned)(d - lo) < n" with n = 0
                // when the sweep does not have that duty (the conditions are wave-uniform; written as conjunctions the
                // compiler keeps each of them as a 64-bit lane mask and the helper's step is mostly scalar mask algebra)
                const unsigned n_hin = (consume && mbox) ? (unsigned)(dl_last + 3) : 0u;                  // d in [-2, dl_last]
                const unsigned n_hout = (publish && mbox) ? (unsigned)(u_last - 13) : 0u;                 // d - 1 in [14, u_last]
                const unsigned n_zprod = (IFL_EXP & 4) ? 0u : (unsigned)ND;                               // d in [1, ND]
                const int l_lo = RPH * wv - PFR;                                                          // r = d + PFR in this wave's rows
                const unsigned n_load = (IFL_EXP & 1) ? 0u : (unsigned)((Hp - RPH * wv) < 0 ? 0 : ((Hp - RPH * wv) < RPH ? (Hp - RPH * wv) : RPH));
                const bool mb_in = consume && mbox, mb_out = publish && mbox;

                // ---- one step; P = d mod 4 (compile-time: it names the row buffers that are due) ----
                auto step = [&](auto p_c, const int d) {
                    constexpr int P = decltype(p_c)::value;
                    IFL_HSTAMP(6); // (the wait at the end of the previous step)
                    asm volatile("s_barrier" ::: "memory");
                    IFL_HSTAMP(0); // barrier
                    const bool h_in = (unsigned)(d + 2) < n_hin && !dead;
                    const bool h_out = (unsigned)(d - 15) < n_hout;
                    const bool zprod = (unsigned)(d - 1) < n_zprod;
                    // ---- LDS requests whose data the step needs: the mailbox line that landed, the fragments of r_{d-1}
                    floatx4_ hq;
                    if (h_in) {
                        // (younger than that line's DMA: the prefetches of PFH - 1 steps)
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PFH - 1) : "memory");
                        lds_read_f32x4(hq, ldsbase + Cfg::OFF_HALO + (d & (Cfg::NHL - 1)) * 1024 + lane * 16);
                    }
                    IFL_HSTAMP(1); // requests
                    // ---- load duty: the row whose first pixel is PFR steps away (this wave's rows come up in consecutive steps)
                    {
                        const int r = d + PFR; // (r mod 4 = P)
                        if ((unsigned)(d - l_lo) < n_load) {
                            const char *src = xg + row_off(r);
                            if constexpr (RPH == 4) {
                                rb_load_row<4 * P * NIM, NIM>(go, src);
                            } else {
                                if ((r >> 2) & 1) rb_load_row<4 * (P + 4) * NIM, NIM>(go, src);
                                else rb_load_row<4 * P * NIM, NIM>(go, src);
                            }
                        }
                    }
                    // ---- the mailbox helper issues exactly one vector-memory operation per step besides its rows:
                    //      the line to be delivered PFH steps from now (lower part) ...
                    if constexpr (((-2 - PFH) % 4 + 4) % 4 == P) if (mb_in && d == -2 - PFH) {
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
                    if (mb_in && !(IFL_EXP & 16)) {
                        const int dl = d + PFH;
                        const bool ok = (unsigned)(dl + 2) < n_hin;
                        const char *line = mb + (size_t)(ok ? dl + 16 : 0) * DUO_LINEB;
                        const unsigned dst = __builtin_amdgcn_readfirstlane(ok ? ldsbase + Cfg::OFF_HALO + (dl & (Cfg::NHL - 1)) * 1024 : dmy);
                        asm volatile("s_mov_b32 m0, %0\n\t"
                                     "s_nop 4\n\t"
                                     "global_load_lds_dwordx4 %1, %2 sc0 sc1" ::"s"(dst), "v"(lane * 16), "s"(line)
                                     : "memory", "m0");
                    }
                    // (this wave's own LDS requests go out behind its memory instructions: the chain waves' fragment reads,
                    // issued right after the barrier, are ahead of them in the LDS queue)
                    half8 Fh[NQ], Fl[NQ];
                    if (zprod) {
                        const unsigned fa = fadr + ((d + 1) & 1) * SLOTB;
                        lds_read_b128_o<0>(Fh[0], fa);
                        lds_read_b128_o<4 * 256>(Fl[0], fa);
                        if constexpr (NQ == 2) {
                            lds_read_b128_o<8 * 256>(Fh[1], fa);
                            lds_read_b128_o<12 * 256>(Fl[1], fa);
                        }
                    }
                    uintx2 pv;
                    if (h_out)
                        asm volatile("ds_read_b64 %0, %1" : "=v"(pv) : "v"(madr + RBB + ((d - 1) & 1) * SLOTB) : "memory");
                    IFL_HSTAMP(2); // row load + x quad
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    IFL_HSTAMP(3); // LDS wait
                    // ---- hand-off in: rows 14, 15 of the upper part's diagonal d + 16 join diagonal d of this tile's ring
                    if (h_in) {
                        asm volatile("" : "+v"(hq));
                        uintx4 q = __builtin_bit_cast(uintx4, hq);
                        if (!__all(q[1] == tag && q[3] == tag)) poll_line(d + 16, q);
                        if (mlane && !dead) {
                            const uintx2 v = {q[0], q[2]};
                            asm volatile("ds_write_b64 %0, %1" ::"v"(madr + (d & 1) * SLOTB), "v"(v) : "memory");
                        }
                    }
                    // ---- z of diagonal d-1 = L^-1 r_{d-1}: this wave's six MFMAs go ahead of the chain wave's (they are few)
                    floatx4 zh, zm;
                    if (zprod) {
                        if constexpr (NQ == 2)
                            asm volatile("s_setprio " IFL_STR(IFL_PRIO_ZPROD) : "+v"(Fh[0]), "+v"(Fl[0]), "+v"(Fh[1]), "+v"(Fl[1]));
                        else
                            asm volatile("s_setprio " IFL_STR(IFL_PRIO_ZPROD) : "+v"(Fh[0]), "+v"(Fl[0]));
                        // (as asm with the accumulators in ordinary registers: the row buffers fill the accumulator half of the
                        // register file.  The first product of each accumulator takes the constant 0: a register zeroed by a
                        // vector instruction just before would be read too early -- nobody inserts wait states around an asm
                        // MFMA.  Same products in the same order as the chain wave's z product of scan_mfma.hip: bit-identical.)
                        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(zh) : "v"(Z[0][0]), "v"(Fh[0]));
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(zm) : "v"(Z[0][0]), "v"(Fl[0]));
#pragma unroll
                        for (int q = 1; q < NQ; ++q) {
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(zh) : "v"(Z[q][0]), "v"(Fh[q]));
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(zm) : "v"(Z[q][0]), "v"(Fl[q]));
                        }
#pragma unroll
                        for (int q = 0; q < NQ; ++q) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(zm) : "v"(Z[q][1]), "v"(Fh[q]));
                        // (the results are read by vector instructions below: the wait states the compiler would insert)
                        asm volatile("s_setprio " IFL_STR(IFL_PRIO_HELPER) "\n\ts_nop 7\n\ts_nop 7" : "+v"(zh), "+v"(zm));
                    }
                    IFL_HSTAMP(4); // hand-off in + z product
                    // ---- ... or the line of the diagonal the chain waves finished in the previous step (upper part; a spare
                    //      line takes the steps without one: the operation count stays exact)
                    if (mb_out) {
                        const int u = d - 1;
                        // (zero outside the image: the operator's padding; the ring keeps older pixels there)
                        const bool in = h_out && mlane && (unsigned)(u - mrow) < (unsigned)W;
                        const uintx4 q = {in ? pv[0] : 0u, tag, in ? pv[1] : 0u, tag};
                        char *line = mb + (size_t)(h_out ? u : DUO_LINES - 2) * DUO_LINEB;
                        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(lane * 16), "v"(q), "s"(line) : "memory");
                    }
                    // ---- x duty: the next QG quads of this wave's row(s) r = d + PFX (mod 4) -> x staging [row][quad % NXS][channel][4]
                    //      (every QG-th time the row is due: the duty's LDS instructions carry 8 QG lanes each).  Here, behind the z
                    //      product, and not at the head of the step: in front of the wait for the r fragments its eight writes
                    //      delayed the z product, and with it the barrier, in almost every step (some helper has a row due).
#pragma unroll
                    for (int j = 0; j < RPH / 4; ++j) {
                        constexpr int KX0 = (P + PFX) & 3;
                        const int r = RPH * wv + KX0 + 4 * j;
                        const int v = d + PFX - r, ql = v >> 2;
                        if ((unsigned)v < (r < Hp ? 4u * NQL : 0u) && (ql & (QG - 1)) == 0 && !(IFL_EXP & 32)) { // (wave-uniform)
                            if (ql == 0 && !(IFL_EXP & 128)) wait_vm(behind[KX0 + 4 * j]); // the row has landed (first use)
                            const int p0 = rw ? NQL - QG - ql : ql; // lowest lane quad of the group
                            const unsigned la = ldsbase + Cfg::OFF_XS + r * Cfg::XROWB + qs_lane;
                            if (j == 0) rb_write_all<4 * KX0 * NIM, NIM>(QMASK << p0, la);
                            else rb_write_all<4 * ((KX0 + 4) % RPH) * NIM, NIM>(QMASK << p0, la);
                        }
                    }
                    // ---- z -> staging, at that column's in-row offset (columns outside the image land in quads that are
                    //      not live: before a row's first quad, or in the parity its last one does not use)
                    if (zprod) {
                        const int wz = d - 1 - n;
                        const unsigned za = zadr + ((wz >> 2) & (Cfg::NZS - 1)) * (C * 16) + (rw ? 3 - (wz & 3) : (wz & 3)) * 4;
                        float zv[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) zv[r] = (zh[r] + zm[r] * LO_INV) * zscale;
                        asm volatile("ds_write2_b32 %0, %1, %2 offset0:0 offset1:4\n\tds_write2_b32 %0, %3, %4 offset0:8 offset1:12" ::"v"(za),
                                     "v"(zv[0]), "v"(zv[1]), "v"(zv[2]), "v"(zv[3])
                                     : "memory");
                        // max |z| from the staged values: a column outside the image repeats an older pixel of its row (the
                        // ring keeps it) or is zero, so the maximum over everything staged is the maximum over the image
                        zmax = fmaxf(fmaxf(zmax, fabsf(zv[0])), fmaxf(fabsf(zv[1]), fmaxf(fabsf(zv[2]), fabsf(zv[3]))));
                    }
                    // ---- z duty: this wave's row(s) r = d - 5 (mod 4) completed a quad of z with diagonal d-2 (staged one step
                    //      ago); when it is the last of a group of QG they join the registers their x came from, and the row's
                    //      last group sends the row out
#pragma unroll
                    for (int j = 0; j < RPH / 4; ++j) {
                        constexpr int KZ0 = (P + 3) & 3; // (d - 5) mod 4
                        const int r = RPH * wv + KZ0 + 4 * j;
                        const int v = d - 5 - r, ql = v >> 2;
                        if ((unsigned)v < (r < Hp ? 4u * NQL : 0u) && (ql & (QG - 1)) == QG - 1 && !(IFL_EXP & 64)) { // (wave-uniform)
                            const int p0 = rw ? NQL - 1 - ql : ql - (QG - 1);
                            const unsigned la = ldsbase + Cfg::OFF_ZQ + r * Cfg::ZROWB + qs_lane;
                            if (j == 0) rb_read_all<4 * KZ0 * NIM, NIM>(QMASK << p0, la);
                            else rb_read_all<4 * ((KZ0 + 4) % RPH) * NIM, NIM>(QMASK << p0, la);
                            if (ql == NQL - 1 && !(IFL_EXP & 2)) {
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                char *dst = zg + row_off(r);
                                if (j == 0) rb_store_row<4 * KZ0 * NIM, NIM>(go, dst);
                                else rb_store_row<4 * ((KZ0 + 4) % RPH) * NIM, NIM>(go, dst);
                            }
                        }
                    }
                    IFL_HSTAMP(5); // z quad + staging
                    // (everything this wave put into the LDS is done before the barrier: operations left in flight across it --
                    // tried -- sit in the LDS queue in front of the chain waves' fragment reads of the next step: +4 us)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                };
                static_assert(PFR % 4 == 0, "the sweep starts at a step that is 0 modulo 4");
                for (int d = dfirst; d <= ND + 1; d += 4) {
                    step(std::integral_constant<int, 0>{}, d);
                    if (d + 1 > ND + 1) break;
                    step(std::integral_constant<int, 1>{}, d + 1);
                    if (d + 2 > ND + 1) break;
                    step(std::integral_constant<int, 2>{}, d + 2);
                    if (d + 3 > ND + 1) break;
                    step(std::integral_constant<int, 3>{}, d + 3);
                }
                // (the last rows' stores and the last mailbox operations are still on their way.  A redo pass reuses the
                // row registers and waits for them; the launched tile's sweep lets the wave run on to the verdict)
                if (pass > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // ================================ between the sweeps ========================================================
            if (pass > 0) {
                __syncthreads();
                continue;
            }
#ifdef IFL_STAMPS
            if (g_stamps && b == 0 && tid == NW * 64) {
                unsigned long long *o = g_stamps + (my_part == 1 ? 32 : 0);
                o[8] = st_slow, o[9] = st_spins, o[10] = st_gate;
                for (int k = 0; k < 7; ++k) o[16 + k] = st_h[k];
            }
#endif
            int bad = __syncthreads_or(dead << 1);
            if (my_part == 0) {
                // The verdict tells the lower part whether this tile is good.  If not, this L2's dirty lines of z go back
                // first (agent-scope release), so that they cannot land on top of the redone rows later (the two workgroups
                // may sit on XCDs with separate L2s).
                if (!bad) reduce_amax();
                if (tid == NW * 64) {
                    if (bad) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
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
            if (IFL_EXP) bad = 0;
            if (!bad) {
                reduce_amax();
                if (tid == NW * 64) flags[b] = 0;
                advance_generation();
                return;
            }
            // redo: scaled sweeps of both tiles through the mailbox, under the second tag
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            zmax = 0.f;
            dead = 0;
        }
        const int bad2 = __syncthreads_or(dead);
        if (tid == NW * 64) flags[b] = 1 + (bad2 ? 4 : 0);
        if (bad2) {
            scan_general_body<Cfg::THREADS>(xin, wf32, zout, geom, rh, rw, 1, (float *)lds, b, tid);
            __syncthreads();
            if (amax) { // the rows stored above are void: take the maximum of what the redo wrote
                const float *zi = zout + (size_t)b * Cr * H * W;
                zmax = 0.f;
                for (int i = tid - NW * 64; i < Cr * H * W; i += NW * 64) zmax = fmaxf(zmax, fabsf(zi[i]));
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

template <int C, int KH, int KW, bool PAD>
static int launch_duo(const float *x, float *z, const void *apack, const Geom &g, int rh, int rw, int *flags,
                      const float *wf32, unsigned *amax, char *state, hipStream_t s)
{
    using Cfg = DuoCfg<C, KH, KW>;
    static_assert(Cfg::LDSB <= 160 * 1024, "ring + staging must fit the CU's LDS");
    static LdsOptIn opt_in;
    if (int rc = lds_opt_in(opt_in, (const void *)k_scan_duo<C, KH, KW, PAD>, Cfg::LDSB)) return rc;
    if (scan_general_lds_bytes(g) > (size_t)Cfg::LDSB)
        IFL_FAIL(IFL_EUNSUPPORTED, "launch_scan_duo: fp32 fallback does not fit the kernel's LDS");
    const int nparts = g.H > 16 ? 2 : 1;
    if (nparts == 2 && (!state || g.B > DUO_MAX_IMAGES)) IFL_FAIL(IFL_EINVAL, "launch_scan_duo: a two-part scan needs the state block");
    SplitState sp{nullptr, nullptr};
    if (state) sp = SplitState{(unsigned long long *)(state + DUO_MBOX_OFF), (unsigned *)state};
#ifdef IFL_STAMPS
    if (const char *e = getenv("IFL_STAMPS")) {
        unsigned long long *ptr = (unsigned long long *)strtoull(e, nullptr, 0);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &ptr, sizeof(ptr));
    }
#endif
    const dim3 grid(nparts == 2 ? 16 * ((g.B + 7) / 8) : g.B);
    hipLaunchKernelGGL((k_scan_duo<C, KH, KW, PAD>), grid, dim3(Cfg::THREADS), Cfg::LDSB, s, x, z, (const half8 *)apack, g.H,
                       g.W, rh, rw, flags, wf32, g, amax, sp, nparts);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

int launch_scan_duo(const float *x, const void *apack, float *z, const Geom &g, int rh, int rw, int *flags,
                    const float *wf32, unsigned *amax, void *state, hipStream_t s)
{
#define IFL_CASE(CC, KK)                                                                                               \
    if (g.C == CC && g.KH == KK && g.KW == KK)                                                                         \
        return launch_duo<CC, KK, KK, false>(x, z, apack, g, rh, rw, flags, wf32, amax, (char *)state, s);
    IFL_CASE(64, 3)
    IFL_CASE(32, 3)
    IFL_CASE(64, 2)
    IFL_CASE(32, 2)
#undef IFL_CASE
    IFL_FAIL(IFL_EUNSUPPORTED, "launch_scan_duo: no instantiation for C=%d K=%dx%d", g.C, g.KH, g.KW);
}

} // namespace ifl
