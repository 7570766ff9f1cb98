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
// LDS (~151 KB at C=64, 32 rows):
//   ring   last KH+KW-1 diagonals of r as split fp16 (hi, lo*2^11), [slot][row block][plane][row%16]
//          16-byte pieces, plane = (k-step, hi/lo, k-group): a lane's MFMA B fragment is one
//          ds_read_b128 and the 16 lanes of every hardware lane group hit 16 different 4-bank
//          columns (conflict-free; the naive [row][channel] layout measured 52 % conflict cycles).
//          Sources above the image read a zero block, pixels left of it are never written and stay
//          zero: exactly the TL zero padding.
//   xs     x quads [row][quad parity][channel], filled by LDS-DMA (global_load_lds_dwordx4: no VGPR
//          destination, nothing for the compiler to track) three steps before their first use.
//   zring  last 5 diagonals of z (fp32) [slot][row][channel]; a row's quad is stored to NCHW with one
//          global_store_dwordx4 per channel once its four diagonals are in.
// Both global streams move 16-byte aligned quads of one (channel, row) line; lane = channel.
//
// The only vector-memory operations of a wave are its own DMAs and stores, whose numbers per step are
// known exactly, so the wait before each barrier is an exact s_waitcnt vmcnt(n): "the DMA issued three
// steps ago has landed", never "everything, including the stores I just issued".
//
// Arithmetic: split-fp16 MFMA with fp32 accumulation.  a*b ~= ah*bh + (ah*bl' + al'*bh) 2^-11 with
// ah = fp16(a), al' = fp16((a-ah) 2^11): three v_mfma_f32_16x16x32_f16 per 32-deep k-step, the
// dropped al*bl term is 2^-22 relative; fp16 denormal operands are kept by the MFMA (checked on
// gfx950 with tools/mfma_f16_denorm_probe.hip).  The folded weights live in AGPRs for the whole scan.
#include <type_traits>

#include "ifl_common.h"
#include "scan_general_body.h"

namespace ifl {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

static constexpr float LO_SCALE = 2048.0f;
static constexpr float LO_INV = 1.0f / 2048.0f;

template <int C, int KH, int KW, int NTILE> struct ScanCfg {
    static constexpr int NW = C / 16;      // 16-channel output groups
    static constexpr int NWAVES = NW * NTILE; // one wave per (row tile, channel group): two waves per SIMD at 32 rows
    static constexpr int NQ = C / 32;      // 32-deep k-steps per tap
    static constexpr int NT = KH * KW;     // taps incl. the diagonal one
    static constexpr int NS = NT;          // A slots: NT-1 folded taps + 1 post matrix (L^-1)
    static constexpr int R = KH + KW - 1;  // r-ring depth (current + KH+KW-2 previous diagonals)
    static constexpr int NPL = NQ * 8;     // planes per row block: (k-step, hi/lo, k-group)
    static constexpr int RBB = NPL * 256;  // bytes of one row block (16 rows x NPL planes x 16 B)
    static constexpr int SLOTB = NTILE * RBB;
    static constexpr int RINGB = R * SLOTB;
    static constexpr int ZEROB = RBB;      // always-zero block read for sources above the image
    static constexpr int XROWB = 2 * C * 16 + 16; // x quads of one row: [parity][channel] + pad
    static constexpr int XSB = 16 * NTILE * XROWB;
    static constexpr int RZ = 5;           // z-ring depth: a quad's 4 diagonals + the one being written
    static constexpr int ZROWB = C * 4 + 16;
    static constexpr int ZSLOTB = 16 * NTILE * ZROWB;
    static constexpr int ZRINGB = RZ * ZSLOTB;
    static constexpr int OFF_ZERO = RINGB, OFF_XS = OFF_ZERO + ZEROB, OFF_ZR = OFF_XS + XSB;
    static constexpr int LDSB = OFF_ZR + ZRINGB;
    static constexpr int THREADS = 64 * NWAVES;
    static constexpr int ROWS_PER_ITER = 4 * NTILE; // rows that start/finish a quad each step
    static constexpr int G = ROWS_PER_ITER / NWAVES; // ... per wave: DMA (and at most as many store) instructions
    static_assert(ROWS_PER_ITER % NWAVES == 0, "rows per step must split evenly over the waves");
    static_assert(C <= 64, "one DMA / store instruction covers one image row of all channels (lane = channel)");
    static_assert(KH <= 16, "a source row is at most one row block up");
};

// LDS fragment read / counted wait as inline asm: hipcc's own waitcnt insertion answers a block of
// outstanding ds_reads with lgkmcnt(0) (measured: every prefetched fragment waited for the youngest one),
// so the fragment pipeline is counted by hand.  LDS operations of a wave complete in order; the wait
// statement redefines the fragments it guards, which keeps their MFMAs behind it.
__device__ __forceinline__ void lds_read_b128(half8 &v, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
}
template <int N> __device__ __forceinline__ void lgkm_wait(half8 &a, half8 &b)
{
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

// "all but the n youngest vector-memory operations of this wave are complete", LDS drained, then the
// workgroup barrier.  n is exact (see the step body) and wave-uniform.
__device__ __forceinline__ void wait_vm_then_barrier(int n)
{
#define IFL_W(N) \
    case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    switch (__builtin_amdgcn_readfirstlane(n)) {
        IFL_W(0) IFL_W(1) IFL_W(2) IFL_W(3) IFL_W(4) IFL_W(5) IFL_W(6) IFL_W(7) IFL_W(8) IFL_W(9) IFL_W(10) IFL_W(11)
        IFL_W(12) IFL_W(13) IFL_W(14) IFL_W(15) IFL_W(16) IFL_W(17) IFL_W(18) IFL_W(19) IFL_W(20) IFL_W(21) IFL_W(22)
        IFL_W(23) IFL_W(24)
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
#undef IFL_W
}

template <int C, int KH, int KW, int NTILE>
__global__ __launch_bounds__(64 * (C / 16) * NTILE) void k_scan_mfma(const float *__restrict__ xin,
                                                                    float *__restrict__ zout,
                                                                    const half8 *__restrict__ apack, int H, int W,
                                                                    int rh, int rw, int *__restrict__ flags,
                                                                    const float *__restrict__ wf32, Geom geom,
                                                                    unsigned *__restrict__ amax)
{
    using Cfg = ScanCfg<C, KH, KW, NTILE>;
    constexpr int NQ = Cfg::NQ, NT = Cfg::NT, NS = Cfg::NS, R = Cfg::R, RBB = Cfg::RBB, SLOTB = Cfg::SLOTB,
                  RZ = Cfg::RZ, G = Cfg::G, NW = Cfg::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *ring = lds;
    unsigned char *xs = lds + Cfg::OFF_XS;
    unsigned char *zr = lds + Cfg::OFF_ZR;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index: scalar
    const int wv = wave % NW; // 16-channel output group of this wave
    const int T = wave / NW;  // 16-row tile of this wave
    const int n = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const int ND = H + W - 1;

    // ---- zero the r-ring and the zero block (zero padding of the operator) ---------------------------
    {
        const floatx4 zz = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid * 16; i < Cfg::OFF_XS; i += Cfg::THREADS * 16) *(floatx4 *)(lds + i) = zz;
    }

    // ---- folded weights -> registers (A fragments, hi and lo) -----------------------------------
    half8 A[NS][NQ][2];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
                A[s][q][hl] = apack[((((size_t)wv * NS + s) * NQ + q) * 2 + hl) * 64 + lane];
                // pin the fragment in the accumulator half of the register file (MFMA reads A from AGPRs
                // directly); without this hipcc re-loads the weights from memory inside the scan loop
                asm volatile("" : "+a"(A[s][q][hl]));
            }

    // ---- per-lane constants -------------------------------------------------------------------
    // compute role: lane (n, g) owns pixel row h = 16T+n and channels c0..c0+3 (C/D layout)
    const int c0 = 16 * wv + 4 * g;
    const int h = 16 * T + n;
    const bool hval = h < H;
    int radr[KH]; // slot-relative LDS offset of this lane's B piece for a source dh rows up (plane 0 of its g)
    bool rzero[KH]; // the source row is above the image
#pragma unroll
    for (int dh = 0; dh < KH; ++dh) {
        const int hs = h - dh;
        rzero[dh] = hs < 0;
        radr[dh] = hs >= 0 ? (hs / 16) * RBB + g * 256 + (hs % 16) * 16 : 0;
    }
    const int wadr = T * RBB + (((c0 / 32) * 2) * 4 + (c0 % 32) / 8) * 256 + n * 16 + ((c0 % 8) / 4) * 8;
    const int xadr = h * Cfg::XROWB + c0 * 16;
    const int zadr = h * Cfg::ZROWB + c0 * 4;
    const int zbase = Cfg::OFF_ZERO + g * 256 + n * 16; // this lane's piece in the zero block
    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;

    // DMA / store role: lane = channel, G rows per wave and step
    const int cl = lane < C ? lane : 0;
    const unsigned voff = (unsigned)((size_t)cl * H * W * sizeof(float)); // per-lane byte offset of its channel
    const char *xg = (const char *)xin + (size_t)b * C * H * W * sizeof(float);
    char *zg = (char *)zout + (size_t)b * C * H * W * sizeof(float);

    // soff[k] / zoff[k]: LDS offsets of the slots holding diagonal d-k (r) and d-1-k (z)
    int soff[R], zoff[RZ];
#pragma unroll
    for (int k = 0; k < R; ++k) soff[k] = ((R - k) % R) * SLOTB;
#pragma unroll
    for (int k = 0; k < RZ; ++k) zoff[k] = ((RZ - k) % RZ) * Cfg::ZSLOTB;
    int nst[3] = {0, 0, 0}; // store instructions this wave issued in the previous three steps
    bool ovf = false;       // an r of this lane left the fp16 range (or is not a number)
    float zmax = 0.f;       // max |z| this lane stored (handed to the weight-gradient kernel as its prescale)

    __syncthreads();

    // One step of the scan, specialised on whether this wave's tile holds pixels of diagonal d, so that
    // the body is straight-line code; LDS fragment reads run one (tap, k-step) unit ahead of the MFMAs
    // that consume them.  With 32 image rows two waves share a SIMD (one per row tile): one wave's LDS
    // latency is covered by the other's MFMAs.
    auto step = [&](auto act_c, const int d) {
        constexpr bool ACT = decltype(act_c)::value;
        const int ph = (((d - 1) % 4) + 4) % 4; // rows h = ph (mod 4) are one step into a quad of x

        // ---- x quads needed three steps from now: rows at w = d-h = 1 (mod 4) fetch their next quad.
        //      The slot-mate (two quads back) was last read in phase C of step d-2, which every wave
        //      left before the barrier of step d-1.  Issued unconditionally (out-of-range rows load a
        //      valid dummy) so that the number of VM operations per step is exact.
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int hr = ph + 4 * (wave * G + i);
            const int wq = d - hr + 3;
            const bool ok = hr < H && wq >= 0 && wq < W;
            const int hc = ok ? hr : 0, wc = ok ? wq : 0;
            const int hs = rh ? H - 1 - hc : hc;
            const int ws = rw ? W - 4 - wc : wc;
            const char *src = xg + ((size_t)hs * W + ws) * sizeof(float); // wave-uniform
            unsigned char *dst = xs + hr * Cfg::XROWB + ((wq >> 2) & 1) * (C * 16); // lane c lands at +16c
            if (C == 64 || lane < C)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + voff),
                                                 (void __attribute__((address_space(3))) *)dst, 16, 0, 0);
        }

        // two accumulator chains: hi*hi and the 2^-11-scaled cross terms hi*lo' + lo'*hi (a third chain for
        // the second cross term measured slower in the full kernel although faster in isolation)
        floatx4 ahi = {0.f, 0.f, 0.f, 0.f}, amid = {0.f, 0.f, 0.f, 0.f};

        // (tap, k-step) units of a phase.  SEL 0: taps two or more diagonals back (no dependence on step
        // d-1), SEL 1: the two taps on diagonal d-1.
        auto unit_count = [](int sel) {
            int c = 0;
            for (int t = 1; t < NT; ++t) {
                const int s2 = t / KW + t % KW;
                if (sel ? s2 == 1 : s2 >= 2) c += NQ;
            }
            return c;
        };
        auto unit_tap = [](int sel, int u) {
            int c = 0;
            for (int t = 1; t < NT; ++t) {
                const int s2 = t / KW + t % KW;
                if (sel ? s2 == 1 : s2 >= 2) {
                    if (u < c + NQ) return t;
                    c += NQ;
                }
            }
            return 1;
        };
        auto unit_q = [](int sel, int u) {
            int c = 0;
            for (int t = 1; t < NT; ++t) {
                const int s2 = t / KW + t % KW;
                if (sel ? s2 == 1 : s2 >= 2) {
                    if (u < c + NQ) return u - c;
                    c += NQ;
                }
            }
            return 0;
        };
        auto load_frag = [&](int sel, int u, half8 &vh, half8 &vl) {
            const int t = unit_tap(sel, u), q = unit_q(sel, u), dh = t / KW, dw = t % KW;
            int a = soff[dh + dw] + radr[dh];
            if (dh > 0) a = rzero[dh] ? zbase : a;
            lds_read_b128(vh, ldsbase + a + (q * 2) * 4 * 256);
            lds_read_b128(vl, ldsbase + a + (q * 2 + 1) * 4 * 256);
        };
        auto mfma3 = [&](int sel, int u, const half8 &vh, const half8 &vl) {
            const int t = unit_tap(sel, u), q = unit_q(sel, u);
            ahi = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][0], vh, ahi, 0, 0, 0);
            amid = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][0], vl, amid, 0, 0, 0);
            amid = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][1], vh, amid, 0, 0, 0);
        };

        // ---- (B) old taps: a PF-deep register ring of fragments keeps the LDS latency (a few hundred
        //      cycles with eight waves reading) off the MFMA stream -------------------------------------
        constexpr int NUB = unit_count(0), NUC = unit_count(1);
        constexpr int PF = NUB < 6 ? (NUB > 0 ? NUB : 1) : 6;
        if constexpr (ACT && NUB > 0) {
            half8 fh[PF], fl[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) load_frag(0, u, fh[u], fl[u]);
            // hipcc's scheduler otherwise sinks every read down to its first use (no prefetch at all):
            // pin the source order of reads and MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NUB; ++u) {
                // reads issued so far: units < min(NUB, u+PF); unit u is the oldest outstanding one
                {
                    const int issued = (u + PF < NUB ? u + PF : NUB);
                    const int allow = 2 * (issued - u - 1);
                    switch (allow) {
                    case 0: lgkm_wait<0>(fh[u % PF], fl[u % PF]); break;
                    case 2: lgkm_wait<2>(fh[u % PF], fl[u % PF]); break;
                    case 4: lgkm_wait<4>(fh[u % PF], fl[u % PF]); break;
                    case 6: lgkm_wait<6>(fh[u % PF], fl[u % PF]); break;
                    case 8: lgkm_wait<8>(fh[u % PF], fl[u % PF]); break;
                    default: lgkm_wait<10>(fh[u % PF], fl[u % PF]); break;
                    }
                }
                mfma3(0, u, fh[u % PF], fl[u % PF]);
                __builtin_amdgcn_sched_barrier(0);
                if (u + PF < NUB) {
                    load_frag(0, u + PF, fh[u % PF], fl[u % PF]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        // The quads this wave DMA'd three steps ago must have landed before anyone reads them in phase C:
        // younger than those are exactly the stores of steps d-3..d-1 and the DMAs of steps d-2..d.
        // Then the barrier: r of diagonal d-1 (and z of diagonal d-2) are complete in LDS.
        wait_vm_then_barrier(3 * G + nst[0] + nst[1] + nst[2]);

        // ---- (C) the two taps on diagonal d-1: all their fragments are requested at once, the MFMAs follow
        //      with exact counted waits ---------------------------------------------------------------------
        const int w = d - h;
        if constexpr (ACT && NUC > 0) {
            half8 ch[NUC], cl2[NUC];
#pragma unroll
            for (int u = 0; u < NUC; ++u) load_frag(1, u, ch[u], cl2[u]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NUC; ++u) {
                switch (2 * (NUC - 1 - u)) {
                case 0: lgkm_wait<0>(ch[u], cl2[u]); break;
                case 2: lgkm_wait<2>(ch[u], cl2[u]); break;
                case 4: lgkm_wait<4>(ch[u], cl2[u]); break;
                case 6: lgkm_wait<6>(ch[u], cl2[u]); break;
                case 8: lgkm_wait<8>(ch[u], cl2[u]); break;
                case 10: lgkm_wait<10>(ch[u], cl2[u]); break;
                default: lgkm_wait<0>(ch[u], cl2[u]); break;
                }
                mfma3(1, u, ch[u], cl2[u]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- requests whose latency hides behind the MFMAs just issued: x of this step, the fragments of
        //      the z product, the finished z quad(s) --------------------------------------------------------
        float xv[4];
        if constexpr (ACT) {
            // this lane's 4 x values: quad (w>>2) of row h, channels c0..c0+3, element w&3
            const unsigned char *xp = xs + xadr + ((w >> 2) & 1) * (C * 16) + (rw ? 3 - (w & 3) : (w & 3)) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) xv[r] = *(const float *)(xp + r * 16);
        }
        const bool actz = d - 1 >= 16 * T && d - 1 <= 16 * T + 15 + W - 1 && d >= 1 && d <= ND && 16 * T < H;
        half8 bh[NQ], bl[NQ];
        if (actz) {
            const unsigned char *pp = lds + soff[1] + radr[0];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                bh[q] = *(const half8 *)(pp + (q * 2) * 4 * 256);
                bl[q] = *(const half8 *)(pp + (q * 2 + 1) * 4 * 256);
            }
        }
        // store role: rows h = ph (mod 4) completed a quad of z with diagonal d-2; its four elements sit
        // in the z-slots of diagonals d-5..d-2 (zoff[k] holds diagonal d-1-k)
        floatx4 sv[G];
        bool sok[G];
        int nstore = 0;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int hr = ph + 4 * (wave * G + i);
            const int w3 = d - 2 - hr; // last column of the quad (w3 = 3 mod 4)
            sok[i] = hr < H && w3 >= 3 && w3 < W; // wave-uniform
            if (sok[i]) {
                nstore += 1;
                const unsigned char *zp = zr + hr * Cfg::ZROWB + cl * 4;
                sv[i][rw ? 3 : 0] = *(const float *)(zp + zoff[4]);
                sv[i][rw ? 2 : 1] = *(const float *)(zp + zoff[3]);
                sv[i][rw ? 1 : 2] = *(const float *)(zp + zoff[2]);
                sv[i][rw ? 0 : 3] = *(const float *)(zp + zoff[1]);
            }
        }

        // ---- (D) z of diagonal d-1 = L^-1 r: independent MFMAs that keep the matrix pipe busy while r_d is
        //      converted and written ---------------------------------------------------------------------
        floatx4 zh = {0.f, 0.f, 0.f, 0.f}, zm = {0.f, 0.f, 0.f, 0.f};
        if (actz) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                zh = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[NS - 1][q][0], bh[q], zh, 0, 0, 0);
                zm = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[NS - 1][q][0], bl[q], zm, 0, 0, 0);
                zm = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[NS - 1][q][1], bh[q], zm, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0); // MFMAs are in the pipe before the VALU-heavy tail starts
#pragma unroll
        for (int i = 0; i < G; ++i)
            if (sok[i]) {
                const int hr = ph + 4 * (wave * G + i);
                const int w3 = d - 2 - hr;
                const int hs = rh ? H - 1 - hr : hr;
                const int ws = rw ? W - 1 - w3 : w3 - 3;
                char *dstp = zg + ((size_t)hs * W + ws) * sizeof(float); // wave-uniform
                if (C == 64 || lane < C) {
                    *(floatx4 *)(dstp + voff) = sv[i];
                    zmax = fmaxf(zmax, fmaxf(fmaxf(fabsf(sv[i][0]), fabsf(sv[i][1])), fmaxf(fabsf(sv[i][2]), fabsf(sv[i][3]))));
                }
            }

        // ---- epilogue: r_d -> split fp16 -> ring; z_{d-1} -> z-ring ----------------------------------------
        if constexpr (ACT) {
            half4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rv = xv[r] + ahi[r] + amid[r] * LO_INV;
                ovf |= (hval && w >= 0 && w < W) && !(fabsf(rv) < 6.0e4f);
                const _Float16 h16 = (_Float16)rv;
                hi[r] = h16;
                lo[r] = (_Float16)((rv - (float)h16) * LO_SCALE);
            }
            if (hval && w >= 0 && w < W) {
                *(half4 *)(ring + soff[0] + wadr) = hi;
                *(half4 *)(ring + soff[0] + wadr + 4 * 256) = lo;
            }
        }
        if (actz) {
            floatx4 zv;
#pragma unroll
            for (int r = 0; r < 4; ++r) zv[r] = zh[r] + zm[r] * LO_INV;
            *(floatx4 *)(zr + zoff[0] + zadr) = zv; // lanes outside the image write values nobody stores
        }

        // rotate the slot tables: diagonal d+1 takes the slot of diagonal d-(R-1), likewise for z
        {
            const int last = soff[R - 1];
#pragma unroll
            for (int k = R - 1; k > 0; --k) soff[k] = soff[k - 1];
            soff[0] = last;
            const int zlast = zoff[RZ - 1];
#pragma unroll
            for (int k = RZ - 1; k > 0; --k) zoff[k] = zoff[k - 1];
            zoff[0] = zlast;
        }
        nst[2] = nst[1];
        nst[1] = nst[0];
        nst[0] = nstore;
    };

    for (int d = -3; d <= ND + 1; ++d) {
        // wave-uniform: does this wave's tile hold any pixel of diagonal d?
        const bool act = d >= 16 * T && d <= 16 * T + 15 + W - 1 && d < ND && 16 * T < H;
        if (act)
            step(std::true_type{}, d);
        else
            step(std::false_type{}, d);
    }
    // Split fp16 cannot hold |r| >= 65504 (a badly conditioned operator grows r along the sweep): such an
    // image is redone here, by the same workgroup, in exact fp32 from the fp32 copy of the same folded
    // weights (general scan body, right-fold form).  Rare, slow, but never a silent Inf/NaN where the exact
    // solver is finite.  flags[] records it for the caller (diagnostics only).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // no LDS-DMA may still be landing in the LDS reused below
    const int redo = __syncthreads_or(ovf ? 1 : 0);
    if (tid == 0) flags[b] = redo ? 1 : 0; // every workgroup owns its word: no clearing pass needed
    if (redo) {
        scan_general_body<Cfg::THREADS>(xin, wf32, zout, geom, rh, rw, 1, (float *)lds, b, tid);
        if (amax) { // the quads stored above are void: take the maximum of what the redo wrote
            __syncthreads();
            const float *zi = zout + (size_t)b * C * H * W;
            zmax = 0.f;
            for (int i = tid; i < C * H * W; i += Cfg::THREADS) zmax = fmaxf(zmax, fabsf(zi[i]));
        }
    }
    if (amax) {
        for (int o = 32; o > 0; o >>= 1) zmax = fmaxf(zmax, __shfl_down(zmax, o, 64));
        if (lane == 0) atomicMax(amax, __float_as_uint(zmax)); // one atomic per wave; max is order-independent
    }
}

// ------------------------------------------------------------------------------------------------
// Fused fold + pack for the MFMA scan: one launch builds the A fragments in the exact per-lane
// register image.
//   slot s < NT-1 : -(W_t L^-1)          (t = s+1; transposed: -(W_t^T L^-T))
//   slot NT-1     :  L^-1                (transposed: L^-T)
// apack[wv][s][q][hl][lane][j] (fp16),  lane = m + 16*gk holds row c = 16wv+m, k = 32q + 8gk + j.
//
// Grid = NT slots x C/16 row groups.  Every workgroup first inverts L in LDS (fp64) by 16x16
// blocks -- diagonal blocks by substitution (16 dependent steps), off-diagonal blocks
// X_ij = -X_ii (sum_k L_ik X_kj) level by level -- which cuts the 64-step serial substitution of
// the exact solver's channel loop (solve_mc.py:96-109) to ~16 + 2(C/16-1) short dependent stages,
// then forms its 16 rows of the product and splits them into fp16 hi / lo*2^11.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t w_index2(int co, int ci, int dh, int dw, int C, int KH, int KW, int flipH,
                                           int flipW)
{
    int kh = KH - 1 - dh, kw = KW - 1 - dw;
    if (flipH) kh = KH - 1 - kh;
    if (flipW) kw = KW - 1 - kw;
    return (((size_t)co * C + ci) * KH + kh) * KW + kw;
}

template <int C>
__global__ __launch_bounds__(256) void k_foldpack(const float *__restrict__ w, _Float16 *__restrict__ apack0,
                                                  float *__restrict__ wf0, _Float16 *__restrict__ apack1,
                                                  float *__restrict__ wf1, Geom g, int transposed0,
                                                  unsigned *__restrict__ zero0, unsigned *__restrict__ zero1)
{
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
    constexpr int NBK = C / 16, XP = C + 1; // XP: fp64 row pitch (conflict-free row- and column-wise)
    __shared__ float sL[C * C];
    __shared__ double sX[C * XP];
    __shared__ double sS[(NBK > 1 ? NBK - 1 : 1) * 256];
    __shared__ float sW[16 * C];
    const int tid = threadIdx.x;
    const int NT = g.KH * g.KW, RG = C / 16, NQ = C / 32;
    const int s = blockIdx.x / RG, rgrp = blockIdx.x % RG;

    // ---- L (effective diagonal tap) and this workgroup's 16 rows of W_t ----------------------------
    for (int idx = tid; idx < C * C; idx += 256) {
        const int i = idx / C, k = idx % C;
        float v = 0.f;
        if (k < i) v = w[w_index2(i, k, 0, 0, C, g.KH, g.KW, g.flipH, g.flipW)];
        else if (k == i) v = g.general_diag ? w[w_index2(i, i, 0, 0, C, g.KH, g.KW, g.flipH, g.flipW)] : 1.f;
        sL[idx] = v;
    }
    if (s < NT - 1) {
        const int t = s + 1, dh = t / g.KW, dw = t % g.KW;
        for (int idx = tid; idx < 16 * C; idx += 256) {
            const int cl = idx / C, m = idx % C, c = 16 * rgrp + cl;
            sW[idx] = transposed ? w[w_index2(m, c, dh, dw, C, g.KH, g.KW, g.flipH, g.flipW)]
                                 : w[w_index2(c, m, dh, dw, C, g.KH, g.KW, g.flipH, g.flipW)];
        }
    }
    for (int idx = tid; idx < C * XP; idx += 256) sX[idx] = 0.0;
    __syncthreads();

    // ---- diagonal blocks of L^-1: column j by forward substitution inside its 16x16 block; the column
    //      lives in registers (fully unrolled), L comes from LDS and does not depend on the chain ------
    if (tid < C) {
        const int j = tid, r0 = (j / 16) * 16, jj = j % 16;
        double col[16];
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) {
            double a0 = (ii == jj) ? 1.0 : 0.0, a1 = 0.0;
#pragma unroll
            for (int kk = 0; kk < 16; kk += 2) {
                if (kk < ii) a0 -= (double)sL[(r0 + ii) * C + r0 + kk] * (kk >= jj ? col[kk] : 0.0);
                if (kk + 1 < ii) a1 -= (double)sL[(r0 + ii) * C + r0 + kk + 1] * (kk + 1 >= jj ? col[kk + 1] : 0.0);
            }
            col[ii] = ii >= jj ? (a0 + a1) / (double)sL[(r0 + ii) * C + r0 + ii] : 0.0;
            sX[(r0 + ii) * XP + j] = col[ii];
        }
    }
    __syncthreads();

    // ---- off-diagonal blocks, one block-distance at a time ----------------------------------------
    {
        const int r = tid / 16, cc = tid % 16;
#pragma unroll 1
        for (int dist = 1; dist < NBK; ++dist) {
            const int npairs = NBK - dist;
            for (int p = 0; p < npairs; ++p) {
                const int bi = p + dist, bj = p;
                const int row = 16 * bi + r, col = 16 * bj + cc;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                for (int kb = bj; kb < bi; ++kb) {
#pragma unroll
                    for (int kk = 0; kk < 16; kk += 4) {
                        const int k = 16 * kb + kk;
                        a0 += (double)sL[row * C + k] * sX[k * XP + col];
                        a1 += (double)sL[row * C + k + 1] * sX[(k + 1) * XP + col];
                        a2 += (double)sL[row * C + k + 2] * sX[(k + 2) * XP + col];
                        a3 += (double)sL[row * C + k + 3] * sX[(k + 3) * XP + col];
                    }
                }
                sS[p * 256 + tid] = (a0 + a1) + (a2 + a3);
            }
            __syncthreads();
            for (int p = 0; p < npairs; ++p) {
                const int bi = p + dist, bj = p;
                const int row = 16 * bi + r, col = 16 * bj + cc;
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int kk = 0; kk < 16; kk += 2) {
                    // X_ii is lower triangular: entries with kk > r are exactly zero in sX
                    a0 += sX[row * XP + 16 * bi + kk] * sS[p * 256 + kk * 16 + cc];
                    a1 += sX[row * XP + 16 * bi + kk + 1] * sS[p * 256 + (kk + 1) * 16 + cc];
                }
                sX[row * XP + col] = -(a0 + a1);
            }
            __syncthreads();
        }
    }

    // ---- this workgroup's 16 rows of the slot: product, split, pack ------------------------------------
    for (int idx = tid; idx < 16 * C; idx += 256) {
        const int cl = idx / C, kc = idx % C, c = 16 * rgrp + cl;
        double acc = 0.0;
        if (s == NT - 1) {
            acc = transposed ? sX[kc * XP + c] : sX[c * XP + kc];
        } else {
            // normal:     (W_t L^-1)[c][kc]     = sum_m W_t[c][m] Linv[m][kc]   (Linv[m][kc] = 0 for m < kc)
            // transposed: (W_t^T L^-T)[c][kc]   = sum_m W_t[m][c] Linv[kc][m]   (Linv[kc][m] = 0 for m > kc)
            // the zeros are stored, so the sums run over all m: fixed trip count, unrolled, 4 chains
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            const int xs_m = transposed ? 1 : XP, xs_0 = transposed ? kc * XP : kc;
#pragma unroll 4
            for (int m = 0; m < C; m += 4) {
                a0 += (double)sW[cl * C + m] * sX[xs_0 + m * xs_m];
                a1 += (double)sW[cl * C + m + 1] * sX[xs_0 + (m + 1) * xs_m];
                a2 += (double)sW[cl * C + m + 2] * sX[xs_0 + (m + 2) * xs_m];
                a3 += (double)sW[cl * C + m + 3] * sX[xs_0 + (m + 3) * xs_m];
            }
            acc = -((a0 + a1) + (a2 + a3));
        }
        const float v = (float)acc;
        if (wf32) wf32[((size_t)s * C + kc) * C + c] = v; // fp32 copy [slot][kc][c] for the fp32 fallback scan
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)((v - (float)hi) * LO_SCALE);
        const int q = kc / 32, gk = (kc % 32) / 8, j = kc % 8;
        const size_t base = ((((size_t)rgrp * NT + s) * NQ + q) * 2) * 64 * 8;
        apack[base + (size_t)(cl + 16 * gk) * 8 + j] = hi;
        apack[base + (size_t)64 * 8 + (size_t)(cl + 16 * gk) * 8 + j] = lo;
    }
}

size_t scan_mfma_pack_bytes(const Geom &g) { return (size_t)g.KH * g.KW * g.C * g.C * 2 * sizeof(_Float16); }

bool scan_mfma_supported(const Geom &g, const void *x, const void *z)
{
    if (!(g.C == 32 || g.C == 64)) return false;
    if (!((g.KH == 3 && g.KW == 3) || (g.KH == 2 && g.KW == 2))) return false;
    if (g.W % 4 != 0 || g.H > 32 || g.H < 1) return false;
    if (((uintptr_t)x | (uintptr_t)z) & 15) return false;
    return true;
}

int launch_foldpack_mfma(const float *w, void *out0, float *wf0, void *out1, float *wf1, const Geom &g, int transposed,
                         int ndir, unsigned *zero0, unsigned *zero1, hipStream_t s)
{
    const dim3 grid(g.KH * g.KW * (g.C / 16), ndir);
    if (g.C == 64)
        hipLaunchKernelGGL(k_foldpack<64>, grid, dim3(256), 0, s, w, (_Float16 *)out0, wf0, (_Float16 *)out1, wf1, g,
                           transposed, zero0, zero1);
    else if (g.C == 32)
        hipLaunchKernelGGL(k_foldpack<32>, grid, dim3(256), 0, s, w, (_Float16 *)out0, wf0, (_Float16 *)out1, wf1, g,
                           transposed, zero0, zero1);
    else
        IFL_FAIL(IFL_EUNSUPPORTED, "launch_foldpack_mfma: C=%d", g.C);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

template <int C, int KH, int KW, int NTILE>
static int launch_one(const float *x, float *z, const void *apack, const Geom &g, int rh, int rw, int *flags,
                      const float *wf32, unsigned *amax, hipStream_t s)
{
    using Cfg = ScanCfg<C, KH, KW, NTILE>;
    static_assert(Cfg::LDSB <= 160 * 1024, "ring + x staging must fit the CU's LDS");
    static bool attr_done = false; // idempotent attribute, benign race
    if (!attr_done) {
        IFL_HIP(hipFuncSetAttribute((const void *)k_scan_mfma<C, KH, KW, NTILE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    Cfg::LDSB));
        attr_done = true;
    }
    if (scan_general_lds_bytes(g) > (size_t)Cfg::LDSB)
        IFL_FAIL(IFL_EUNSUPPORTED, "launch_scan_mfma: fp32 fallback does not fit the kernel's LDS");
    hipLaunchKernelGGL((k_scan_mfma<C, KH, KW, NTILE>), dim3(g.B), dim3(Cfg::THREADS), Cfg::LDSB, s, x, z,
                       (const half8 *)apack, g.H, g.W, rh, rw, flags, wf32, g, amax);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

int launch_scan_mfma(const float *x, const void *apack, float *z, const Geom &g, int rh, int rw, int *flags,
                     const float *wf32, unsigned *amax, hipStream_t s)
{
    const int nt = g.H <= 16 ? 1 : 2;
#define IFL_CASE(CC, KK, NN) \
    if (g.C == CC && g.KH == KK && g.KW == KK && nt == NN) return launch_one<CC, KK, KK, NN>(x, z, apack, g, rh, rw, flags, wf32, amax, s);
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
