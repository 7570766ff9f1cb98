// MFMA wavefront back-substitution scan for C in {32, 64} (gfx950, wave64).
//
// Formulation ("right fold"): with L the diagonal-tap matrix and W_t the other taps,
//     r_p = x_p - sum_t (W_t L^-1) r_{p-t},      z_p = L^-1 r_p
// so the sequential chain carries r only; x enters as the plain fp32 accumulator seed and
// z = L^-1 r is a per-pixel product that is *off* the dependency chain.
//
// Mapping.  One workgroup per image, one wave per 16 output channels (C/16 waves).  MFMA tile =
// 16 channels x 16 image rows: MFMA column n of tile T is image row h = 16T+n for the whole
// kernel, and walks along w = d-h as the anti-diagonal index d advances (pixel (h,w) is on
// diagonal h+w; every source (h-dh, w-dw) is on an earlier diagonal: solve_mc.py:88-114 in
// diagonal order, cf. solve_parallel, solve_mc.py:8-50).  Because a lane keeps its row, its x
// values and z results are consecutive in memory over consecutive steps: x is read and z written
// as aligned 16-byte quads straight from/to NCHW, no staging through LDS.
//
// LDS holds only a ring of the last KH+KW-1 diagonals of r as split fp16 (hi, lo*2^11), laid out
// [slot][row][hi C | lo C] so that a lane's MFMA B fragment (8 consecutive channels of one pixel)
// is one ds_read_b128.  Rows above the image and pixels left of it are never written and stay
// zero, which is exactly the TL zero padding.
//
// Arithmetic: split-fp16 MFMA with fp32 accumulation.  a*b ~= ah*bh + (ah*bl' + al'*bh) 2^-11 with
// ah = fp16(a), al' = fp16((a-ah) 2^11): three v_mfma_f32_16x16x32_f16 per 32-deep k-step, the
// dropped al*bl term is 2^-22 relative.  The folded weights live in registers for the whole scan
// (144 VGPRs at C=64, K=3).
#include "ifl_common.h"

namespace ifl {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

static constexpr float LO_SCALE = 2048.0f;
static constexpr float LO_INV = 1.0f / 2048.0f;

__device__ __forceinline__ float sel4(const floatx4 &v, int i)
{
    float r = v[0];
    r = i == 1 ? v[1] : r;
    r = i == 2 ? v[2] : r;
    r = i == 3 ? v[3] : r;
    return r;
}

__device__ __forceinline__ void ins4(floatx4 &v, int i, float x)
{
    v[0] = i == 0 ? x : v[0];
    v[1] = i == 1 ? x : v[1];
    v[2] = i == 2 ? x : v[2];
    v[3] = i == 3 ? x : v[3];
}

template <int C, int KH, int KW, int NTILE> struct ScanCfg {
    static constexpr int NW = C / 16;          // waves = 16-channel output groups
    static constexpr int NQ = C / 32;          // 32-deep k-steps per tap
    static constexpr int NT = KH * KW;         // taps incl. the diagonal one
    static constexpr int NS = NT;              // A slots: NT-1 folded taps + 1 post matrix (L^-1)
    static constexpr int R = KH + KW - 1;      // ring depth (current + KH+KW-2 previous diagonals)
    static constexpr int PADR = KH - 1;        // always-zero rows above the image
    static constexpr int ROWB = 4 * C + 16;    // bytes per ring row: hi C*2 | lo C*2 | 16 pad
    static constexpr int NROW = 16 * NTILE + PADR;
    static constexpr int SLOTB = NROW * ROWB;
    static constexpr int LDSB = R * SLOTB;
    static constexpr int THREADS = 64 * NW;
};

template <int C, int KH, int KW, int NTILE>
__global__ __launch_bounds__(64 * (C / 16)) void k_scan_mfma(const float *__restrict__ xin, float *__restrict__ zout,
                                                             const half8 *__restrict__ apack, int H, int W, int rh,
                                                             int rw)
{
    using Cfg = ScanCfg<C, KH, KW, NTILE>;
    constexpr int NQ = Cfg::NQ, NT = Cfg::NT, NS = Cfg::NS, R = Cfg::R, PADR = Cfg::PADR, ROWB = Cfg::ROWB,
                  SLOTB = Cfg::SLOTB;
    __shared__ __attribute__((aligned(16))) unsigned char ring[Cfg::LDSB];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const int ND = H + W - 1;

    // ---- folded weights -> registers (A fragments, hi and lo) -----------------------------------
    half8 A[NS][NQ][2];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) A[s][q][hl] = apack[((((size_t)wv * NS + s) * NQ + q) * 2 + hl) * 64 + lane];

    // ---- zero the ring (zero padding of the operator) ------------------------------------------------
    {
        const floatx4 zz = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid * 16; i < Cfg::LDSB; i += Cfg::THREADS * 16) *(floatx4 *)(ring + i) = zz;
    }

    // ---- per-lane constants -------------------------------------------------------------------
    const int c0 = 16 * wv + 4 * g; // first of this lane's 4 output channels (C/D layout rows)
    int hrow[NTILE];                // image row of this lane in tile T
    bool hval[NTILE];
    size_t gbase[NTILE];            // element offset of (b, c0, stored row, 0)
    int rbase[NTILE];               // LDS byte offset of (row h, k-group g) inside a slot
    int wbase[NTILE];               // LDS byte offset of this lane's 4 hi halves inside a slot
    floatx4 xc[NTILE][4], xn[NTILE][4], zo[NTILE][4];
#pragma unroll
    for (int T = 0; T < NTILE; ++T) {
        hrow[T] = 16 * T + n;
        hval[T] = hrow[T] < H;
        const int hs = rh ? H - 1 - hrow[T] : hrow[T];
        gbase[T] = (((size_t)b * C + c0) * H + (hval[T] ? hs : 0)) * W;
        rbase[T] = (hrow[T] + PADR) * ROWB + g * 16;
        wbase[T] = (hrow[T] + PADR) * ROWB + c0 * 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xc[T][r] = floatx4{0.f, 0.f, 0.f, 0.f};
            xn[T][r] = floatx4{0.f, 0.f, 0.f, 0.f};
            zo[T][r] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const size_t cstride = (size_t)H * W;

    // soff[k] = LDS byte offset of the slot holding diagonal d-k
    int soff[R];
#pragma unroll
    for (int k = 0; k < R; ++k) soff[k] = ((R - k) % R) * SLOTB;

    __syncthreads();

    for (int d = -4; d <= ND; ++d) {
        floatx4 ahi[NTILE], amid[NTILE];
        bool act[NTILE];
#pragma unroll
        for (int T = 0; T < NTILE; ++T) {
            ahi[T] = floatx4{0.f, 0.f, 0.f, 0.f};
            amid[T] = floatx4{0.f, 0.f, 0.f, 0.f};
            // wave-uniform: does tile T hold any pixel of diagonal d?
            act[T] = d >= 16 * T && d <= 16 * T + 15 + W - 1 && d < ND && 16 * T < H;
        }

        // ---- (A) x quads: enter the next quad / prefetch the one after -------------------------
#pragma unroll
        for (int T = 0; T < NTILE; ++T) {
            const int w = d - hrow[T];
            if (hval[T] && (w & 3) == 0) {
                if (w >= 0 && w < W) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) xc[T][r] = xn[T][r];
                }
                const int wq = w + 4;
                if (wq >= 0 && wq < W) {
                    const int ws = rw ? W - 4 - wq : wq;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        xn[T][r] = *(const floatx4 *)(xin + gbase[T] + r * cstride + ws);
                }
            }
        }

        // ---- (B) taps whose sources are two or more diagonals back: no dependence on step d-1 ----
#pragma unroll
        for (int T = 0; T < NTILE; ++T) {
            if (act[T]) {
#pragma unroll
                for (int t = 1; t < NT; ++t) {
                    const int dh = t / KW, dw = t % KW;
                    if (dh + dw >= 2) {
                        const unsigned char *rowp = ring + soff[dh + dw] + rbase[T] - dh * ROWB;
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            const half8 bh = *(const half8 *)(rowp + q * 64);
                            const half8 bl = *(const half8 *)(rowp + 2 * C + q * 64);
                            ahi[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][0], bh, ahi[T], 0, 0, 0);
                            amid[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][0], bl, amid[T], 0, 0, 0);
                            amid[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][1], bh, amid[T], 0, 0, 0);
                        }
                    }
                }
            }
        }

        __syncthreads(); // r of diagonal d-1 is complete in the ring

        // ---- (C) the two taps on diagonal d-1, then r_d -> ring ---------------------------------------
#pragma unroll
        for (int T = 0; T < NTILE; ++T) {
            if (act[T]) {
#pragma unroll
                for (int t = 1; t < NT; ++t) {
                    const int dh = t / KW, dw = t % KW;
                    if (dh + dw == 1) {
                        const unsigned char *rowp = ring + soff[1] + rbase[T] - dh * ROWB;
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            const half8 bh = *(const half8 *)(rowp + q * 64);
                            const half8 bl = *(const half8 *)(rowp + 2 * C + q * 64);
                            ahi[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][0], bh, ahi[T], 0, 0, 0);
                            amid[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][0], bl, amid[T], 0, 0, 0);
                            amid[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[t - 1][q][1], bh, amid[T], 0, 0, 0);
                        }
                    }
                }
                const int w = d - hrow[T];
                const int ph = rw ? 3 - (w & 3) : (w & 3);
                half4 hi, lo;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float rv = sel4(xc[T][r], ph) + ahi[T][r] + amid[T][r] * LO_INV;
                    const _Float16 h16 = (_Float16)rv;
                    hi[r] = h16;
                    lo[r] = (_Float16)((rv - (float)h16) * LO_SCALE);
                }
                if (hval[T] && w >= 0 && w < W) {
                    *(half4 *)(ring + soff[0] + wbase[T]) = hi;
                    *(half4 *)(ring + soff[0] + wbase[T] + 2 * C) = lo;
                }
            }
        }

        // ---- (D) z of diagonal d-1 = L^-1 r (off the chain), gathered into quads and stored ----
#pragma unroll
        for (int T = 0; T < NTILE; ++T) {
            const bool actz = d - 1 >= 16 * T && d - 1 <= 16 * T + 15 + W - 1 && d >= 1 && 16 * T < H;
            if (actz) {
                floatx4 zh = {0.f, 0.f, 0.f, 0.f}, zm = {0.f, 0.f, 0.f, 0.f};
                const unsigned char *rowp = ring + soff[1] + rbase[T];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const half8 bh = *(const half8 *)(rowp + q * 64);
                    const half8 bl = *(const half8 *)(rowp + 2 * C + q * 64);
                    zh = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[NS - 1][q][0], bh, zh, 0, 0, 0);
                    zm = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[NS - 1][q][0], bl, zm, 0, 0, 0);
                    zm = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[NS - 1][q][1], bh, zm, 0, 0, 0);
                }
                const int wz = d - 1 - hrow[T];
                const int phz = rw ? 3 - (wz & 3) : (wz & 3);
#pragma unroll
                for (int r = 0; r < 4; ++r) ins4(zo[T][r], phz, zh[r] + zm[r] * LO_INV);
                if (hval[T] && wz >= 0 && wz < W && (wz & 3) == 3) {
                    const int ws = rw ? W - 1 - wz : wz - 3;
#pragma unroll
                    for (int r = 0; r < 4; ++r) *(floatx4 *)(zout + gbase[T] + r * cstride + ws) = zo[T][r];
                }
            }
        }

        // rotate the slot table: diagonal d+1 takes the slot of diagonal d-(R-1)
        {
            const int last = soff[R - 1];
#pragma unroll
            for (int k = R - 1; k > 0; --k) soff[k] = soff[k - 1];
            soff[0] = last;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fold + pack for the MFMA scan: A fragments in the exact per-lane register image.
//   slot s < NT-1 : -(W_t L^-1)          (t = s+1; transposed: -(W_t^T L^-T))
//   slot NT-1     :  L^-1                (transposed: L^-T)
// apack[wv][s][q][hl][lane][j] (fp16),  lane = m + 16*gk holds row c = 16wv+m, k = 32q + 8gk + j.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t w_index2(int co, int ci, int dh, int dw, int C, int KH, int KW, int flipH,
                                           int flipW)
{
    int kh = KH - 1 - dh, kw = KW - 1 - dw;
    if (flipH) kh = KH - 1 - kh;
    if (flipW) kw = KW - 1 - kw;
    return (((size_t)co * C + ci) * KH + kh) * KW + kw;
}

__global__ void k_pack_mfma(const float *__restrict__ w, const double *__restrict__ linv, _Float16 *__restrict__ apack,
                            Geom g, int transposed)
{
    const int C = g.C, NT = g.KH * g.KW, NQ = C / 32;
    const size_t total = (size_t)NT * C * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int kc = (int)(i % C);
        const int c = (int)((i / C) % C);
        const int s = (int)(i / ((size_t)C * C));
        double acc = 0.0;
        if (s == NT - 1) {
            acc = transposed ? linv[(size_t)kc * C + c] : linv[(size_t)c * C + kc];
        } else {
            const int t = s + 1, dh = t / g.KW, dw = t % g.KW;
            if (!transposed) {
                // (W_t L^-1)[c][kc] = sum_{m>=kc} w[c][m][t] Linv[m][kc]
                for (int m = kc; m < C; ++m)
                    acc += (double)w[w_index2(c, m, dh, dw, C, g.KH, g.KW, g.flipH, g.flipW)] * linv[(size_t)m * C + kc];
            } else {
                // (W_t^T L^-T)[c][kc] = sum_{m<=kc} w[m][c][t] Linv[kc][m]
                for (int m = 0; m <= kc; ++m)
                    acc += (double)w[w_index2(m, c, dh, dw, C, g.KH, g.KW, g.flipH, g.flipW)] * linv[(size_t)kc * C + m];
            }
            acc = -acc;
        }
        const float v = (float)acc;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)((v - (float)hi) * LO_SCALE);
        const int wv = c / 16, m16 = c % 16, q = kc / 32, gk = (kc % 32) / 8, j = kc % 8;
        const size_t base = ((((size_t)wv * NT + s) * NQ + q) * 2) * 64 * 8;
        apack[base + (size_t)(m16 + 16 * gk) * 8 + j] = hi;
        apack[base + (size_t)64 * 8 + (size_t)(m16 + 16 * gk) * 8 + j] = lo;
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

int launch_pack_mfma(const float *w, const double *linv, void *apack, const Geom &g, int transposed, hipStream_t s)
{
    const size_t total = (size_t)g.KH * g.KW * g.C * g.C;
    size_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_pack_mfma, dim3((unsigned)blocks), dim3(256), 0, s, w, linv, (_Float16 *)apack, g, transposed);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

template <int C, int KH, int KW, int NTILE>
static int launch_one(const float *x, float *z, const void *apack, const Geom &g, int rh, int rw, hipStream_t s)
{
    hipLaunchKernelGGL((k_scan_mfma<C, KH, KW, NTILE>), dim3(g.B), dim3(64 * (C / 16)), 0, s, x, z, (const half8 *)apack,
                       g.H, g.W, rh, rw);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

int launch_scan_mfma(const float *x, const void *apack, float *z, const Geom &g, int rh, int rw, hipStream_t s)
{
    const int nt = g.H <= 16 ? 1 : 2;
#define IFL_CASE(CC, KK, NN) \
    if (g.C == CC && g.KH == KK && g.KW == KK && nt == NN) return launch_one<CC, KK, KK, NN>(x, z, apack, g, rh, rw, s);
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
