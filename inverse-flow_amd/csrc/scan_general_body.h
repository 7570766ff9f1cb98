// General fp32 wavefront scan body shared by k_scan_general (scan_general.hip) and by the in-kernel
// overflow fallback of the MFMA scan (scan_mfma.hip).  See scan_general.hip for the description.
#pragma once
#include "bf16_util.h"
#include "ifl_common.h"

namespace ifl {

static constexpr int SCAN_NP = 8;

__device__ __forceinline__ size_t pix_addr(int b, int c, int h, int w, const Geom &g, int rh, int rw)
{
    const int hs = rh ? g.H - 1 - h : h;
    const int ws = rw ? g.W - 1 - w : w;
    return (((size_t)b * g.C + c) * g.H + hs) * g.W + ws;
}

// rf = 0: wf is the left fold [t][kc][c] (t=0: L^-1, t>0: L^-1 W_t):  z_p = Wf0 x_p - sum Wf_t z_{p-t}
// rf = 1: wf is the right fold [s][kc][c] the MFMA path uses (s<NT-1: -(W_{s+1} L^-1), s=NT-1: L^-1):
//         r_p = x_p + sum Wr_s r_{p-t},  z_p = L^-1 r_p  (the ring then carries r, two barriers per step)
// All threads of the workgroup (NTHR of them) must call this; smem needs scan_general_lds_bytes(g).
// TI: storage type of x (float, or bf16_t widened at the load); z goes to zout (fp32) and / or zout16 (bf16, rounded to
// nearest even at the store), whichever is not NULL.
template <int NTHR, typename TI = float>
__device__ __forceinline__ void scan_general_body(const TI *__restrict__ xin, const float *__restrict__ wf,
                                                  float *__restrict__ zout, const Geom g, int rh, int rw, int rf,
                                                  float *smem, int b, int tid, bf16_t *__restrict__ zout16 = nullptr)
{
    auto put = [&](size_t a, float v) {
        if (zout) zout[a] = v;
        if (zout16) zout16[a] = narrow_bf16(v);
    };
    constexpr int SCAN_T = NTHR;
    const int C = g.C, H = g.H, W = g.W, KH = g.KH, KW = g.KW;
    const int R = KH + KW - 1;
    const int Cp = C | 1;
    const int NT = KH * KW;
    float *ring = smem;               // [R][H][Cp]
    float *xs = ring + (size_t)R * H * Cp; // [2][H][Cp]
    float *zero = xs + (size_t)2 * H * Cp; // [Cp]
    const int ND = H + W - 1;

    const int CT = C < SCAN_T ? C : SCAN_T; // lanes along channels
    const int G = SCAN_T / CT;              // pixel groups
    const int tc = tid % CT, tg = tid / CT;
    const bool worker = tg < G;

    for (int i = tid; i < Cp; i += SCAN_T) zero[i] = 0.f;

    auto load_x = [&](int d, float *dst) {
        const int hmin = d - (W - 1) > 0 ? d - (W - 1) : 0;
        const int hmax = d < H - 1 ? d : H - 1;
        const int n = hmax - hmin + 1;
        for (int it = tid; it < n * C; it += SCAN_T) {
            const int c = it % C, h = hmin + it / C;
            dst[h * Cp + c] = widen(xin[pix_addr(b, c, h, d - h, g, rh, rw)]);
        }
    };

    load_x(0, xs);
    __syncthreads();

    for (int d = 0; d < ND; ++d) {
        const float *xcur = xs + (size_t)(d & 1) * H * Cp;
        if (d + 1 < ND) load_x(d + 1, xs + (size_t)((d + 1) & 1) * H * Cp);
        const int hmin = d - (W - 1) > 0 ? d - (W - 1) : 0;
        const int hmax = d < H - 1 ? d : H - 1;
        const int n = hmax - hmin + 1;
        float *zcur = ring + (size_t)(d % R) * H * Cp;
        if (worker) {
            for (int c = tc; c < C; c += CT) {
                for (int pb = tg; pb < n; pb += G * SCAN_NP) {
                    float acc[SCAN_NP];
                    const float *zp[SCAN_NP];
#pragma unroll
                    for (int j = 0; j < SCAN_NP; ++j) {
                        const int p = pb + j * G;
                        zp[j] = p < n ? xcur + (hmin + p) * Cp : zero;
                        acc[j] = rf ? zp[j][c] : 0.f;
                    }
                    if (!rf) {
                        const float *wt = wf + c;
                        // (unrolled: the weight loads of 8 k-steps are in flight together; one by one this loop is a chain
                        // of L2 latencies -- 3.5 ms per scan at C = 256)
#pragma unroll 8
                        for (int kc = 0; kc < C; ++kc) {
                            const float wv = wt[(size_t)kc * C];
#pragma unroll
                            for (int j = 0; j < SCAN_NP; ++j) acc[j] = fmaf(wv, zp[j][kc], acc[j]);
                        }
                    }
                    for (int t = 1; t < NT; ++t) {
                        const int dh = t / KW, dw = t % KW;
                        const int ds = d - dh - dw;
                        const float *zsrc = ring + (size_t)(((ds % R) + R) % R) * H * Cp;
                        bool any = false;
#pragma unroll
                        for (int j = 0; j < SCAN_NP; ++j) {
                            const int p = pb + j * G;
                            const int h = hmin + p;
                            const int hh = h - dh, ww = d - h - dw;
                            const bool ok = p < n && hh >= 0 && ww >= 0;
                            zp[j] = ok ? zsrc + hh * Cp : zero;
                            any |= ok;
                        }
                        if (!any) continue;
                        const float *wt = wf + (size_t)(rf ? t - 1 : t) * C * C + c;
                        const float sgn = rf ? 1.f : -1.f;
#pragma unroll 8
                        for (int kc = 0; kc < C; ++kc) {
                            const float wv = sgn * wt[(size_t)kc * C];
#pragma unroll
                            for (int j = 0; j < SCAN_NP; ++j) acc[j] = fmaf(wv, zp[j][kc], acc[j]);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < SCAN_NP; ++j) {
                        const int p = pb + j * G;
                        if (p < n) {
                            const int h = hmin + p;
                            zcur[h * Cp + c] = acc[j];
                            if (!rf) put(pix_addr(b, c, h, d - h, g, rh, rw), acc[j]);
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (rf) {
            // z = L^-1 r for the pixels of this diagonal (the next step only writes another slot)
            const float *pw = wf + (size_t)(NT - 1) * C * C;
            if (worker)
                for (int c = tc; c < C; c += CT)
                    for (int p = tg; p < n; p += G) {
                        const float *rp = zcur + (hmin + p) * Cp;
                        float acc = 0.f;
                        for (int kc = 0; kc < C; ++kc) acc = fmaf(pw[(size_t)kc * C + c], rp[kc], acc);
                        put(pix_addr(b, c, hmin + p, d - hmin - p, g, rh, rw), acc);
                    }
        }
    }
}

} // namespace ifl
