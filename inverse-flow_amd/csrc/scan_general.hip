// General wavefront back-substitution scan (any C, any KH x KW, any H x W): fp32 VALU.
//
// One workgroup owns one image for all H+W-1 anti-diagonals (the reference launches one kernel
// + one cudaDeviceSynchronize per (diagonal, channel): inv_conv_with_bp_kernel_general.cu:97-126).
// With the folded taps (prep.hip) every (pixel, channel) of a diagonal is independent:
//     z_p = Wf[0] x_p - sum_{t>0} Wf[t] z_{p-t}
// LDS holds a ring of the last KH+KW-1 diagonals of z ([slot][h][c], c innermost, odd row
// pitch) and a double-buffered staging tile of x for the current/next diagonal.
// This is the fallback for shapes the MFMA scan does not cover; it is exact fp32.
#include "ifl_common.h"

namespace ifl {

static constexpr int SCAN_T = 256;
static constexpr int SCAN_NP = 8;

__device__ __forceinline__ size_t pix_addr(int b, int c, int h, int w, const Geom &g, int rh, int rw)
{
    const int hs = rh ? g.H - 1 - h : h;
    const int ws = rw ? g.W - 1 - w : w;
    return (((size_t)b * g.C + c) * g.H + hs) * g.W + ws;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_general(const float *__restrict__ xin, const float *__restrict__ wf,
                                                         float *__restrict__ zout, Geom g, int rh, int rw)
{
    extern __shared__ float smem[];
    const int C = g.C, H = g.H, W = g.W, KH = g.KH, KW = g.KW;
    const int R = KH + KW - 1;
    const int Cp = C | 1;
    const int NT = KH * KW;
    float *ring = smem;               // [R][H][Cp]
    float *xs = ring + (size_t)R * H * Cp; // [2][H][Cp]
    float *zero = xs + (size_t)2 * H * Cp; // [Cp]
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
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
            dst[h * Cp + c] = xin[pix_addr(b, c, h, d - h, g, rh, rw)];
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
                        acc[j] = 0.f;
                    }
                    {
                        const float *wt = wf + c;
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
                        const float *wt = wf + (size_t)t * C * C + c;
                        for (int kc = 0; kc < C; ++kc) {
                            const float wv = wt[(size_t)kc * C];
#pragma unroll
                            for (int j = 0; j < SCAN_NP; ++j) acc[j] = fmaf(-wv, zp[j][kc], acc[j]);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < SCAN_NP; ++j) {
                        const int p = pb + j * G;
                        if (p < n) {
                            const int h = hmin + p;
                            zcur[h * Cp + c] = acc[j];
                            zout[pix_addr(b, c, h, d - h, g, rh, rw)] = acc[j];
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

size_t scan_general_lds_bytes(const Geom &g)
{
    const size_t Cp = (size_t)(g.C | 1);
    const size_t R = (size_t)(g.KH + g.KW - 1);
    return ((R + 2) * g.H * Cp + Cp) * sizeof(float);
}

int launch_scan_general(const float *x, const float *wf, float *z, const Geom &g, int rh, int rw, hipStream_t s)
{
    const size_t lds = scan_general_lds_bytes(g);
    if (lds > 160 * 1024)
        IFL_FAIL(IFL_EUNSUPPORTED, "inverse scan: C=%d H=%d K=%dx%d needs %zu B of LDS (> 160 KiB)", g.C, g.H, g.KH,
                 g.KW, lds);
    static bool attr_done = false; // idempotent attribute, benign race
    if (!attr_done) {
        IFL_HIP(hipFuncSetAttribute((const void *)k_scan_general, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL(k_scan_general, dim3(g.B), dim3(SCAN_T), lds, s, x, wf, z, g, rh, rw);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

} // namespace ifl
