// Kernels for small layers (a handful of channels, small images: the reference's MNIST configurations, C = 1 at
// 28x28, C = 4 at 14x14, C = 8 at 7x7 -- if_cnn_mnist.py, if_glow_mnist.py:156-190), where the general kernels spend
// their time in per-step global round trips and in a handful of workgroups:
//   k_scan_resident    the wavefront scan with the whole image, its result and the folded taps resident in LDS: one
//                      workgroup per image, one barrier per anti-diagonal, no global access inside the sweep;
//   k_wgrad_batchpar   the weight gradient with one workgroup per image (the direct kernel uses one per (co, ci):
//                      a single workgroup at C = 1) + a fixed-order reduction over the batch (deterministic).
// Exact fp32 like the general kernels (left fold, scan_general.hip).
#include "ifl_common.h"

namespace ifl {

__device__ __forceinline__ size_t stored_addr(int b, int c, int h, int w, const Geom &g, int rh, int rw)
{
    const int hs = rh ? g.H - 1 - h : h;
    const int ws = rw ? g.W - 1 - w : w;
    return (((size_t)b * g.C + c) * g.H + hs) * g.W + ws;
}

// wf[t][kc][c] (t = 0: L^-1, t > 0: L^-1 W_t):  z_p = Wf0 x_p - sum_t Wf_t z_{p-t}
__global__ __launch_bounds__(256) void k_scan_resident(const float *__restrict__ xin, const float *__restrict__ wf,
                                                       float *__restrict__ zout, Geom g, int rh, int rw)
{
    extern __shared__ float smem[];
    const int C = g.C, H = g.H, W = g.W, KH = g.KH, KW = g.KW, NT = KH * KW;
    const int HW = H * W, n = C * HW;
    float *xs = smem;          // [C][H][W] logical coordinates
    float *zs = xs + n;        // [C][H][W]
    float *ws = zs + n;        // [NT][C][C]
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < n; i += 256) {
        const int w = i % W, h = (i / W) % H, c = i / HW;
        xs[i] = xin[stored_addr(b, c, h, w, g, rh, rw)];
    }
    for (int i = tid; i < NT * C * C; i += 256) ws[i] = wf[i];
    __syncthreads();
    const int ND = H + W - 1;
    for (int d = 0; d < ND; ++d) {
        const int hmin = d - (W - 1) > 0 ? d - (W - 1) : 0;
        const int hmax = d < H - 1 ? d : H - 1;
        const int items = (hmax - hmin + 1) * C;
        for (int it = tid; it < items; it += 256) {
            const int c = it % C, h = hmin + it / C, w = d - h;
            float acc = 0.f;
            for (int kc = 0; kc < C; ++kc) acc = fmaf(ws[kc * C + c], xs[kc * HW + h * W + w], acc);
            for (int t = 1; t < NT; ++t) {
                const int hh = h - t / KW, ww = w - t % KW;
                if (hh < 0 || ww < 0) continue;
                const float *wt = ws + (size_t)t * C * C + c;
                const float *zp = zs + hh * W + ww;
                for (int kc = 0; kc < C; ++kc) acc = fmaf(-wt[kc * C], zp[kc * HW], acc);
            }
            zs[c * HW + h * W + w] = acc;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += 256) {
        const int w = i % W, h = (i / W) % H, c = i / HW;
        zout[stored_addr(b, c, h, w, g, rh, rw)] = zs[i];
    }
}

size_t scan_resident_lds_bytes(const Geom &g)
{
    return ((size_t)2 * g.C * g.H * g.W + (size_t)g.KH * g.KW * g.C * g.C) * sizeof(float);
}

bool scan_resident_supported(const Geom &g) { return g.C <= 8 && scan_resident_lds_bytes(g) <= 64 * 1024; }

int launch_scan_resident(const float *x, const float *wf, float *z, const Geom &g, int rh, int rw, hipStream_t s)
{
    hipLaunchKernelGGL(k_scan_resident, dim3(g.B), dim3(256), scan_resident_lds_bytes(g), s, x, wf, z, g, rh, rw);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

// partial[b][co][ci][t] = sum_{oh,ow} gz[b][co][oh][ow] * x[b][ci][oh-pt+kh][ow-pl+kw]   (same-size layers)
__global__ __launch_bounds__(256) void k_wgrad_batchpar(const float *__restrict__ gz, const float *__restrict__ x,
                                                        float *__restrict__ partial, int C, int H, int W, int KH, int KW,
                                                        int pt, int pl)
{
    extern __shared__ float smem[];
    const int HW = H * W, n = C * HW, NT = KH * KW, NO = C * C * NT;
    float *gs = smem, *xs = smem + n;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < n; i += 256) {
        gs[i] = gz[(size_t)b * n + i];
        xs[i] = x[(size_t)b * n + i];
    }
    __syncthreads();
    for (int o = tid; o < NO; o += 256) {
        const int t = o % NT, ci = (o / NT) % C, co = o / (NT * C);
        const int kh = t / KW, kw = t % KW;
        // rows / columns of the output for which the tap reads inside the image
        const int oh0 = pt - kh > 0 ? pt - kh : 0, oh1 = H + pt - kh < H ? H + pt - kh : H;
        const int ow0 = pl - kw > 0 ? pl - kw : 0, ow1 = W + pl - kw < W ? W + pl - kw : W;
        float acc = 0.f;
        for (int oh = oh0; oh < oh1; ++oh) {
            const float *gp = gs + co * HW + oh * W, *xp = xs + ci * HW + (oh - pt + kh) * W + (kw - pl);
            for (int ow = ow0; ow < ow1; ++ow) acc = fmaf(gp[ow], xp[ow], acc);
        }
        partial[(size_t)b * NO + o] = acc;
    }
}

// dw[o] = scale * sum_b partial[b][o] (fixed order), masked like k_wgrad_direct
__global__ __launch_bounds__(256) void k_wgrad_batchred(const float *__restrict__ partial, float *__restrict__ dw, int B,
                                                        int C, int KH, int KW, float scale, int mask_mode, int mkh, int mkw)
{
    const int NT = KH * KW, NO = C * C * NT;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= NO) return;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += partial[(size_t)b * NO + o];
    const int t = o % NT, ci = (o / NT) % C, co = o / (NT * C);
    float val = acc * scale;
    if (mask_mode && t / KW == mkh && t % KW == mkw) {
        if (mask_mode == 1 && ci >= co) val = 0.f;
        if (mask_mode == 2 && ci > co) val = 0.f;
    }
    dw[o] = val;
}

size_t wgrad_small_workspace_bytes(int B, int C, int KH, int KW) { return (size_t)B * C * C * KH * KW * sizeof(float) + 256; }

bool wgrad_small_supported(int B, int C, int H, int W) { return B >= 1 && C <= 8 && (size_t)2 * C * H * W * sizeof(float) <= 64 * 1024; }

int launch_wgrad_small(const float *gz, const float *x, float *dw, void *ws, int B, int C, int H, int W, int KH, int KW,
                       int pt, int pl, float scale, int mask_mode, int mkh, int mkw, hipStream_t s)
{
    float *partial = (float *)ws;
    hipLaunchKernelGGL(k_wgrad_batchpar, dim3(B), dim3(256), (size_t)2 * C * H * W * sizeof(float), s, gz, x, partial, C, H, W,
                       KH, KW, pt, pl);
    IFL_HIP(hipGetLastError());
    const int NO = C * C * KH * KW;
    hipLaunchKernelGGL(k_wgrad_batchred, dim3((NO + 255) / 256), dim3(256), 0, s, partial, dw, B, C, KH, KW, scale, mask_mode,
                       mkh, mkw);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

} // namespace ifl
