// Wide layers (C > 64, a multiple of 16: the C = 256 layers of the ImageNet-32 Glow, if_multiGPU_imagenet32.py) at small
// spatial sizes.  The weights (9 C^2 floats = 2.4 MB at C = 256) fit neither registers nor LDS, and a batch shard has
// too few images to fill the GPU with one workgroup per image.  Here an anti-diagonal of ALL images is one small GEMM
//     Z_d (C x cols) = Wf0 X_d - sum_t Wf_t Z_{d-t}        cols = (image, pixel of the diagonal)
// tiled 16 channels x 16 columns per workgroup over the whole GPU; the diagonals are separate launches (the stream is
// the global barrier between them).  Exact fp32, left fold (wf[t][kc][c], prep.hip), like the general scan.
#include "ifl_common.h"

namespace ifl {

__device__ __forceinline__ size_t wide_addr(int b, int c, int h, int w, const Geom &g, int rh, int rw)
{
    const int hs = rh ? g.H - 1 - h : h;
    const int ws = rw ? g.W - 1 - w : w;
    return (((size_t)b * g.C + c) * g.H + hs) * g.W + ws;
}

__global__ __launch_bounds__(256) void k_scan_wide_step(const float *__restrict__ x, const float *__restrict__ wf,
                                                        float *__restrict__ z, Geom g, int rh, int rw, int d)
{
    constexpr int KT = 32; // k-chunk
    __shared__ float As[KT][17], Bs[KT][17];
    const int C = g.C, H = g.H, W = g.W, KW = g.KW, NT = g.KH * g.KW;
    const int hmin = d - (W - 1) > 0 ? d - (W - 1) : 0;
    const int hmax = d < H - 1 ? d : H - 1;
    const int npix = hmax - hmin + 1, ncols = g.B * npix;
    const int tid = threadIdx.x, ty = tid % 16, tx = tid / 16; // ty: channel in the tile, tx: column in the tile
    const int c0 = blockIdx.x * 16, col0 = blockIdx.y * 16;
    // this thread's output column
    const int col = col0 + tx;
    const bool cok = col < ncols;
    const int ob = cok ? col / npix : 0, oh = hmin + (cok ? col % npix : 0), ow = d - oh;
    float acc = 0.f;
    // chunks of the reduction index (tap, KT input channels), the next chunk's operands prefetched into registers
    // while the current one is multiplied (one chunk at a time this loop is a chain of memory latencies)
    const int nchunk = NT * (C / KT);
    float ra[KT / 16], rb[KT / 16];
    auto fetch = [&](int it) {
        const int t = it / (C / KT), k0 = (it % (C / KT)) * KT;
        const int dh = t / KW, dw = t % KW;
        const float *src = t == 0 ? x : z;
        const int hh = oh - dh, ww = ow - dw;
        const bool ok = cok && hh >= 0 && ww >= 0;
#pragma unroll
        for (int r = 0; r < KT / 16; ++r) {
            // A: wf[t][k0 + kk][c0 + ty], kk = tx + 16 r (16 consecutive channels per row)
            ra[r] = wf[((size_t)t * C + k0 + tx + 16 * r) * C + c0 + ty];
            // B: source pixel of this thread's column for input channel k0 + ty + 16 r
            rb[r] = ok ? src[wide_addr(ob, k0 + ty + 16 * r, hh, ww, g, rh, rw)] : 0.f;
        }
    };
    fetch(0);
    for (int it = 0; it < nchunk; ++it) {
#pragma unroll
        for (int r = 0; r < KT / 16; ++r) {
            As[tx + 16 * r][ty] = ra[r];
            Bs[ty + 16 * r][tx] = rb[r];
        }
        __syncthreads();
        if (it + 1 < nchunk) fetch(it + 1);
        float part = 0.f;
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) part = fmaf(As[kk][ty], Bs[kk][tx], part);
        acc += it < C / KT ? part : -part; // tap 0 is the x term
        __syncthreads();
    }
    if (cok) z[wide_addr(ob, c0 + ty, oh, ow, g, rh, rw)] = acc;
}

bool scan_wide_supported(const Geom &g) { return g.C > 64 && g.C % 32 == 0 && g.H * g.W <= 1024; }

int launch_scan_wide(const float *x, const float *wf, float *z, const Geom &g, int rh, int rw, hipStream_t s)
{
    const int ND = g.H + g.W - 1;
    for (int d = 0; d < ND; ++d) {
        const int hmin = d - (g.W - 1) > 0 ? d - (g.W - 1) : 0;
        const int hmax = d < g.H - 1 ? d : g.H - 1;
        const int ncols = g.B * (hmax - hmin + 1);
        hipLaunchKernelGGL(k_scan_wide_step, dim3(g.C / 16, (ncols + 15) / 16), dim3(256), 0, s, x, wf, z, g, rh, rw, d);
    }
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

} // namespace ifl
