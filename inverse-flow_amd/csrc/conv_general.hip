// General direct convolution and convolution weight-gradient (any channel counts, any K,
// asymmetric top/left padding): fp32 VALU.  These serve
//   - xhat = A z  (ifl_forward_f32; the order is just where the padding goes:
//     inf/layers/inv_conv.py:126-144, conv.py:103-108),
//   - dW of the inverse conv = -wgrad(dx, z),
//   - SelfNormConv's conv2d / backward_weight / backward_input (inf/layers/selfnorm.py:42-82)
// for shapes the MFMA kernels do not cover.
#include "ifl_common.h"

namespace ifl {

// One thread per output element; consecutive threads walk (oh,ow) so input reads coalesce
// along W and the weight address is wave-uniform whenever OH*OW >= 64.
__global__ __launch_bounds__(256) void k_conv_direct(const float *__restrict__ in, const float *__restrict__ w,
                                                     const float *__restrict__ bias, float *__restrict__ out, int B,
                                                     int Ci, int Co, int H, int W, int OH, int OW, int KH, int KW,
                                                     int pt, int pl)
{
    const size_t total = (size_t)B * Co * OH * OW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ow = (int)(i % OW);
        const int oh = (int)((i / OW) % OH);
        const int co = (int)((i / ((size_t)OW * OH)) % Co);
        const int b = (int)(i / ((size_t)OW * OH * Co));
        float acc = bias ? bias[co] : 0.f;
        const float *wb = w + (size_t)co * Ci * KH * KW;
        const float *ib = in + (size_t)b * Ci * H * W;
        for (int ci = 0; ci < Ci; ++ci) {
            for (int kh = 0; kh < KH; ++kh) {
                const int ih = oh - pt + kh;
                if (ih < 0 || ih >= H) continue;
                for (int kw = 0; kw < KW; ++kw) {
                    const int iw = ow - pl + kw;
                    if (iw < 0 || iw >= W) continue;
                    acc = fmaf(wb[(ci * KH + kh) * KW + kw], ib[((size_t)ci * H + ih) * W + iw], acc);
                }
            }
        }
        out[i] = acc;
    }
}

int launch_conv_direct(const float *in, const float *w, const float *bias, float *out, int B, int Ci, int Co, int H,
                       int W, int OH, int OW, int KH, int KW, int pt, int pl, hipStream_t s)
{
    const size_t total = (size_t)B * Co * OH * OW;
    if (total == 0) return IFL_OK;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_conv_direct, dim3((unsigned)blocks), dim3(256), 0, s, in, w, bias, out, B, Ci, Co, H, W, OH, OW,
                       KH, KW, pt, pl);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

// One workgroup per (co, ci): threads stride over (b, oh, ow), keep up to WG_TAPS tap partials
// in registers, then a fixed-order wave-shuffle + LDS reduction (deterministic, no atomics).
static constexpr int WG_TAPS = 9;

__global__ __launch_bounds__(256) void k_wgrad_direct(const float *__restrict__ gz, const float *__restrict__ x,
                                                      float *__restrict__ dw, int B, int Ci, int Co, int H, int W,
                                                      int OH, int OW, int KH, int KW, int pt, int pl, float scale,
                                                      int mask_mode, int mkh, int mkw)
{
    __shared__ float red[4][WG_TAPS];
    const int co = blockIdx.x / Ci, ci = blockIdx.x % Ci;
    const int NT = KH * KW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t npix = (size_t)OH * OW;
    const size_t total = (size_t)B * npix;
    for (int t0 = 0; t0 < NT; t0 += WG_TAPS) {
        const int nt = NT - t0 < WG_TAPS ? NT - t0 : WG_TAPS;
        float acc[WG_TAPS];
#pragma unroll
        for (int j = 0; j < WG_TAPS; ++j) acc[j] = 0.f;
        for (size_t i = threadIdx.x; i < total; i += blockDim.x) {
            const int ow = (int)(i % OW);
            const int oh = (int)((i / OW) % OH);
            const int b = (int)(i / npix);
            const float gv = gz[(((size_t)b * Co + co) * OH + oh) * OW + ow];
            const float *xb = x + ((size_t)b * Ci + ci) * H * W;
#pragma unroll
            for (int j = 0; j < WG_TAPS; ++j) {
                if (j < nt) {
                    const int t = t0 + j;
                    const int kh = t / KW, kw = t % KW;
                    const int ih = oh - pt + kh, iw = ow - pl + kw;
                    if (ih >= 0 && ih < H && iw >= 0 && iw < W) acc[j] = fmaf(gv, xb[(size_t)ih * W + iw], acc[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < WG_TAPS; ++j) {
            float v = acc[j];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if (lane == 0) red[wave][j] = v;
        }
        __syncthreads();
        if (threadIdx.x < (unsigned)nt) {
            const int t = t0 + threadIdx.x;
            const int kh = t / KW, kw = t % KW;
            float v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) * scale;
            if (mask_mode && kh == mkh && kw == mkw) {
                if (mask_mode == 1 && ci >= co) v = 0.f;
                if (mask_mode == 2 && ci > co) v = 0.f;
            }
            dw[(((size_t)co * Ci + ci) * KH + kh) * KW + kw] = v;
        }
        __syncthreads();
    }
}

int launch_wgrad_direct(const float *gz, const float *x, float *dw, int B, int Ci, int Co, int H, int W, int OH,
                        int OW, int KH, int KW, int pt, int pl, float scale, int mask_mode, int mkh, int mkw,
                        hipStream_t s)
{
    if (Ci * Co == 0) return IFL_OK;
    hipLaunchKernelGGL(k_wgrad_direct, dim3((unsigned)(Ci * Co)), dim3(256), 0, s, gz, x, dw, B, Ci, Co, H, W, OH, OW,
                       KH, KW, pt, pl, scale, mask_mode, mkh, mkw);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

} // namespace ifl
