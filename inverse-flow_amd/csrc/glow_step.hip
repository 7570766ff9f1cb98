// The Glow-step neighbours of the inverse-conv layer: ActNorm, Squeeze / UnSqueeze and the
// affine part of Coupling.  All of them are one pass over an NCHW activation: HBM-bound elementwise work with, at
// most, a per-channel or per-image reduction -- 16 bytes per lane, whole (image, channel) planes per workgroup,
// fixed-order two-stage reductions (deterministic: no atomics on floats).
//
//   ActNorm   inf/layers/actnorm.py:18-69      y = (x - t_c) exp(-ls_c),  logdet = -H W sum_c ls_c
//   Squeeze   inf/layers/squeeze.py:5-25       space_to_depth / depth_to_space (a permutation)
//   Coupling  inf/layers/coupling.py:66-98     z2 = x2 exp(log_s) + t,  log_s = 2 tanh(h_s / 2),  h = net(x1)
//             (the conditioner net is three library convolutions and stays where it is: only what touches the
//              activation elementwise is here)
#include "ifl_common.h"
#include "bf16_util.h"

namespace ifl {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

static constexpr int GS_T = 256; // threads of every kernel here

// Storage types of an activation: float, or bf16 as its 16-bit pattern (the *_bf16 entry points: SURVEY 8b dtype row,
// "bf16 storage / fp32 accumulate").  Arithmetic is fp32 either way; with the same loop structure and summation order, so
// that a bf16 call returns exactly the rounded result of the f32 call on the widened inputs.
__device__ __forceinline__ void put(float *p, float v) { *p = v; }
__device__ __forceinline__ void put(bf16_t *p, float v) { *p = narrow_bf16(v); }
__device__ __forceinline__ f4 ld4(const float *p, int i) { return ((const f4 *)p)[i]; }
__device__ __forceinline__ f4 ld4(const bf16_t *p, int i)
{
    const us4 v = ((const us4 *)p)[i];
    return f4{widen(v[0]), widen(v[1]), widen(v[2]), widen(v[3])};
}
__device__ __forceinline__ void st4(float *p, int i, f4 v) { ((f4 *)p)[i] = v; }
__device__ __forceinline__ void st4(bf16_t *p, int i, f4 v)
{
    ((us4 *)p)[i] = us4{narrow_bf16(v[0]), narrow_bf16(v[1]), narrow_bf16(v[2]), narrow_bf16(v[3])};
}
// all pointers aligned for four-element accesses of their type
template <class T> __device__ __forceinline__ bool vec_ok(const T *a) { return (((uintptr_t)a) & (4 * sizeof(T) - 1)) == 0; }
template <class T, class... R> __device__ __forceinline__ bool vec_ok(const T *a, const R *...r) { return vec_ok(a) && vec_ok(r...); }

__device__ __forceinline__ float block_sum(float v, float *sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads(); // (sh may still be read from a previous call)
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3]; // fixed order
}

// ---- ActNorm ------------------------------------------------------------------------------------------------------
// one workgroup per (image, channel) plane; reverse: y = x exp(ls) + t  (actnorm.py:40-54)
template <class T>
__global__ __launch_bounds__(GS_T) void k_actnorm(const T *__restrict__ x, const float *__restrict__ tr,
                                                  const float *__restrict__ ls, T *__restrict__ y, int C, int HW, int reverse,
                                                  float *__restrict__ logdet, int B)
{
    __shared__ float sh[4];
    const size_t plane = blockIdx.x;
    const int c = (int)(plane % C);
    const float t = tr[c], l = ls[c];
    const float sc = reverse ? expf(l) : expf(-l);
    const float of = reverse ? t : -t * sc; // forward: (x - t) sc = x sc - t sc
    const T *xp = x + plane * HW;
    T *yp = y + plane * HW;
    if ((HW & 3) == 0 && vec_ok(xp, yp)) {
        for (int i = threadIdx.x; i < HW / 4; i += GS_T) {
            const f4 v = ld4(xp, i);
            st4(yp, i, f4{fmaf(v[0], sc, of), fmaf(v[1], sc, of), fmaf(v[2], sc, of), fmaf(v[3], sc, of)});
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += GS_T) put(yp + i, fmaf(widen(xp[i]), sc, of));
    }
    if (logdet && blockIdx.x == 0) { // logdet[b] = -HW sum_c ls_c (actnorm.py:59-67): the first workgroup, behind its plane
        float a = 0.f;
        for (int cc = threadIdx.x; cc < C; cc += GS_T) a += ls[cc];
        const float s = -block_sum(a, sh) * (float)HW;
        for (int b = threadIdx.x; b < B; b += GS_T) logdet[b] = s;
    }
}
// note on rounding: the reference computes (x - t) * exp(-ls); x*sc - t*sc differs by one rounding of t*sc
// (relative 6e-8 of |t sc|): inside the 1e-5 tolerance of the path, and one FMA instead of two operations.

// backward, stage 1: gx = gy exp(-ls); partial[plane] = {sum gy, sum gy x} over the plane
template <class T>
__global__ __launch_bounds__(GS_T) void k_actnorm_bwd(const T *__restrict__ gy, const T *__restrict__ x,
                                                      const float *__restrict__ ls, T *__restrict__ gx,
                                                      float *__restrict__ partial, int C, int HW)
{
    __shared__ float sh[4];
    const size_t plane = blockIdx.x;
    const int c = (int)(plane % C);
    const float sc = expf(-ls[c]);
    const T *gp = gy + plane * HW, *xp = x + plane * HW;
    T *op = gx + plane * HW;
    float s0 = 0.f, s1 = 0.f;
    if ((HW & 3) == 0 && vec_ok(gp, xp, op)) {
        for (int i = threadIdx.x; i < HW / 4; i += GS_T) {
            const f4 g = ld4(gp, i), v = ld4(xp, i);
            st4(op, i, g * sc);
            s0 += (g[0] + g[1]) + (g[2] + g[3]);
            s1 += (g[0] * v[0] + g[1] * v[1]) + (g[2] * v[2] + g[3] * v[3]);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += GS_T) {
            const float g = widen(gp[i]);
            put(op + i, g * sc);
            s0 += g;
            s1 += g * widen(xp[i]);
        }
    }
    const float t0 = block_sum(s0, sh), t1 = block_sum(s1, sh);
    if (threadIdx.x == 0) {
        partial[2 * plane] = t0;
        partial[2 * plane + 1] = t1;
    }
}
// stage 2 (one wave per channel: lane i sums the images i, i+64, ... in order, then a fixed shuffle tree):
//   gt_c = -sc sum gy;  gls_c = -sc (sum gy x - t sum gy) - HW sum_b g_logdet[b]
__global__ __launch_bounds__(64) void k_actnorm_bwd_fin(const float *__restrict__ partial, const float *__restrict__ tr,
                                                        const float *__restrict__ ls, const float *__restrict__ g_logdet,
                                                        float *__restrict__ gt, float *__restrict__ gls, int B, int C, int HW)
{
    const int c = blockIdx.x;
    float s0 = 0.f, s1 = 0.f, sl = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) {
        s0 += partial[2 * ((size_t)b * C + c)];
        s1 += partial[2 * ((size_t)b * C + c) + 1];
        if (g_logdet) sl += g_logdet[b];
    }
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_down(s0, o, 64);
        s1 += __shfl_down(s1, o, 64);
        sl += __shfl_down(sl, o, 64);
    }
    if (threadIdx.x == 0) {
        const float sc = expf(-ls[c]);
        if (gt) gt[c] = -sc * s0;
        if (gls) gls[c] = -sc * (s1 - tr[c] * s0) - (float)HW * sl;
    }
}

// data-dependent initialisation (actnorm.py:21-28): mean_c and log(std_c + 1e-8), std unbiased (torch.std).
// stage 1: per plane {sum x, sum x^2 about the plane's own mean}; stage 2 combines the planes of a channel
// (Chan's parallel variance: no cancellation of large sums).
__global__ __launch_bounds__(GS_T) void k_plane_moments(const float *__restrict__ x, float *__restrict__ partial, int HW)
{
    __shared__ float sh[4];
    const size_t plane = blockIdx.x;
    const float *xp = x + plane * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += GS_T) s += xp[i];
    const float mean = block_sum(s, sh) / (float)HW;
    float q = 0.f;
    for (int i = threadIdx.x; i < HW; i += GS_T) { // (second read of the plane: L2-resident, 4 KB at 32x32)
        const float d = xp[i] - mean;
        q += d * d;
    }
    const float m2 = block_sum(q, sh);
    if (threadIdx.x == 0) {
        partial[2 * plane] = mean;
        partial[2 * plane + 1] = m2;
    }
}
__global__ void k_actnorm_stats_fin(const float *__restrict__ partial, float *__restrict__ mean, float *__restrict__ logstd,
                                    int B, int C, int HW)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double n = 0.0, mu = 0.0, m2 = 0.0;
    for (int b = 0; b < B; ++b) {
        const double nb = HW, mb = partial[2 * ((size_t)b * C + c)], qb = partial[2 * ((size_t)b * C + c) + 1];
        const double d = mb - mu, nn = n + nb;
        mu += d * nb / nn;
        m2 += qb + d * d * n * nb / nn;
        n = nn;
    }
    mean[c] = (float)mu;
    logstd[c] = (float)log(sqrt(m2 / (n - 1.0)) + 1e-8);
}

// ---- Squeeze: y[b][4c + 2dy + dx][h2][w2] = x[b][c][2 h2 + dy][2 w2 + dx]  (squeeze.py:5-13); reverse is the inverse
// permutation (squeeze.py:16-25).  One workgroup per (image, input channel of the large layout); a thread moves
// four consecutive columns of the large image (two pixels of each of two small planes).
template <class E>
__global__ __launch_bounds__(GS_T) void k_squeeze(const E *__restrict__ src, E *__restrict__ dst, int H, int W, int reverse)
{
    // H, W: size of the LARGE image (even).  large: [plane][H][W]; small: [plane*4 + 2dy+dx][H/2][W/2]
    const size_t plane = blockIdx.x;
    const int H2 = H / 2, W2 = W / 2;
    typedef E e4 __attribute__((ext_vector_type(4)));
    typedef E e2 __attribute__((ext_vector_type(2)));
    const E *lg_r = src + plane * H * W;       // forward reads the large layout
    E *lg_w = dst + plane * H * W;             // reverse writes it
    const E *sm_r = src + plane * 4 * H2 * W2; // reverse reads the small layout
    E *sm_w = dst + plane * 4 * H2 * W2;
    if ((W & 3) == 0 && vec_ok(src, dst)) {
        const int Q = W / 4;
        for (int i = threadIdx.x; i < H * Q; i += GS_T) {
            const int h = i / Q, q = i % Q, h2 = h >> 1, dy = h & 1;
            const size_t lo = (size_t)h * W + 4 * q;
            const size_t s0 = ((size_t)(2 * dy) * H2 + h2) * W2 + 2 * q, s1 = ((size_t)(2 * dy + 1) * H2 + h2) * W2 + 2 * q;
            if (!reverse) {
                const e4 v = *(const e4 *)(lg_r + lo);
                *(e2 *)(sm_w + s0) = e2{v[0], v[2]};
                *(e2 *)(sm_w + s1) = e2{v[1], v[3]};
            } else {
                const e2 a = *(const e2 *)(sm_r + s0), b = *(const e2 *)(sm_r + s1);
                *(e4 *)(lg_w + lo) = e4{a[0], b[0], a[1], b[1]};
            }
        }
    } else {
        for (int i = threadIdx.x; i < H * W; i += GS_T) {
            const int h = i / W, w = i % W;
            const size_t so = ((size_t)(2 * (h & 1) + (w & 1)) * H2 + (h >> 1)) * W2 + (w >> 1);
            if (!reverse) sm_w[so] = lg_r[i];
            else lg_w[i] = sm_r[so];
        }
    }
}

// ---- Coupling, the affine part (coupling.py:66-98) ------------------------------------------------------------------
// x, y: (B, C, H, W); h = net(x1): (B, C, H, W) with h_s = h[:, 0::2], t = h[:, 1::2] (C/2 channels each).
// One workgroup per (image, j), j < C/2: copies plane x1_j, transforms plane x2_j with the channel pair (2j, 2j+1) of h.
//   forward: z2 = x2 exp(log_s) + t,  partial[b][j] = sum log_s;   reverse: z2 = (x2 - t) exp(-log_s)
__device__ __forceinline__ float coupling_logs(float hs) { return 2.0f * tanhf(0.5f * hs); }

template <class T>
__global__ __launch_bounds__(GS_T) void k_coupling(const T *__restrict__ x, const T *__restrict__ h, T *__restrict__ y,
                                                   float *__restrict__ partial, int C, int HW, int reverse)
{
    __shared__ float sh[4];
    const int Ch = C / 2;
    const int b = blockIdx.x / Ch, j = blockIdx.x % Ch;
    const T *x1 = x + ((size_t)b * C + j) * HW, *x2 = x + ((size_t)b * C + Ch + j) * HW;
    const T *hs = h + ((size_t)b * C + 2 * j) * HW, *ht = hs + HW;
    T *y1 = y + ((size_t)b * C + j) * HW, *y2 = y + ((size_t)b * C + Ch + j) * HW;
    float acc = 0.f;
    auto one = [&](float xv, float hsv, float htv, float &out) {
        const float l = coupling_logs(hsv);
        out = reverse ? (xv - htv) * expf(-l) : fmaf(xv, expf(l), htv);
        acc += l;
    };
    if ((HW & 3) == 0 && vec_ok(x, h, y)) {
        for (int i = threadIdx.x; i < HW / 4; i += GS_T) {
            if (y1 != x1) st4(y1, i, ld4(x1, i)); // (a bf16 value widened and narrowed again is itself)
            const f4 xv = ld4(x2, i), a = ld4(hs, i), t = ld4(ht, i);
            f4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float r;
                one(xv[e], a[e], t[e], r);
                o[e] = r;
            }
            st4(y2, i, o);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += GS_T) {
            if (y1 != x1) y1[i] = x1[i];
            float r;
            one(widen(x2[i]), widen(hs[i]), widen(ht[i]), r);
            put(y2 + i, r);
        }
    }
    if (!reverse && partial) {
        const float s = block_sum(acc, sh);
        if (threadIdx.x == 0) partial[(size_t)b * Ch + j] = s;
    }
}
// logdet[b] = sum_j partial[b][j] (channel pairs in order)
__global__ void k_coupling_logdet(const float *__restrict__ partial, float *__restrict__ logdet, int B, int Ch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int j = 0; j < Ch; ++j) s += partial[(size_t)b * Ch + j];
    logdet[b] = s;
}
// backward of the forward direction: given gy (B,C,H,W) and g_logdet (B, may be NULL)
//   gx1 = gy1 (the part through the net is the caller's: it backpropagates gh through the conditioner)
//   gx2 = gy2 exp(log_s);  gh_t = gy2;  gh_s = (gy2 x2 exp(log_s) + g_logdet[b]) (1 - tanh^2(h_s/2))
template <class T>
__global__ __launch_bounds__(GS_T) void k_coupling_bwd(const T *__restrict__ gy, const float *__restrict__ g_logdet,
                                                       const T *__restrict__ x, const T *__restrict__ h, T *__restrict__ gx,
                                                       T *__restrict__ gh, int C, int HW)
{
    const int Ch = C / 2;
    const int b = blockIdx.x / Ch, j = blockIdx.x % Ch;
    const T *g1 = gy + ((size_t)b * C + j) * HW, *g2 = gy + ((size_t)b * C + Ch + j) * HW;
    const T *x2 = x + ((size_t)b * C + Ch + j) * HW;
    const T *hs = h + ((size_t)b * C + 2 * j) * HW;
    T *o1 = gx + ((size_t)b * C + j) * HW, *o2 = gx + ((size_t)b * C + Ch + j) * HW;
    T *ghs = gh + ((size_t)b * C + 2 * j) * HW, *ght = ghs + HW;
    const float gl = g_logdet ? g_logdet[b] : 0.f;
    auto one = [&](float g, float xv, float hsv, float &ox2, float &ohs) {
        const float th = tanhf(0.5f * hsv), e = expf(2.0f * th);
        ox2 = g * e;
        ohs = (g * xv * e + gl) * (1.0f - th * th);
    };
    if ((HW & 3) == 0 && vec_ok(gy, x, h, gx, gh)) {
        for (int i = threadIdx.x; i < HW / 4; i += GS_T) {
            const f4 g = ld4(g2, i), xv = ld4(x2, i), a = ld4(hs, i);
            st4(o1, i, ld4(g1, i));
            st4(ght, i, g);
            f4 ox, oh;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float r0, r1;
                one(g[e], xv[e], a[e], r0, r1);
                ox[e] = r0;
                oh[e] = r1;
            }
            st4(o2, i, ox);
            st4(ghs, i, oh);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += GS_T) {
            const float g = widen(g2[i]);
            float r0, r1;
            one(g, widen(x2[i]), widen(hs[i]), r0, r1);
            o1[i] = g1[i];
            put(o2 + i, r0);
            ght[i] = g2[i];
            put(ghs + i, r1);
        }
    }
}

static int check_dims(const char *who, int B, int C, int H, int W)
{
    if (B < 0 || C < 1 || H < 1 || W < 1) IFL_FAIL(IFL_EINVAL, "%s: bad shape B=%d C=%d H=%d W=%d", who, B, C, H, W);
    return IFL_OK;
}

} // namespace ifl

using namespace ifl;

extern "C" {

size_t ifl_glow_workspace_bytes(int B, int C)
{
    return (size_t)2 * (B > 0 ? B : 0) * (C > 0 ? C : 0) * sizeof(float) + 256;
}

} // extern "C"

namespace ifl {
template <class T>
static int actnorm_impl(const char *who, const T *x, const float *translation, const float *log_scale, T *y, float *logdet,
                        int B, int C, int H, int W, int reverse, ifl_stream_t stream)
{
    clear_error();
    if (int rc = check_dims(who, B, C, H, W)) return rc;
    if (B == 0) return IFL_OK;
    if (!x || !translation || !log_scale || !y) IFL_FAIL(IFL_EINVAL, "%s: null pointer", who);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_actnorm<T>, dim3((unsigned)((size_t)B * C)), dim3(GS_T), 0, s, x, translation, log_scale, y, C, H * W,
                       reverse, reverse ? (float *)nullptr : logdet, B);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}
template <class T>
static int actnorm_backward_impl(const char *who, const T *gy, const float *g_logdet, const T *x, const float *translation,
                                 const float *log_scale, T *gx, float *g_translation, float *g_log_scale, int B, int C, int H,
                                 int W, void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    if (int rc = check_dims(who, B, C, H, W)) return rc;
    if (!gy || !x || !translation || !log_scale || !gx) IFL_FAIL(IFL_EINVAL, "%s: null pointer", who);
    if (ws_bytes < ifl_glow_workspace_bytes(B, C) || (!ws && B > 0))
        IFL_FAIL(IFL_EWORKSPACE, "%s: workspace of %zu bytes needed", who, ifl_glow_workspace_bytes(B, C));
    hipStream_t s = (hipStream_t)stream;
    float *partial = (float *)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    if (B > 0)
        hipLaunchKernelGGL(k_actnorm_bwd<T>, dim3((unsigned)((size_t)B * C)), dim3(GS_T), 0, s, gy, x, log_scale, gx, partial, C,
                           H * W);
    hipLaunchKernelGGL(k_actnorm_bwd_fin, dim3(C), dim3(64), 0, s, partial, translation, log_scale, g_logdet, g_translation,
                       g_log_scale, B, C, H * W);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}
} // namespace ifl

extern "C" {

int ifl_actnorm_f32(const float *x, const float *translation, const float *log_scale, float *y, float *logdet, int B,
                    int C, int H, int W, int reverse, ifl_stream_t stream)
{
    return actnorm_impl("ifl_actnorm_f32", x, translation, log_scale, y, logdet, B, C, H, W, reverse, stream);
}
int ifl_actnorm_bf16(const uint16_t *x, const float *translation, const float *log_scale, uint16_t *y, float *logdet, int B,
                     int C, int H, int W, int reverse, ifl_stream_t stream)
{
    return actnorm_impl("ifl_actnorm_bf16", x, translation, log_scale, y, logdet, B, C, H, W, reverse, stream);
}

int ifl_actnorm_backward_f32(const float *gy, const float *g_logdet, const float *x, const float *translation,
                             const float *log_scale, float *gx, float *g_translation, float *g_log_scale, int B, int C,
                             int H, int W, void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    return actnorm_backward_impl("ifl_actnorm_backward_f32", gy, g_logdet, x, translation, log_scale, gx, g_translation,
                                 g_log_scale, B, C, H, W, ws, ws_bytes, stream);
}
int ifl_actnorm_backward_bf16(const uint16_t *gy, const float *g_logdet, const uint16_t *x, const float *translation,
                              const float *log_scale, uint16_t *gx, float *g_translation, float *g_log_scale, int B, int C,
                              int H, int W, void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    return actnorm_backward_impl("ifl_actnorm_backward_bf16", gy, g_logdet, x, translation, log_scale, gx, g_translation,
                                 g_log_scale, B, C, H, W, ws, ws_bytes, stream);
}

int ifl_actnorm_stats_f32(const float *x, float *mean, float *log_std, int B, int C, int H, int W, void *ws,
                          size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    if (int rc = check_dims("ifl_actnorm_stats_f32", B, C, H, W)) return rc;
    if (!x || !mean || !log_std || !ws) IFL_FAIL(IFL_EINVAL, "ifl_actnorm_stats_f32: null pointer");
    if ((size_t)B * H * W < 2) IFL_FAIL(IFL_EINVAL, "ifl_actnorm_stats_f32: needs at least two values per channel");
    if (ws_bytes < ifl_glow_workspace_bytes(B, C))
        IFL_FAIL(IFL_EWORKSPACE, "ifl_actnorm_stats_f32: workspace of %zu bytes needed", ifl_glow_workspace_bytes(B, C));
    hipStream_t s = (hipStream_t)stream;
    float *partial = (float *)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    hipLaunchKernelGGL(k_plane_moments, dim3((unsigned)((size_t)B * C)), dim3(GS_T), 0, s, x, partial, H * W);
    hipLaunchKernelGGL(k_actnorm_stats_fin, dim3((C + 63) / 64), dim3(64), 0, s, partial, mean, log_std, B, C, H * W);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

} // extern "C"

namespace ifl {
template <class E>
static int squeeze_impl(const char *who, const E *x, E *y, int B, int C, int H, int W, int reverse, ifl_stream_t stream)
{
    clear_error();
    /* (C, H, W): the LARGE layout -- the input of space_to_depth, the output of depth_to_space */
    if (int rc = check_dims(who, B, C, H, W)) return rc;
    if ((H | W) & 1) IFL_FAIL(IFL_EINVAL, "%s: H=%d, W=%d must be even", who, H, W);
    if (B == 0) return IFL_OK;
    if (!x || !y || x == y) IFL_FAIL(IFL_EINVAL, "%s: null or aliased pointers", who);
    hipLaunchKernelGGL(k_squeeze<E>, dim3((unsigned)((size_t)B * C)), dim3(GS_T), 0, (hipStream_t)stream, x, y, H, W, reverse);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}
template <class T>
static int coupling_impl(const char *who, const T *x, const T *h, T *y, float *logdet, int B, int C, int H, int W, int reverse,
                         void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    if (int rc = check_dims(who, B, C, H, W)) return rc;
    if (C & 1) IFL_FAIL(IFL_EINVAL, "%s: C=%d must be even", who, C);
    if (B == 0) return IFL_OK;
    if (!x || !h || !y) IFL_FAIL(IFL_EINVAL, "%s: null pointer", who);
    const bool want_ld = logdet && !reverse;
    if (want_ld && (!ws || ws_bytes < ifl_glow_workspace_bytes(B, C)))
        IFL_FAIL(IFL_EWORKSPACE, "%s: workspace of %zu bytes needed", who, ifl_glow_workspace_bytes(B, C));
    hipStream_t s = (hipStream_t)stream;
    float *partial = want_ld ? (float *)(((uintptr_t)ws + 255) & ~(uintptr_t)255) : nullptr;
    hipLaunchKernelGGL(k_coupling<T>, dim3((unsigned)((size_t)B * (C / 2))), dim3(GS_T), 0, s, x, h, y, partial, C, H * W, reverse);
    if (want_ld) hipLaunchKernelGGL(k_coupling_logdet, dim3((B + 63) / 64), dim3(64), 0, s, partial, logdet, B, C / 2);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}
template <class T>
static int coupling_backward_impl(const char *who, const T *gy, const float *g_logdet, const T *x, const T *h, T *gx, T *gh, int B,
                                  int C, int H, int W, ifl_stream_t stream)
{
    clear_error();
    if (int rc = check_dims(who, B, C, H, W)) return rc;
    if (C & 1) IFL_FAIL(IFL_EINVAL, "%s: C=%d must be even", who, C);
    if (B == 0) return IFL_OK;
    if (!gy || !x || !h || !gx || !gh) IFL_FAIL(IFL_EINVAL, "%s: null pointer", who);
    hipLaunchKernelGGL(k_coupling_bwd<T>, dim3((unsigned)((size_t)B * (C / 2))), dim3(GS_T), 0, (hipStream_t)stream, gy, g_logdet, x,
                       h, gx, gh, C, H * W);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}
} // namespace ifl

extern "C" {

int ifl_squeeze_f32(const float *x, float *y, int B, int C, int H, int W, int reverse, ifl_stream_t stream)
{
    return squeeze_impl("ifl_squeeze_f32", x, y, B, C, H, W, reverse, stream);
}
int ifl_squeeze_bf16(const uint16_t *x, uint16_t *y, int B, int C, int H, int W, int reverse, ifl_stream_t stream)
{
    return squeeze_impl("ifl_squeeze_bf16", x, y, B, C, H, W, reverse, stream);
}

int ifl_coupling_f32(const float *x, const float *h, float *y, float *logdet, int B, int C, int H, int W, int reverse,
                     void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    return coupling_impl("ifl_coupling_f32", x, h, y, logdet, B, C, H, W, reverse, ws, ws_bytes, stream);
}
int ifl_coupling_bf16(const uint16_t *x, const uint16_t *h, uint16_t *y, float *logdet, int B, int C, int H, int W, int reverse,
                      void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    return coupling_impl("ifl_coupling_bf16", x, h, y, logdet, B, C, H, W, reverse, ws, ws_bytes, stream);
}

int ifl_coupling_backward_f32(const float *gy, const float *g_logdet, const float *x, const float *h, float *gx, float *gh,
                              int B, int C, int H, int W, ifl_stream_t stream)
{
    return coupling_backward_impl("ifl_coupling_backward_f32", gy, g_logdet, x, h, gx, gh, B, C, H, W, stream);
}
int ifl_coupling_backward_bf16(const uint16_t *gy, const float *g_logdet, const uint16_t *x, const uint16_t *h, uint16_t *gx,
                               uint16_t *gh, int B, int C, int H, int W, ifl_stream_t stream)
{
    return coupling_backward_impl("ifl_coupling_backward_bf16", gy, g_logdet, x, h, gx, gh, B, C, H, W, stream);
}

} // extern "C"

// =====================================================================================================================
// Activations of the Glow step (inf/layers/activations.py): SmoothLeakyRelu (the CIFAR model) and SplineActivation with
// shared weights (the ImageNet-32 model).  The reference evaluates the spline on (B,C,H,W,n_bins) expansions of its
// 3 n_bins - 1 parameters through some fifty eager kernels; here the knot tables (n_bins + 1 entries each, computed by
// the host layer from the parameters with the reference's formulas) sit in registers and an element is one pass.
// =====================================================================================================================
namespace ifl {

// ---- SmoothLeakyRelu (activations.py:37-54): y = a x + (1 - a) softplus(x), y' = a + (1 - a) sigmoid(x),
//      logdet[b] = sum log y'; reverse by Newton-Raphson with the reference's clamp and iteration count, in registers.
// (hardware exp / log: 1 ulp of the 2^x / log2 units after the base change.  log(1 + e) for a tiny e loses e's low bits,
// an ABSOLUTE error of 6e-8 next to max(x, 0): inside the path's 1e-5 tolerance, 2.5x faster than the libm forms)
__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + __logf(1.0f + __expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

__global__ __launch_bounds__(GS_T) void k_slr(const float *__restrict__ x, float *__restrict__ y, float *__restrict__ partial,
                                              int HW, float alpha, int reverse, int n_iter)
{
    __shared__ float sh[4];
    const size_t plane = blockIdx.x;
    const float *xp = x + plane * HW;
    float *yp = y + plane * HW;
    float acc = 0.f;
    for (int i = threadIdx.x; i < HW; i += GS_T) {
        const float v = xp[i];
        if (!reverse) {
            yp[i] = alpha * v + (1.f - alpha) * softplus_f(v);
            acc += __logf(alpha + (1.f - alpha) * sigmoid_f(v));
        } else { // newton_raphson_inverse (activations.py:27-34): x <- x - (f(x) - y) / max(f'(x), 1e-2), x0 = y
            float t = v;
            for (int it = 0; it < n_iter; ++it) {
                const float fp = fmaxf(alpha + (1.f - alpha) * sigmoid_f(t), 1e-2f);
                t = t - (alpha * t + (1.f - alpha) * softplus_f(t) - v) / fp;
            }
            yp[i] = t;
        }
    }
    if (!reverse && partial) {
        const float s = block_sum(acc, sh);
        if (threadIdx.x == 0) partial[plane] = s;
    }
}
// gx = gy y' + g_logdet[b] y'' / y',  y'' = (1 - a) s (1 - s)
__global__ __launch_bounds__(GS_T) void k_slr_bwd(const float *__restrict__ gy, const float *__restrict__ g_logdet,
                                                  const float *__restrict__ x, float *__restrict__ gx, int C, int HW,
                                                  float alpha)
{
    const size_t plane = blockIdx.x;
    const float gl = g_logdet ? g_logdet[plane / C] : 0.f;
    const float *gp = gy + plane * HW, *xp = x + plane * HW;
    float *op = gx + plane * HW;
    for (int i = threadIdx.x; i < HW; i += GS_T) {
        const float s = sigmoid_f(xp[i]);
        const float d1 = alpha + (1.f - alpha) * s, d2 = (1.f - alpha) * s * (1.f - s);
        op[i] = gp[i] * d1 + gl * d2 / d1;
    }
}
// logdet[b] = sum_c partial[b][c] (channels in order)
__global__ void k_plane_sums(const float *__restrict__ partial, float *__restrict__ logdet, int B, int C)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += partial[(size_t)b * C + c];
    logdet[b] = s;
}

// ---- rational-quadratic spline with linear tails (splines/rational_quadratic.py:20-175), shared knots ---------------
// cw, ch: NB + 1 knot positions (cw[0] = ch[0] = -tail, cw[NB] = ch[NB] = +tail), dv: NB + 1 knot derivatives.
static constexpr int RQ_MAXB = 16;   // shared knots (the reference's 32x32x3 builders use 10 bins: if_glow_cifar.py:23-26)
static constexpr int RQ_PE_MAXB = 8; // one set of knots per element (the MNIST builder uses 5)
struct RqTables {
    float cw[RQ_MAXB + 1], ch[RQ_MAXB + 1], dv[RQ_MAXB + 1];
};

// forward-mode derivative carrier over the 7 quantities an element depends on: x, cw[k], cw[k+1], ch[k], ch[k+1],
// dv[k], dv[k+1]
template <int N> struct Dual {
    float v, d[N];
};
template <int N> __device__ __forceinline__ Dual<N> dvar(float v, int i)
{
    Dual<N> r;
    r.v = v;
#pragma unroll
    for (int k = 0; k < N; ++k) r.d[k] = k == i ? 1.f : 0.f;
    return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator+(const Dual<N> &a, const Dual<N> &b)
{
    Dual<N> r;
    r.v = a.v + b.v;
#pragma unroll
    for (int k = 0; k < N; ++k) r.d[k] = a.d[k] + b.d[k];
    return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N> &a, const Dual<N> &b)
{
    Dual<N> r;
    r.v = a.v - b.v;
#pragma unroll
    for (int k = 0; k < N; ++k) r.d[k] = a.d[k] - b.d[k];
    return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N> &a, const Dual<N> &b)
{
    Dual<N> r;
    r.v = a.v * b.v;
#pragma unroll
    for (int k = 0; k < N; ++k) r.d[k] = a.d[k] * b.v + a.v * b.d[k];
    return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N> &a, const Dual<N> &b)
{
    Dual<N> r;
    const float ib = 1.0f / b.v;
    r.v = a.v * ib;
#pragma unroll
    for (int k = 0; k < N; ++k) r.d[k] = (a.d[k] - r.v * b.d[k]) * ib;
    return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator*(float s, const Dual<N> &a)
{
    Dual<N> r;
    r.v = s * a.v;
#pragma unroll
    for (int k = 0; k < N; ++k) r.d[k] = s * a.d[k];
    return r;
}
template <int N> __device__ __forceinline__ Dual<N> dlog(const Dual<N> &a)
{
    Dual<N> r;
    const float ia = 1.0f / a.v;
    r.v = __logf(a.v);
#pragma unroll
    for (int k = 0; k < N; ++k) r.d[k] = a.d[k] * ia;
    return r;
}
// searchsorted (rational_quadratic.py:12-17): #(knots <= v) - 1, the last knot raised by 1e-6
template <int NB> __device__ __forceinline__ int rq_bin(const float (&knots)[RQ_MAXB + 1], float v)
{
    int k = -1;
#pragma unroll
    for (int j = 0; j <= NB; ++j) k += (v >= (j == NB ? knots[j] + 1e-6f : knots[j])) ? 1 : 0;
    return k < 0 ? 0 : (k > NB - 1 ? NB - 1 : k);
}
template <int NB> __device__ __forceinline__ void rq_pick(const RqTables &t, int k, float &a, float &b, float &c, float &e,
                                                          float &d0, float &d1)
{
    a = b = c = e = d0 = d1 = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j)
        if (j == k) {
            a = t.cw[j];
            b = t.cw[j + 1];
            c = t.ch[j];
            e = t.ch[j + 1];
            d0 = t.dv[j];
            d1 = t.dv[j + 1];
        }
}
// the spline's value and log-derivative (forward direction), generic in the scalar type (float or Dual)
template <class T> __device__ __forceinline__ void rq_eval(const T &x, const T &a, const T &b, const T &c, const T &e,
                                                           const T &d0, const T &d1, T &y, T &lad, T (*logfn)(const T &))
{
    const T w = b - a, h = e - c, delta = h / w, theta = (x - a) / w;
    const T om = (b - x) / w; // 1 - theta
    const T t1 = theta * om;
    const T num = h * (delta * theta * theta + d0 * t1);
    const T den = delta + (d0 + d1 - (delta + delta)) * t1;
    y = c + num / den;
    const T dnum = delta * delta * (d1 * theta * theta + (delta + delta) * t1 + d0 * om * om);
    lad = logfn(dnum) - (logfn(den) + logfn(den));
}
__device__ __forceinline__ float flog(const float &a) { return __logf(a); }

// Knot tables of the shared-weight spline from its parameters (rational_quadratic.py:35-46,97-116), by the first wave of a
// workgroup (every thread of the workgroup must call: two barriers inside); lane i of each half-wave owns bin i of the
// widths / heights, every sum is added by one lane in the serial order (see k_rq_tables below: the same bits).
//   cw_j = 2T cum_j - T,  cum_j = sum_{i<j} (m + (1 - m K) softmax(uw)_i),  cw_0 = -T, cw_K = T   (ch likewise from uh)
//   dv_j = m + softplus(ud_{j-1} + c) for 0 < j < K,  dv_0 = dv_K = m + softplus(c),  c = log(e^{1-m} - 1),  m = 1e-6
__device__ __forceinline__ void rq_tables_compute(const float *__restrict__ uw, const float *__restrict__ uh, const float *__restrict__ ud,
                                                  int K, float T, float *cw, float *ch, float *dv, float (*ex)[32])
{
    const bool w0 = threadIdx.x < 64;
    const int which = (threadIdx.x >> 5) & 1, i = threadIdx.x & 31;
    const float m = 1e-6f, c = logf(expf(1.0f - m) - 1.0f);
    const float *u = which ? uh : uw;
    float *out = which ? ch : cw;
    float mx = 0.f;
    if (w0) {
        mx = u[0];
        for (int t = 1; t < K; ++t) mx = fmaxf(mx, u[t]);
        if (i < K) ex[which][i] = expf(u[i] - mx);
    }
    __syncthreads();
    if (w0) {
        float den = 0.f;
        for (int t = 0; t < K; ++t) den += ex[which][t];
        if (i < K) {
            float cum = 0.f;
            for (int t = 0; t <= i; ++t) cum += m + (1.0f - m * K) * (ex[which][t] / den);
            out[i + 1] = i + 1 == K ? T : 2.0f * T * cum - T;
        }
        if (i == 0) out[0] = -T;
        if (which == 0 && i <= K) {
            const float a = (i == 0 || i == K) ? c : ud[i - 1] + c;
            dv[i] = m + (fmaxf(a, 0.f) + log1pf(expf(-fabsf(a))));
        }
    }
    __syncthreads();
}

template <int NB>
__global__ __launch_bounds__(GS_T) void k_rqspline(const float *__restrict__ x, float *__restrict__ y,
                                                   float *__restrict__ partial, const float *__restrict__ cw,
                                                   const float *__restrict__ ch, const float *__restrict__ dv, int HW,
                                                   float tail, int inverse, const float *__restrict__ uw = nullptr,
                                                   const float *__restrict__ uh = nullptr, const float *__restrict__ ud = nullptr,
                                                   float *__restrict__ tables_out = nullptr)
{
    __shared__ float sh[4];
    __shared__ float tb[3][RQ_MAXB + 1], ex[2][32];
    RqTables t;
    if (uw) { // from the parameters: every workgroup computes the tables itself (a launch less), the first one keeps them
        rq_tables_compute(uw, uh, ud, NB, tail, tb[0], tb[1], tb[2], ex);
#pragma unroll
        for (int j = 0; j <= NB; ++j) {
            t.cw[j] = tb[0][j];
            t.ch[j] = tb[1][j];
            t.dv[j] = tb[2][j];
        }
        if (tables_out && blockIdx.x == 0 && threadIdx.x < 3 * (NB + 1))
            tables_out[threadIdx.x] = tb[threadIdx.x / (NB + 1)][threadIdx.x % (NB + 1)];
    } else { // (wave-uniform loads: the tables live in scalar registers)
#pragma unroll
        for (int j = 0; j <= NB; ++j) {
            t.cw[j] = cw[j];
            t.ch[j] = ch[j];
            t.dv[j] = dv[j];
        }
    }
    const size_t plane = blockIdx.x;
    const float *xp = x + plane * HW;
    float *yp = y + plane * HW;
    float acc = 0.f;
    for (int i = threadIdx.x; i < HW; i += GS_T) {
        const float v = xp[i];
        float out = v, lad = 0.f;
        if (v >= -tail && v <= tail) {
            float a, b, c, e, d0, d1;
            if (!inverse) {
                rq_pick<NB>(t, rq_bin<NB>(t.cw, v), a, b, c, e, d0, d1);
                rq_eval<float>(v, a, b, c, e, d0, d1, out, lad, flog);
            } else { // rational_quadratic.py:132-157
                rq_pick<NB>(t, rq_bin<NB>(t.ch, v), a, b, c, e, d0, d1);
                const float w = b - a, h = e - c, delta = h / w, s = d0 + d1 - 2.f * delta, r = v - c;
                const float qa = r * s + h * (delta - d0), qb = h * d0 - r * s, qc = -delta * r;
                const float root = (2.f * qc) / (-qb - sqrtf(qb * qb - 4.f * qa * qc));
                out = root * w + a;
                const float t1 = root * (1.f - root), den = delta + s * t1;
                const float dnum = delta * delta * (d1 * root * root + 2.f * delta * t1 + d0 * (1.f - root) * (1.f - root));
                lad = -(__logf(dnum) - 2.f * __logf(den));
            }
        }
        yp[i] = out;
        acc += lad;
    }
    if (partial) {
        const float s = block_sum(acc, sh);
        if (threadIdx.x == 0) partial[plane] = s;
    }
}

// backward of the forward direction: gx, and per plane the gradient with respect to the three knot tables
// (tpart[plane][3][NB+1]); forward-mode derivatives of (y, lad) over the 7 quantities of the element's bin
template <int NB>
__global__ __launch_bounds__(GS_T) void k_rqspline_bwd(const float *__restrict__ gy, const float *__restrict__ g_logdet,
                                                       const float *__restrict__ x, float *__restrict__ gx,
                                                       float *__restrict__ tpart, const float *__restrict__ cw,
                                                       const float *__restrict__ ch, const float *__restrict__ dv, int C,
                                                       int HW, float tail)
{
    typedef Dual<7> D;
    RqTables t;
#pragma unroll
    for (int j = 0; j <= NB; ++j) {
        t.cw[j] = cw[j];
        t.ch[j] = ch[j];
        t.dv[j] = dv[j];
    }
    const size_t plane = blockIdx.x;
    const float gl = g_logdet ? g_logdet[plane / C] : 0.f;
    const float *gp = gy + plane * HW, *xp = x + plane * HW;
    float *op = gx + plane * HW;
    float gcw[NB + 1], gch[NB + 1], gdv[NB + 1];
#pragma unroll
    for (int j = 0; j <= NB; ++j) gcw[j] = gch[j] = gdv[j] = 0.f;
    for (int i = threadIdx.x; i < HW; i += GS_T) {
        const float v = xp[i], g = gp[i];
        float gxi = g; // linear tails: y = x, log-derivative 0
        if (v >= -tail && v <= tail) {
            const int k = rq_bin<NB>(t.cw, v);
            float a, b, c, e, d0, d1;
            rq_pick<NB>(t, k, a, b, c, e, d0, d1);
            D yy, ll;
            rq_eval<D>(dvar<7>(v, 0), dvar<7>(a, 1), dvar<7>(b, 2), dvar<7>(c, 3), dvar<7>(e, 4), dvar<7>(d0, 5), dvar<7>(d1, 6), yy, ll,
                       dlog<7>);
            float q[7];
#pragma unroll
            for (int m = 0; m < 7; ++m) q[m] = g * yy.d[m] + gl * ll.d[m];
            gxi = q[0];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float on = j == k ? 1.f : 0.f;
                gcw[j] += on * q[1];
                gcw[j + 1] += on * q[2];
                gch[j] += on * q[3];
                gch[j + 1] += on * q[4];
                gdv[j] += on * q[5];
                gdv[j + 1] += on * q[6];
            }
        }
        op[i] = gxi;
    }
    // one reduction for the 3 (NB + 1) sums: shuffle tree per wave, the four waves' results through LDS, fixed order
    __shared__ float red[4][3 * (NB + 1)];
#pragma unroll
    for (int j = 0; j <= NB; ++j) {
        float s0 = gcw[j], s1 = gch[j], s2 = gdv[j];
        for (int o = 32; o > 0; o >>= 1) {
            s0 += __shfl_down(s0, o, 64);
            s1 += __shfl_down(s1, o, 64);
            s2 += __shfl_down(s2, o, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            red[threadIdx.x >> 6][j] = s0;
            red[threadIdx.x >> 6][(NB + 1) + j] = s1;
            red[threadIdx.x >> 6][2 * (NB + 1) + j] = s2;
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 * (NB + 1))
        tpart[plane * 3 * (NB + 1) + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// out[e] = sum over planes of tpart[plane][e] (one wave per table entry, planes strided over its lanes in order,
// fixed shuffle tree)
__global__ __launch_bounds__(64) void k_table_sums(const float *__restrict__ tpart, float *__restrict__ out, size_t planes, int ne)
{
    const int e = blockIdx.x;
    float s = 0.f;
    for (size_t p = threadIdx.x; p < planes; p += 64) s += tpart[p * ne + e];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (threadIdx.x == 0) out[e] = s;
}

// ---- knot tables of the shared-weight spline from its parameters (rational_quadratic.py:35-46,97-116) and the way back.
// 3 n_bins - 1 numbers in, 3 (n_bins + 1) out; replaces ~25 eager launches each way.  One wave: lane i of each half owns bin i
// of the widths / heights (the exponentials and logarithms run side by side -- as one serial thread these two kernels took
// 9-10 us each, 31-144 times a step); every sum is still added by one lane in the serial code's order, so the results are the
// same bits.
//   cw_j = 2T cum_j - T,  cum_j = sum_{i<j} (m + (1 - m K) softmax(uw)_i),  cw_0 = -T, cw_K = T   (ch likewise from uh)
//   dv_j = m + softplus(ud_{j-1} + c) for 0 < j < K,  dv_0 = dv_K = m + softplus(c),  c = log(e^{1-m} - 1),  m = 1e-6
__global__ __launch_bounds__(64) void k_rq_tables(const float *__restrict__ uw, const float *__restrict__ uh, const float *__restrict__ ud,
                                                  float *__restrict__ cw, float *__restrict__ ch, float *__restrict__ dv, int K, float T)
{
    __shared__ float ex[2][32];
    if (blockIdx.x != 0) return;
    rq_tables_compute(uw, uh, ud, K, T, cw, ch, dv, ex);
}
// g_tables = (g_cw, g_ch, g_dv), 3 (K + 1) floats -> gradients of the parameters
__device__ __forceinline__ void rq_tables_bwd_compute(const float *gt, const float *__restrict__ uw, const float *__restrict__ uh,
                                                      const float *__restrict__ ud, float *__restrict__ guw, float *__restrict__ guh,
                                                      float *__restrict__ gud, int K, float T, float (*pr)[32], float (*gv)[32])
{ // (the first wave of a workgroup; every thread of it must call: two barriers inside)
    const bool w0 = threadIdx.x < 64;
    const int which = (threadIdx.x >> 5) & 1, i = threadIdx.x & 31;
    const float m = 1e-6f, c = logf(expf(1.0f - m) - 1.0f);
    const float *u = which ? uh : uw;
    const float *g = gt + which * (K + 1);
    float *out = which ? guh : guw;
    if (w0) {
        float mx = u[0];
        for (int t = 1; t < K; ++t) mx = fmaxf(mx, u[t]);
        if (i < K) pr[which][i] = expf(u[i] - mx);
    }
    __syncthreads();
    float den = 0.f;
    if (w0) {
        for (int t = 0; t < K; ++t) den += pr[which][t];
        // g_v_i = (1 - m K) 2T sum_{j = i+1}^{K-1} g_knot_j  (the end knots are constants; added from the last knot down)
        if (i < K) {
            float tail = 0.f;
            for (int t = K - 1; t > i; --t) tail += g[t];
            gv[which][i] = (1.0f - m * K) * 2.0f * T * tail;
        }
    }
    __syncthreads();
    if (w0) {
        float dot = 0.f;
        for (int t = 0; t < K; ++t) dot += (pr[which][t] / den) * gv[which][t];
        if (i < K) out[i] = (pr[which][i] / den) * (gv[which][i] - dot);
        const float *gd = gt + 2 * (K + 1);
        if (which == 1 && i >= 1 && i < K) gud[i - 1] = gd[i] / (1.0f + expf(-(ud[i - 1] + c)));
    }
}
__global__ __launch_bounds__(64) void k_rq_tables_bwd(const float *__restrict__ gt, const float *__restrict__ uw,
                                                      const float *__restrict__ uh, const float *__restrict__ ud, float *__restrict__ guw,
                                                      float *__restrict__ guh, float *__restrict__ gud, int K, float T)
{
    __shared__ float pr[2][32], gv[2][32];
    if (blockIdx.x != 0) return;
    rq_tables_bwd_compute(gt, uw, uh, ud, guw, guh, gud, K, T, pr, gv);
}

// ---- the spline with one set of knots per element ("individual weights", activations.py:135-144: parameters of shape
// (1, C, H, W, n_bins): the MNIST Glow's activation).  A thread owns one element position e of the P = C H W and a group of
// images: it builds that element's knot tables in registers (the formulas of k_rq_tables), walks its images, and -- backward
// -- turns the sums of the table gradients into parameter gradients itself (k_rq_tables_bwd's formulas): the three
// parameter tensors are read once per thread and their gradients leave as one partial per image group, summed in a fixed
// order afterwards.  The reference's torch expressions expand three (B, C, H, W, n_bins + 1) tensors and gather along the
// bin axis: ~40 launches forward, ~80 backward.
template <int NB> __device__ __forceinline__ void rq_tables_of(const float *uw, const float *uh, const float *ud, float T, RqTables &t,
                                                               float (&pw)[NB], float (&ph)[NB])
{
    const float m = 1e-6f, c = logf(expf(1.0f - m) - 1.0f);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const float *u = which ? uh : uw;
        float mx = u[0];
#pragma unroll
        for (int i = 1; i < NB; ++i) mx = fmaxf(mx, u[i]);
        float ex[NB], den = 0.f;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            ex[i] = expf(u[i] - mx);
            den += ex[i];
        }
        float cum = 0.f;
        float *out = which ? t.ch : t.cw;
        out[0] = -T;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const float pi = ex[i] / den;
            (which ? ph : pw)[i] = pi; // softmax, kept for the way back
            cum += m + (1.0f - m * NB) * pi;
            out[i + 1] = i + 1 == NB ? T : 2.0f * T * cum - T;
        }
    }
#pragma unroll
    for (int j = 0; j <= NB; ++j) {
        const float a = (j == 0 || j == NB) ? c : ud[j - 1] + c;
        t.dv[j] = m + (fmaxf(a, 0.f) + log1pf(expf(-fabsf(a))));
    }
}

template <int NB>
__global__ __launch_bounds__(64) void k_rqspline_pe(const float *__restrict__ x, const float *__restrict__ uw,
                                                    const float *__restrict__ uh, const float *__restrict__ ud,
                                                    float *__restrict__ y, float *__restrict__ lpart, int B, int P, int BG,
                                                    float tail, int inverse)
{
    const int e = blockIdx.x * 64 + threadIdx.x, b0 = blockIdx.y * BG, b1 = b0 + BG < B ? b0 + BG : B;
    const bool ok = e < P;
    const int ec = ok ? e : P - 1;
    float pu[NB], ph_[NB], pd[NB > 1 ? NB - 1 : 1];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        pu[j] = uw[(size_t)ec * NB + j];
        ph_[j] = uh[(size_t)ec * NB + j];
    }
#pragma unroll
    for (int j = 0; j + 1 < NB; ++j) pd[j] = ud[(size_t)ec * (NB - 1) + j];
    RqTables t;
    float sw[NB], sh_[NB];
    rq_tables_of<NB>(pu, ph_, pd, tail, t, sw, sh_);
    for (int b = b0; b < b1; ++b) {
        float lad = 0.f;
        if (ok) {
            const float v = x[(size_t)b * P + e];
            float out = v;
            if (v >= -tail && v <= tail) {
                float a, bb, c, ee, d0, d1;
                if (!inverse) {
                    rq_pick<NB>(t, rq_bin<NB>(t.cw, v), a, bb, c, ee, d0, d1);
                    rq_eval<float>(v, a, bb, c, ee, d0, d1, out, lad, flog);
                } else {
                    rq_pick<NB>(t, rq_bin<NB>(t.ch, v), a, bb, c, ee, d0, d1);
                    const float w = bb - a, h = ee - c, delta = h / w, s = d0 + d1 - 2.f * delta, r = v - c;
                    const float qa = r * s + h * (delta - d0), qb = h * d0 - r * s, qc = -delta * r;
                    const float root = (2.f * qc) / (-qb - sqrtf(qb * qb - 4.f * qa * qc));
                    out = root * w + a;
                    const float t1 = root * (1.f - root), den = delta + s * t1;
                    const float dnum = delta * delta * (d1 * root * root + 2.f * delta * t1 + d0 * (1.f - root) * (1.f - root));
                    lad = -(__logf(dnum) - 2.f * __logf(den));
                }
            }
            y[(size_t)b * P + e] = out;
        }
        if (lpart) {
            for (int o = 32; o > 0; o >>= 1) lad += __shfl_down(lad, o, 64);
            if (threadIdx.x == 0) lpart[(size_t)b * gridDim.x + blockIdx.x] = lad;
        }
    }
}

// gpart[group][uw: P NB | uh: P NB | ud: P (NB - 1)]
template <int NB>
__global__ __launch_bounds__(64) void k_rqspline_pe_bwd(const float *__restrict__ gy, const float *__restrict__ g_logdet,
                                                        const float *__restrict__ x, const float *__restrict__ uw,
                                                        const float *__restrict__ uh, const float *__restrict__ ud,
                                                        float *__restrict__ gx, float *__restrict__ gpart, int B, int P, int BG,
                                                        float tail)
{
    typedef Dual<7> D;
    const int e = blockIdx.x * 64 + threadIdx.x, b0 = blockIdx.y * BG, b1 = b0 + BG < B ? b0 + BG : B;
    if (e >= P) return;
    float pu[NB], ph_[NB], pd[NB > 1 ? NB - 1 : 1];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        pu[j] = uw[(size_t)e * NB + j];
        ph_[j] = uh[(size_t)e * NB + j];
    }
#pragma unroll
    for (int j = 0; j + 1 < NB; ++j) pd[j] = ud[(size_t)e * (NB - 1) + j];
    RqTables t;
    float sw[NB], sh_[NB];
    rq_tables_of<NB>(pu, ph_, pd, tail, t, sw, sh_);
    float gcw[NB + 1], gch[NB + 1], gdv[NB + 1];
#pragma unroll
    for (int j = 0; j <= NB; ++j) gcw[j] = gch[j] = gdv[j] = 0.f;
    for (int b = b0; b < b1; ++b) {
        const float v = x[(size_t)b * P + e], g = gy[(size_t)b * P + e], gl = g_logdet ? g_logdet[b] : 0.f;
        float gxi = g;
        if (v >= -tail && v <= tail) {
            const int k = rq_bin<NB>(t.cw, v);
            float a, bb, c, ee, d0, d1;
            rq_pick<NB>(t, k, a, bb, c, ee, d0, d1);
            D yy, ll;
            rq_eval<D>(dvar<7>(v, 0), dvar<7>(a, 1), dvar<7>(bb, 2), dvar<7>(c, 3), dvar<7>(ee, 4), dvar<7>(d0, 5), dvar<7>(d1, 6), yy, ll,
                       dlog<7>);
            float q[7];
#pragma unroll
            for (int m = 0; m < 7; ++m) q[m] = g * yy.d[m] + gl * ll.d[m];
            gxi = q[0];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float on = j == k ? 1.f : 0.f;
                gcw[j] += on * q[1];
                gcw[j + 1] += on * q[2];
                gch[j] += on * q[3];
                gch[j + 1] += on * q[4];
                gdv[j] += on * q[5];
                gdv[j + 1] += on * q[6];
            }
        }
        gx[(size_t)b * P + e] = gxi;
    }
    // tables -> parameters (k_rq_tables_bwd): knot j of a cumulative table is fed by the softmax entries 0 .. j - 1
    const float m = 1e-6f, c0 = logf(expf(1.0f - m) - 1.0f);
    float *gp = gpart + (size_t)blockIdx.y * P * (3 * NB - 1);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const float *gk = which ? gch : gcw;
        const float *pi = which ? sh_ : sw;
        float gv[NB], tailsum = 0.f, dot = 0.f;
#pragma unroll
        for (int i = NB - 1; i >= 0; --i) {
            gv[i] = (1.0f - m * NB) * 2.0f * tail * tailsum;
            if (i >= 1) tailsum += gk[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) dot += pi[i] * gv[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) gp[(size_t)which * P * NB + (size_t)e * NB + i] = pi[i] * (gv[i] - dot);
    }
#pragma unroll
    for (int j = 1; j < NB; ++j) gp[(size_t)2 * P * NB + (size_t)e * (NB - 1) + j - 1] = gdv[j] / (1.0f + expf(-(pd[j - 1] + c0)));
}
// out[i] = sum over groups of part[g][i], in group order
__global__ __launch_bounds__(256) void k_group_sums(const float *__restrict__ part, float *__restrict__ out, int G, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += part[(size_t)g * n + i];
    out[i] = s;
}
static int rq_pe_groups(int B) { return B < 4 ? 1 : (B + 3) / 4; } // four images per thread

static int rq_check(const char *who, const float *cw, const float *ch, const float *dv, int nb)
{
    if (nb < 1 || nb > RQ_MAXB) IFL_FAIL(IFL_EUNSUPPORTED, "%s: n_bins=%d (1..%d supported)", who, nb, RQ_MAXB);
    if (!cw || !ch || !dv) IFL_FAIL(IFL_EINVAL, "%s: null knot table", who);
    return IFL_OK;
}

} // namespace ifl

extern "C" {

size_t ifl_activation_workspace_bytes(int B, int C, int n_bins)
{
    const size_t planes = (size_t)(B > 0 ? B : 0) * (C > 0 ? C : 0);
    const size_t row = 3 * (size_t)(n_bins > 0 ? n_bins + 1 : 1);
    return (planes * (row + 1) + row) * sizeof(float) + 256; // (per-plane table partials, per-plane sums, one row of totals)
}

int ifl_slr_f32(const float *x, float *y, float *logdet, int B, int C, int H, int W, float alpha, int reverse, void *ws,
                size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    if (int rc = check_dims("ifl_slr_f32", B, C, H, W)) return rc;
    if (B == 0) return IFL_OK;
    if (!x || !y) IFL_FAIL(IFL_EINVAL, "ifl_slr_f32: null pointer");
    const bool want_ld = logdet && !reverse;
    if (want_ld && (!ws || ws_bytes < ifl_activation_workspace_bytes(B, C, 0)))
        IFL_FAIL(IFL_EWORKSPACE, "ifl_slr_f32: workspace of %zu bytes needed", ifl_activation_workspace_bytes(B, C, 0));
    hipStream_t s = (hipStream_t)stream;
    float *partial = want_ld ? (float *)(((uintptr_t)ws + 255) & ~(uintptr_t)255) : nullptr;
    hipLaunchKernelGGL(k_slr, dim3((unsigned)((size_t)B * C)), dim3(GS_T), 0, s, x, y, partial, H * W, alpha, reverse, 100);
    if (want_ld) hipLaunchKernelGGL(k_plane_sums, dim3((B + 63) / 64), dim3(64), 0, s, partial, logdet, B, C);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

int ifl_slr_backward_f32(const float *gy, const float *g_logdet, const float *x, float *gx, int B, int C, int H, int W,
                         float alpha, ifl_stream_t stream)
{
    clear_error();
    if (int rc = check_dims("ifl_slr_backward_f32", B, C, H, W)) return rc;
    if (B == 0) return IFL_OK;
    if (!gy || !x || !gx) IFL_FAIL(IFL_EINVAL, "ifl_slr_backward_f32: null pointer");
    hipLaunchKernelGGL(k_slr_bwd, dim3((unsigned)((size_t)B * C)), dim3(GS_T), 0, (hipStream_t)stream, gy, g_logdet, x, gx, C, H * W,
                       alpha);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

/* parameters (DEVICE: n_bins, n_bins, n_bins - 1 floats) -> knot tables (DEVICE: n_bins + 1 floats each) */
int ifl_rqspline_tables_f32(const float *uw, const float *uh, const float *ud, int n_bins, float tail_bound, float *cw,
                            float *ch, float *dv, ifl_stream_t stream)
{
    clear_error();
    if (n_bins < 1 || n_bins > RQ_MAXB) IFL_FAIL(IFL_EUNSUPPORTED, "ifl_rqspline_tables_f32: n_bins=%d (1..%d supported)", n_bins, RQ_MAXB);
    if (!uw || !uh || (!ud && n_bins > 1) || !cw || !ch || !dv) IFL_FAIL(IFL_EINVAL, "ifl_rqspline_tables_f32: null pointer");
    hipLaunchKernelGGL(k_rq_tables, dim3(1), dim3(64), 0, (hipStream_t)stream, uw, uh, ud, cw, ch, dv, n_bins, tail_bound);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

/* g_tables (DEVICE, 3 (n_bins + 1) floats: d loss / d cw, ch, dv) -> gradients of the three parameter vectors */
int ifl_rqspline_tables_backward_f32(const float *g_tables, const float *uw, const float *uh, const float *ud, int n_bins,
                                     float tail_bound, float *g_uw, float *g_uh, float *g_ud, ifl_stream_t stream)
{
    clear_error();
    if (n_bins < 1 || n_bins > RQ_MAXB)
        IFL_FAIL(IFL_EUNSUPPORTED, "ifl_rqspline_tables_backward_f32: n_bins=%d (1..%d supported)", n_bins, RQ_MAXB);
    if (!g_tables || !uw || !uh || (!ud && n_bins > 1) || !g_uw || !g_uh || (!g_ud && n_bins > 1))
        IFL_FAIL(IFL_EINVAL, "ifl_rqspline_tables_backward_f32: null pointer");
    hipLaunchKernelGGL(k_rq_tables_bwd, dim3(1), dim3(64), 0, (hipStream_t)stream, g_tables, uw, uh, ud, g_uw, g_uh, g_ud, n_bins,
                       tail_bound);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

/* cw, ch, dv: DEVICE arrays of n_bins + 1 floats (the knot tables: no host round trip between the layer's parameters
 * and the launch) */
int ifl_rqspline_f32(const float *x, const float *cw, const float *ch, const float *dv, int n_bins, float tail_bound, float *y,
                     float *logdet, int B, int C, int H, int W, int inverse, void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    if (int rc = check_dims("ifl_rqspline_f32", B, C, H, W)) return rc;
    if (int rc = rq_check("ifl_rqspline_f32", cw, ch, dv, n_bins)) return rc;
    if (B == 0) return IFL_OK;
    if (!x || !y) IFL_FAIL(IFL_EINVAL, "ifl_rqspline_f32: null pointer");
    if (logdet && (!ws || ws_bytes < ifl_activation_workspace_bytes(B, C, 0)))
        IFL_FAIL(IFL_EWORKSPACE, "ifl_rqspline_f32: workspace of %zu bytes needed", ifl_activation_workspace_bytes(B, C, 0));
    hipStream_t s = (hipStream_t)stream;
    float *partial = logdet ? (float *)(((uintptr_t)ws + 255) & ~(uintptr_t)255) : nullptr;
    const dim3 grid((unsigned)((size_t)B * C));
#define IFL_RQ(NB) \
    case NB: hipLaunchKernelGGL(k_rqspline<NB>, grid, dim3(GS_T), 0, s, x, y, partial, cw, ch, dv, H * W, tail_bound, inverse); break;
    switch (n_bins) {
        IFL_RQ(1) IFL_RQ(2) IFL_RQ(3) IFL_RQ(4) IFL_RQ(5) IFL_RQ(6) IFL_RQ(7) IFL_RQ(8)
        IFL_RQ(9) IFL_RQ(10) IFL_RQ(11) IFL_RQ(12) IFL_RQ(13) IFL_RQ(14) IFL_RQ(15) IFL_RQ(16)
    }
#undef IFL_RQ
    if (logdet) hipLaunchKernelGGL(k_plane_sums, dim3((B + 63) / 64), dim3(64), 0, s, partial, logdet, B, C);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

/* g_tables: DEVICE array of 3 (n_bins + 1) floats: d loss / d cw, d ch, d dv */
int ifl_rqspline_backward_f32(const float *gy, const float *g_logdet, const float *x, const float *cw, const float *ch,
                              const float *dv, int n_bins, float tail_bound, float *gx, float *g_tables, int B, int C, int H,
                              int W, void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    if (int rc = check_dims("ifl_rqspline_backward_f32", B, C, H, W)) return rc;
    if (int rc = rq_check("ifl_rqspline_backward_f32", cw, ch, dv, n_bins)) return rc;
    if (!gy || !x || !gx || !g_tables) IFL_FAIL(IFL_EINVAL, "ifl_rqspline_backward_f32: null pointer");
    if (!ws || ws_bytes < ifl_activation_workspace_bytes(B, C, n_bins))
        IFL_FAIL(IFL_EWORKSPACE, "ifl_rqspline_backward_f32: workspace of %zu bytes needed", ifl_activation_workspace_bytes(B, C, n_bins));
    hipStream_t s = (hipStream_t)stream;
    float *tpart = (float *)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    const size_t planes = (size_t)B * C;
    if (planes) {
        const dim3 grid((unsigned)planes);
#define IFL_RQ(NB) \
    case NB: hipLaunchKernelGGL(k_rqspline_bwd<NB>, grid, dim3(GS_T), 0, s, gy, g_logdet, x, gx, tpart, cw, ch, dv, C, H * W, tail_bound); break;
        switch (n_bins) {
            IFL_RQ(1) IFL_RQ(2) IFL_RQ(3) IFL_RQ(4) IFL_RQ(5) IFL_RQ(6) IFL_RQ(7) IFL_RQ(8)
            IFL_RQ(9) IFL_RQ(10) IFL_RQ(11) IFL_RQ(12) IFL_RQ(13) IFL_RQ(14) IFL_RQ(15) IFL_RQ(16)
        }
#undef IFL_RQ
    }
    hipLaunchKernelGGL(k_table_sums, dim3(3 * (n_bins + 1)), dim3(64), 0, s, tpart, g_tables, planes, 3 * (n_bins + 1));
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

/* the shared-weight spline straight from its parameters: the tables are computed inside the launch (and kept in `tables`:
 * 3 (n_bins + 1) floats, for the backward); one launch less each way than tables + spline */
int ifl_rqspline_p_f32(const float *x, const float *uw, const float *uh, const float *ud, int n_bins, float tail_bound, float *y,
                       float *logdet, float *tables, int B, int C, int H, int W, int inverse, void *ws, size_t ws_bytes,
                       ifl_stream_t stream)
{
    clear_error();
    if (int rc = check_dims("ifl_rqspline_p_f32", B, C, H, W)) return rc;
    if (n_bins < 1 || n_bins > RQ_MAXB) IFL_FAIL(IFL_EUNSUPPORTED, "ifl_rqspline_p_f32: n_bins=%d (1..%d supported)", n_bins, RQ_MAXB);
    if (!uw || !uh || (!ud && n_bins > 1)) IFL_FAIL(IFL_EINVAL, "ifl_rqspline_p_f32: null pointer");
    if (B == 0) return IFL_OK;
    if (!x || !y) IFL_FAIL(IFL_EINVAL, "ifl_rqspline_p_f32: null pointer");
    if (logdet && (!ws || ws_bytes < ifl_activation_workspace_bytes(B, C, 0)))
        IFL_FAIL(IFL_EWORKSPACE, "ifl_rqspline_p_f32: workspace of %zu bytes needed", ifl_activation_workspace_bytes(B, C, 0));
    hipStream_t s = (hipStream_t)stream;
    float *partial = logdet ? (float *)(((uintptr_t)ws + 255) & ~(uintptr_t)255) : nullptr;
    const dim3 grid((unsigned)((size_t)B * C));
#define IFL_RQ(NB)                                                                                                              \
    case NB:                                                                                                                    \
        hipLaunchKernelGGL(k_rqspline<NB>, grid, dim3(GS_T), 0, s, x, y, partial, (const float *)nullptr, (const float *)nullptr, \
                           (const float *)nullptr, H * W, tail_bound, inverse, uw, uh, ud, tables);                             \
        break;
    switch (n_bins) {
        IFL_RQ(1) IFL_RQ(2) IFL_RQ(3) IFL_RQ(4) IFL_RQ(5) IFL_RQ(6) IFL_RQ(7) IFL_RQ(8)
        IFL_RQ(9) IFL_RQ(10) IFL_RQ(11) IFL_RQ(12) IFL_RQ(13) IFL_RQ(14) IFL_RQ(15) IFL_RQ(16)
    }
#undef IFL_RQ
    if (logdet) hipLaunchKernelGGL(k_plane_sums, dim3((B + 63) / 64), dim3(64), 0, s, partial, logdet, B, C);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

/* its backward: gx and the gradients of the three parameter vectors (tables: what the forward kept) */
int ifl_rqspline_p_backward_f32(const float *gy, const float *g_logdet, const float *x, const float *tables, const float *uw,
                                const float *uh, const float *ud, int n_bins, float tail_bound, float *gx, float *g_uw, float *g_uh,
                                float *g_ud, int B, int C, int H, int W, void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    if (int rc = check_dims("ifl_rqspline_p_backward_f32", B, C, H, W)) return rc;
    if (n_bins < 1 || n_bins > RQ_MAXB) IFL_FAIL(IFL_EUNSUPPORTED, "ifl_rqspline_p_backward_f32: n_bins=%d (1..%d supported)", n_bins, RQ_MAXB);
    if (!gy || !x || !gx || !tables || !uw || !uh || (!ud && n_bins > 1) || !g_uw || !g_uh || (!g_ud && n_bins > 1))
        IFL_FAIL(IFL_EINVAL, "ifl_rqspline_p_backward_f32: null pointer");
    if (!ws || ws_bytes < ifl_activation_workspace_bytes(B, C, n_bins))
        IFL_FAIL(IFL_EWORKSPACE, "ifl_rqspline_p_backward_f32: workspace of %zu bytes needed", ifl_activation_workspace_bytes(B, C, n_bins));
    hipStream_t s = (hipStream_t)stream;
    float *tpart = (float *)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    const size_t planes = (size_t)B * C;
    const float *cw = tables, *ch = tables + (n_bins + 1), *dv = tables + 2 * (n_bins + 1);
    if (planes) {
        const dim3 grid((unsigned)planes);
#define IFL_RQ(NB) \
    case NB: hipLaunchKernelGGL(k_rqspline_bwd<NB>, grid, dim3(GS_T), 0, s, gy, g_logdet, x, gx, tpart, cw, ch, dv, C, H * W, tail_bound); break;
        switch (n_bins) {
            IFL_RQ(1) IFL_RQ(2) IFL_RQ(3) IFL_RQ(4) IFL_RQ(5) IFL_RQ(6) IFL_RQ(7) IFL_RQ(8)
            IFL_RQ(9) IFL_RQ(10) IFL_RQ(11) IFL_RQ(12) IFL_RQ(13) IFL_RQ(14) IFL_RQ(15) IFL_RQ(16)
        }
#undef IFL_RQ
    }
    // (the table sums and their chain to the parameters stay two launches: merged into one workgroup they took 11 us against
    // 4.8 + 4.7)
    float *gt = tpart + planes * 3 * (n_bins + 1); // (the workspace's row of totals)
    hipLaunchKernelGGL(k_table_sums, dim3(3 * (n_bins + 1)), dim3(64), 0, s, tpart, gt, planes, 3 * (n_bins + 1));
    hipLaunchKernelGGL(k_rq_tables_bwd, dim3(1), dim3(64), 0, s, (const float *)gt, uw, uh, ud, g_uw, g_uh, g_ud, n_bins, tail_bound);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

size_t ifl_rqspline_pe_workspace_bytes(int B, int P, int n_bins)
{
    if (B < 0 || P < 1 || n_bins < 1) return 0;
    const size_t eb = ((size_t)P + 63) / 64;
    return (size_t)B * eb * sizeof(float) + (size_t)rq_pe_groups(B) * P * (3 * (size_t)n_bins - 1) * sizeof(float) + 512;
}

int ifl_rqspline_pe_f32(const float *x, const float *uw, const float *uh, const float *ud, int n_bins, float tail_bound, float *y,
                        float *logdet, int B, int P, int inverse, void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    if (B < 0 || P < 1) IFL_FAIL(IFL_EINVAL, "ifl_rqspline_pe_f32: bad shape B=%d P=%d", B, P);
    if (n_bins < 1 || n_bins > RQ_PE_MAXB) IFL_FAIL(IFL_EUNSUPPORTED, "ifl_rqspline_pe_f32: n_bins=%d (1..%d supported)", n_bins, RQ_PE_MAXB);
    if (B == 0) return IFL_OK;
    if (!x || !y || !uw || !uh || (!ud && n_bins > 1)) IFL_FAIL(IFL_EINVAL, "ifl_rqspline_pe_f32: null pointer");
    if (logdet && (!ws || ws_bytes < ifl_rqspline_pe_workspace_bytes(B, P, n_bins)))
        IFL_FAIL(IFL_EWORKSPACE, "ifl_rqspline_pe_f32: workspace of %zu bytes needed", ifl_rqspline_pe_workspace_bytes(B, P, n_bins));
    hipStream_t s = (hipStream_t)stream;
    const int eb = (P + 63) / 64, G = rq_pe_groups(B), BG = (B + G - 1) / G;
    float *lpart = logdet ? (float *)(((uintptr_t)ws + 255) & ~(uintptr_t)255) : nullptr;
    const dim3 grid(eb, G);
    switch (n_bins) {
#define IFL_RQ(NB) \
    case NB: hipLaunchKernelGGL(k_rqspline_pe<NB>, grid, dim3(64), 0, s, x, uw, uh, ud, y, lpart, B, P, BG, tail_bound, inverse); break;
        IFL_RQ(1) IFL_RQ(2) IFL_RQ(3) IFL_RQ(4) IFL_RQ(5) IFL_RQ(6) IFL_RQ(7) IFL_RQ(8)
#undef IFL_RQ
    }
    if (logdet) hipLaunchKernelGGL(k_plane_sums, dim3((B + 63) / 64), dim3(64), 0, s, lpart, logdet, B, eb);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

int ifl_rqspline_pe_backward_f32(const float *gy, const float *g_logdet, const float *x, const float *uw, const float *uh,
                                 const float *ud, int n_bins, float tail_bound, float *gx, float *g_params, int B, int P, void *ws,
                                 size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    if (B < 0 || P < 1) IFL_FAIL(IFL_EINVAL, "ifl_rqspline_pe_backward_f32: bad shape B=%d P=%d", B, P);
    if (n_bins < 1 || n_bins > RQ_PE_MAXB)
        IFL_FAIL(IFL_EUNSUPPORTED, "ifl_rqspline_pe_backward_f32: n_bins=%d (1..%d supported)", n_bins, RQ_PE_MAXB);
    if (!g_params) IFL_FAIL(IFL_EINVAL, "ifl_rqspline_pe_backward_f32: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)P * (3 * (size_t)n_bins - 1);
    if (B == 0) {
        IFL_HIP(hipMemsetAsync(g_params, 0, n * sizeof(float), s));
        return IFL_OK;
    }
    if (!gy || !x || !gx || !uw || !uh || (!ud && n_bins > 1)) IFL_FAIL(IFL_EINVAL, "ifl_rqspline_pe_backward_f32: null pointer");
    if (!ws || ws_bytes < ifl_rqspline_pe_workspace_bytes(B, P, n_bins))
        IFL_FAIL(IFL_EWORKSPACE, "ifl_rqspline_pe_backward_f32: workspace of %zu bytes needed", ifl_rqspline_pe_workspace_bytes(B, P, n_bins));
    const int eb = (P + 63) / 64, G = rq_pe_groups(B), BG = (B + G - 1) / G;
    float *gpart = (float *)((((uintptr_t)ws + 255) & ~(uintptr_t)255) + (((size_t)B * eb * sizeof(float) + 255) & ~(size_t)255));
    const dim3 grid(eb, G);
    switch (n_bins) {
#define IFL_RQ(NB) \
    case NB: hipLaunchKernelGGL(k_rqspline_pe_bwd<NB>, grid, dim3(64), 0, s, gy, g_logdet, x, uw, uh, ud, gx, gpart, B, P, BG, tail_bound); break;
        IFL_RQ(1) IFL_RQ(2) IFL_RQ(3) IFL_RQ(4) IFL_RQ(5) IFL_RQ(6) IFL_RQ(7) IFL_RQ(8)
#undef IFL_RQ
    }
    hipLaunchKernelGGL(k_group_sums, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gpart, g_params, G, n);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

} // extern "C"

// =====================================================================================================================
// Adam / AdamW over ONE flat parameter buffer (the train step's gradient bucket has the same layout: data_parallel.GradBucket).
// torch's fused multi-tensor Adam takes at most 36 tensors a launch (its kernel-argument block): a model of 1 300 small
// tensors is 45 launches of ~48 us, each a handful of latency-bound workgroups -- 2.2 ms of the configs[4] step for 235 MB
// of traffic.  Flat, it is one elementwise pass.  The arithmetic of torch.optim.Adam (step_size = lr / (1 - b1^t),
// denom = sqrt(v) / sqrt(1 - b2^t) + eps), lr and the step count read from device memory (a captured step sees changes).
// =====================================================================================================================
namespace ifl {
__global__ __launch_bounds__(256) void k_adam_flat(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, size_t n, const float *__restrict__ lr_p,
                                                   const float *__restrict__ step_p, float b1, float b2, float eps, float wd,
                                                   int decoupled)
{
    const float lr = *lr_p, t = *step_p;
    const float bc1 = 1.0f - powf(b1, t), bc2s = sqrtf(1.0f - powf(b2, t));
    const float step_size = lr / bc1;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float pi = p[i], gi = g[i];
        if (wd != 0.f) {
            if (decoupled) pi *= 1.0f - lr * wd;
            else gi = fmaf(wd, pi, gi);
        }
        const float mi = fmaf(1.0f - b1, gi - m[i], m[i]); // lerp(m, g, 1 - b1)
        const float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * (mi / (sqrtf(vi) / bc2s + eps));
    }
}
} // namespace ifl

extern "C" int ifl_adam_flat_f32(float *p, const float *g, float *m, float *v, size_t n, const float *lr, const float *step, float beta1,
                                 float beta2, float eps, float weight_decay, int decoupled, ifl_stream_t stream)
{
    ifl::clear_error();
    if (n == 0) return IFL_OK;
    if (!p || !g || !m || !v || !lr || !step) IFL_FAIL(IFL_EINVAL, "ifl_adam_flat_f32: null pointer");
    size_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(ifl::k_adam_flat, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, step, beta1, beta2,
                       eps, weight_decay, decoupled);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

