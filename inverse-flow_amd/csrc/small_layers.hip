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

// The same sweep with the kernel size and the channel count as template arguments (2x2 / 3x3, C <= 8: every layer of the
// reference's MNIST models): the taps unrolled, a thread's column of every folded tap in registers, no integer division
// in the sweep (the generic kernel above spends 1.3 us per anti-diagonal, most of it in t / KW, t % KW and it % C), and z
// kept with a zero halo of the kernel's reach above and to the left, so that no read is conditional.
//   LDS: xs [C][H][W], zs [C][H + KH - 1][W + KW - 1], ws [NT][C][C]
template <int KH, int KW, int C>
__global__ __launch_bounds__(256) void k_scan_resident_t(const float *__restrict__ xin, const float *__restrict__ wf,
                                                         float *__restrict__ zout, Geom g, int rh, int rw)
{
    extern __shared__ float smem[];
    constexpr int NT = KH * KW, CP = C <= 1 ? 1 : (C <= 2 ? 2 : (C <= 4 ? 4 : 8)), LCP = CP == 1 ? 0 : (CP == 2 ? 1 : (CP == 4 ? 2 : 3));
    const int H = g.H, W = g.W, HW = H * W, n = C * HW;
    const int HP = H + KH - 1, WP = W + KW - 1, HWP = HP * WP;
    float *xs = smem;
    float *zs = xs + n;
    float *ws = zs + C * HWP;
    const int b = blockIdx.x, tid = threadIdx.x;
    // element i -> (c, h, w) by two float multiplications (exact: the quotient's rounding error, about (i / W) 2^-23, stays far
    // below the 0.5 / W that (i + 0.5) / W keeps from an integer as long as i < 2^21; here i < 2^13); four loads in flight
    // per thread before the first LDS write
    const float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H;
    auto decode = [&](int i, int &cc, int &h, int &w) {
        const int r = (int)(((float)i + 0.5f) * inv_w);
        w = i - r * W;
        cc = (int)(((float)r + 0.5f) * inv_h);
        h = r - cc * H;
    };
    for (int i0 = tid; i0 < n; i0 += 1024) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            int cc, h, w;
            decode(i < n ? i : 0, cc, h, w);
            v[u] = xin[stored_addr(b, cc, h, w, g, rh, rw)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + 256 * u < n) xs[i0 + 256 * u] = v[u];
    }
    for (int i = tid; i < C * HWP; i += 256) zs[i] = 0.f;
    for (int i = tid; i < NT * C * C; i += 256) ws[i] = wf[i];
    __syncthreads();
    // this thread's output channel and its column of the folded taps
    const int c = tid & (CP - 1), j0 = tid >> LCP;
    float wr[NT][C];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int kc = 0; kc < C; ++kc) wr[t][kc] = c < C ? ws[(t * C + kc) * C + c] : 0.f;
    const int ND = H + W - 1;
    for (int d = 0; d < ND; ++d) {
        const int hmin = d - (W - 1) > 0 ? d - (W - 1) : 0;
        const int hmax = d < H - 1 ? d : H - 1;
        const int nh = hmax - hmin + 1;
        if (c < C) {
            for (int j = j0; j < nh; j += 256 / CP) {
                const int h = hmin + j, w = d - h;
                const float *xp = xs + h * W + w;
                const float *zp = zs + (h + KH - 1) * WP + (w + KW - 1);
                float acc = 0.f;
#pragma unroll
                for (int kc = 0; kc < C; ++kc) acc = fmaf(wr[0][kc], xp[kc * HW], acc);
#pragma unroll
                for (int t = 1; t < NT; ++t)
#pragma unroll
                    for (int kc = 0; kc < C; ++kc) acc = fmaf(-wr[t][kc], zp[kc * HWP - (t / KW) * WP - (t % KW)], acc);
                zs[c * HWP + (h + KH - 1) * WP + (w + KW - 1)] = acc;
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += 256) {
        int cc, h, w;
        decode(i, cc, h, w);
        zout[stored_addr(b, cc, h, w, g, rh, rw)] = zs[cc * HWP + (h + KH - 1) * WP + (w + KW - 1)];
    }
}

static bool scan_resident_templated(const Geom &g)
{
    return g.C >= 1 && g.C <= 8 && ((g.KH == 2 && g.KW == 2) || (g.KH == 3 && g.KW == 3));
}

size_t scan_resident_lds_bytes(const Geom &g)
{
    const size_t zplane = scan_resident_templated(g) ? (size_t)(g.H + g.KH - 1) * (g.W + g.KW - 1) : (size_t)g.H * g.W;
    return ((size_t)g.C * g.H * g.W + (size_t)g.C * zplane + (size_t)g.KH * g.KW * g.C * g.C) * sizeof(float);
}

bool scan_resident_supported(const Geom &g) { return g.C <= 8 && scan_resident_lds_bytes(g) <= 64 * 1024; }

template <int KH, int KW> static void launch_resident_k(const float *x, const float *wf, float *z, const Geom &g, int rh, int rw,
                                                        size_t lds, hipStream_t s)
{
#define IFL_RES(CC) \
    case CC: hipLaunchKernelGGL((k_scan_resident_t<KH, KW, CC>), dim3(g.B), dim3(256), lds, s, x, wf, z, g, rh, rw); break;
    switch (g.C) {
        IFL_RES(1) IFL_RES(2) IFL_RES(3) IFL_RES(4) IFL_RES(5) IFL_RES(6) IFL_RES(7) IFL_RES(8)
    }
#undef IFL_RES
}

int launch_scan_resident(const float *x, const float *wf, float *z, const Geom &g, int rh, int rw, hipStream_t s)
{
    const size_t lds = scan_resident_lds_bytes(g);
    if (scan_resident_templated(g)) {
        if (g.KH == 2) launch_resident_k<2, 2>(x, wf, z, g, rh, rw, lds, s);
        else launch_resident_k<3, 3>(x, wf, z, g, rh, rw, lds, s);
    } else {
        hipLaunchKernelGGL(k_scan_resident, dim3(g.B), dim3(256), lds, s, x, wf, z, g, rh, rw);
    }
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

// partial[b][co][ci][t] = sum_{oh,ow} gz[b][co][oh][ow] * x[b][ci][oh-pt+kh][ow-pl+kw]   (same-size layers)
// One workgroup per image, both tensors of the image in LDS: gz as [co][H*W], x with a zero halo of the kernel's reach
// as [ci][(H+KH-1)(W+KW-1)] (no bounds checks in the loop); plane pitches odd, so that the threads of a wave -- same
// co, consecutive ci -- read one broadcast word of gz and conflict-free words of x.  A thread owns (co, ci) pairs and
// keeps the KH*KW sums of a pair in registers.
template <int KH, int KW>
__global__ __launch_bounds__(256) void k_wgrad_batchpar(const float *__restrict__ gz, const float *__restrict__ x,
                                                        float *__restrict__ partial, int C, int H, int W, int pt, int pl,
                                                        int nsplit)
{
    extern __shared__ float smem[];
    constexpr int NT = KH * KW;
    const int HW = H * W, PG = HW | 1, WH = W + KW - 1, HH = H + KH - 1, PX = (HH * WH) | 1;
    float *gs = smem, *xs = smem + C * PG;
    // blockIdx.x = image * nsplit + part: a part sums the output rows [r0, r1) of its image (small batches: more
    // workgroups than images); its slab is partial[blockIdx.x]
    const int b = blockIdx.x / nsplit, part = blockIdx.x % nsplit, tid = threadIdx.x;
    const int r0 = part * H / nsplit, r1 = (part + 1) * H / nsplit, HR = r1 - r0;
    for (int i = tid; i < C * PX; i += 256) xs[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < C * HW; i += 256) {
        const int c = i / HW, r = i % HW, h = r / W, w = r % W;
        gs[c * PG + r] = gz[(size_t)b * C * HW + i];
        // x[h][w] is read by output (oh, ow) through tap (kh, kw) at h = oh - pt + kh: stored at (h + pt, w + pl)
        xs[c * PX + (h + pt) * WH + (w + pl)] = x[(size_t)b * C * HW + i];
    }
    __syncthreads();
    // fewer pairs than threads: the image rows are cut into S segments per pair, summed afterwards in order
    const int CC = C * C, S = CC >= 256 ? 1 : (256 / CC < HR ? 256 / CC : (HR > 0 ? HR : 1));
    float *red = xs + C * PX; // [pair][segment][tap], only when S > 1
    for (int it = tid; it < CC * S; it += 256) {
        const int pr = it / S, sg = it % S;
        const int co = pr / C, ci = pr % C;
        const int oh0 = r0 + sg * HR / S, oh1 = r0 + (sg + 1) * HR / S;
        const float *gp = gs + co * PG, *xp = xs + ci * PX;
        float acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = 0.f;
        for (int oh = oh0; oh < oh1; ++oh)
            for (int ow = 0; ow < W; ++ow) {
                const float g = gp[oh * W + ow];
#pragma unroll
                for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                    for (int kw = 0; kw < KW; ++kw) acc[kh * KW + kw] = fmaf(g, xp[(oh + kh) * WH + ow + kw], acc[kh * KW + kw]);
            }
        float *out = S > 1 ? red + (size_t)it * NT : partial + ((size_t)blockIdx.x * CC + pr) * NT;
#pragma unroll
        for (int t = 0; t < NT; ++t) out[t] = acc[t];
    }
    if (S > 1) {
        __syncthreads();
        for (int o = tid; o < CC * NT; o += 256) {
            const int pr = o / NT, t = o % NT;
            float a = 0.f;
            for (int sg = 0; sg < S; ++sg) a += red[((size_t)pr * S + sg) * NT + t];
            partial[(size_t)blockIdx.x * CC * NT + o] = a;
        }
    }
}

// The same sums on the matrix pipe for the layers with 12 and more channels (the CIFAR / ImageNet-32 levels: C = 12, 24,
// 48): per image part a GEMM  partial[co][(ci, t)] = sum_p gz[co][p] * x[ci][p + tap t]  with M = C, N = C KH KW and the
// pixels as the reduction index, on v_mfma_f32_16x16x4_f32 -- fp32 products and sums, as the FMA form, at two LDS words a
// lane per 1024 multiply-adds where the FMA form reads ten per 576: 16.1 -> 11.1 us a launch at (32, 12, 16, 16),
// 12.9 -> 11.0 at (32, 24, 8, 8), 19.2 -> 17.6 at (32, 48, 4, 4) (what remains is staging and the slab's stores; four
// independent sums with a leaner staging pass measured the same within 1 us either way).  Staging as above; the pitch of a gz plane is 4 (mod 8) words, so that the 16 rows x 4 pixels of an A
// fragment fall into 64 different banks.  A wave owns output tiles (16 co x 16 (ci, t)); a reduction step is four
// consecutive pixels of an image row (lane / 16 picks the pixel; columns past W contribute zeros).
__host__ __device__ inline int wgrad_mma_pitch(int HW) { return HW + ((4 - HW) & 7); }

typedef float wg_f4 __attribute__((ext_vector_type(4)));

template <int KH, int KW>
__global__ __launch_bounds__(256) void k_wgrad_batchmma(const float *__restrict__ gz, const float *__restrict__ x,
                                                        float *__restrict__ partial, int C, int H, int W, int pt, int pl,
                                                        int nsplit)
{
    extern __shared__ float smem[];
    constexpr int NT = KH * KW;
    const int HW = H * W, PG = wgrad_mma_pitch(HW), WH = W + KW - 1, HH = H + KH - 1, PX = (HH * WH) | 1;
    float *gs = smem, *xs = smem + C * PG;
    const int b = blockIdx.x / nsplit, part = blockIdx.x % nsplit, tid = threadIdx.x;
    const int r0 = part * H / nsplit, r1 = (part + 1) * H / nsplit;
    for (int i = tid; i < C * PX; i += 256) xs[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < C * HW; i += 256) {
        const int c = i / HW, r = i % HW, h = r / W, w = r % W;
        gs[c * PG + r] = gz[(size_t)b * C * HW + i];
        xs[c * PX + (h + pt) * WH + (w + pl)] = x[(size_t)b * C * HW + i];
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6, m = lane & 15, kq = lane >> 4;
    const int N = C * NT, NTL = (N + 15) / 16, tiles = ((C + 15) / 16) * NTL;
    float *out = partial + (size_t)blockIdx.x * C * N;
    for (int tile = wv; tile < tiles; tile += 4) {
        const int mt = tile / NTL, nt = tile % NTL;
        const int co = 16 * mt + m, n = 16 * nt + m;
        const bool aok = co < C, bok = n < N;
        const int nn = bok ? n : 0, ci = nn / NT, t = nn % NT;
        const float *gp = gs + (aok ? co : 0) * PG;
        const float *xp = xs + ci * PX + (t / KW) * WH + (t % KW);
        wg_f4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int oh = r0; oh < r1; ++oh)
            for (int ow0 = 0; ow0 < W; ow0 += 4) {
                const int ow = ow0 + kq;
                const bool ok = ow < W;
                const float av = gp[oh * W + (ok ? ow : 0)], bv = xp[oh * WH + (ok ? ow : 0)];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aok && ok ? av : 0.f, bok && ok ? bv : 0.f, acc, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + 4 * kq + r;
            if (row < C && bok) out[(size_t)row * N + n] = acc[r];
        }
    }
}

// dw[o] = scale * sum_b partial[b][o], masked like k_wgrad_direct.  64 outputs per workgroup; the images are summed in
// four interleaved sequences (b = s, s+4, ...), combined as (s0 + s1) + (s2 + s3): a fixed order, deterministic.
__global__ __launch_bounds__(256) void k_wgrad_batchred(const float *__restrict__ partial, float *__restrict__ dw, int B,
                                                        int C, int KH, int KW, float scale, int mask_mode, int mkh, int mkw)
{
    __shared__ float sh[4][64];
    const int NT = KH * KW, NO = C * C * NT;
    const int ol = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + ol;
    float acc = 0.f;
    if (o < NO) {
#pragma unroll 4
        for (int b = seg; b < B; b += 4) acc += partial[(size_t)b * NO + o];
    }
    sh[seg][ol] = acc;
    __syncthreads();
    if (seg != 0 || o >= NO) return;
    const int t = o % NT, ci = (o / NT) % C, co = o / (NT * C);
    float val = ((sh[0][ol] + sh[1][ol]) + (sh[2][ol] + sh[3][ol])) * scale;
    if (mask_mode && t / KW == mkh && t % KW == mkw) {
        if (mask_mode == 1 && ci >= co) val = 0.f;
        if (mask_mode == 2 && ci > co) val = 0.f;
    }
    dw[o] = val;
}

// (room for two slabs per image: small batches split an image's rows over two workgroups)
size_t wgrad_small_workspace_bytes(int B, int C, int KH, int KW) { return (size_t)2 * B * C * C * KH * KW * sizeof(float) + 256; }

static bool wgrad_small_mma(int C) { return C >= 12; }
static size_t wgrad_small_lds_bytes(int C, int H, int W, int KH, int KW)
{
    if (wgrad_small_mma(C)) return (size_t)C * (wgrad_mma_pitch(H * W) + (((H + KH - 1) * (W + KW - 1)) | 1)) * sizeof(float);
    const size_t red = C * C < 256 ? (size_t)256 * KH * KW : 0; // segment sums of the layers with few (co, ci) pairs
    return ((size_t)C * (((H * W) | 1) + (((H + KH - 1) * (W + KW - 1)) | 1)) + red) * sizeof(float);
}

// one workgroup per image pays as long as both tensors of the image fit the LDS and a thread's (co, ci) pairs are few:
// the small layers of the reference models (C = 1..8 MNIST, C = 12 / 24 / 48 ImageNet-32 and CIFAR levels that the
// MFMA kernel does not cover).  K in {2x2, 3x3} (register-resident tap sums).
bool wgrad_small_supported(int B, int C, int H, int W, int KH, int KW)
{
    if (!((KH == 2 && KW == 2) || (KH == 3 && KW == 3))) return false;
    return B >= 1 && C <= 48 && wgrad_small_lds_bytes(C, H, W, KH, KW) <= 64 * 1024;
}

int launch_wgrad_small(const float *gz, const float *x, float *dw, void *ws, int B, int C, int H, int W, int KH, int KW,
                       int pt, int pl, float scale, int mask_mode, int mkh, int mkw, hipStream_t s)
{
    float *partial = (float *)ws;
    const size_t lds = wgrad_small_lds_bytes(C, H, W, KH, KW);
    const int nsplit = (B <= 128 && H >= 8) ? 2 : 1; // (batches that leave half the compute units idle)
    if (wgrad_small_mma(C)) {
        if (KH == 2)
            hipLaunchKernelGGL((k_wgrad_batchmma<2, 2>), dim3(B * nsplit), dim3(256), lds, s, gz, x, partial, C, H, W, pt, pl, nsplit);
        else
            hipLaunchKernelGGL((k_wgrad_batchmma<3, 3>), dim3(B * nsplit), dim3(256), lds, s, gz, x, partial, C, H, W, pt, pl, nsplit);
    } else if (KH == 2)
        hipLaunchKernelGGL((k_wgrad_batchpar<2, 2>), dim3(B * nsplit), dim3(256), lds, s, gz, x, partial, C, H, W, pt, pl, nsplit);
    else
        hipLaunchKernelGGL((k_wgrad_batchpar<3, 3>), dim3(B * nsplit), dim3(256), lds, s, gz, x, partial, C, H, W, pt, pl, nsplit);
    IFL_HIP(hipGetLastError());
    const int NO = C * C * KH * KW;
    hipLaunchKernelGGL(k_wgrad_batchred, dim3((NO + 63) / 64), dim3(256), 0, s, partial, dw, B * nsplit, C, KH, KW, scale,
                       mask_mode, mkh, mkw);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

} // namespace ifl
