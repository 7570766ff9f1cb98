// bf16 storage variants of the layer's three entry points (dtype row: "bf16 storage / fp32 accumulate"; the
// reference dispatches on the tensor's dtype, inv_conv_with_bp_kernel_general.cu:112).
//
// Activations (x, z, g, dx, xhat) are bf16 in HBM; weights, weight gradients, log-determinants and every intermediate
// are fp32.  A bf16 call returns exactly the rounded result of the f32 call on the widened inputs.
//
// Layers that take the duo scan (32 or 64 channels, 32-pixel rows: scan_duo.hip) run on bf16 storage NATIVELY: the scan's
// rows arrive as half-width lines by LDS-DMA and are widened inside the LDS-resident tile, finished rows leave rounded (and
// -- the adjoint scan of a backward -- as fp32 too, into the workspace, for the weight gradient, which contracts fp32 dx
// with fp32 z: only z is widened by a streaming pass).  Every other shape widens its inputs into the caller's workspace
// with one streaming pass, runs the f32 path unchanged and narrows the result with another (6 bytes per element and pass;
// 8.5-10 us each at the north-star shape).
#include "ifl_common.h"
#include "bf16_util.h"
#include "../../include/invflow.h"

namespace ifl {

typedef float f4 __attribute__((ext_vector_type(4)));

// eight elements per thread and trip: 16 bytes of bf16, 32 bytes of fp32; up to three tensors of one size per launch
// (blockIdx.y: the backward widens g, z and, for the recon term, x)
struct WidenJobs {
    const bf16_t *src[3];
    float *dst[3];
};
__global__ __launch_bounds__(256) void k_widen(WidenJobs jobs, size_t n)
{
    const bf16_t *__restrict__ src = jobs.src[blockIdx.y];
    float *__restrict__ dst = jobs.dst[blockIdx.y];
    const size_t n8 = n / 8, stride = (size_t)gridDim.x * 256;
    const bool vec = ((((uintptr_t)src) & 15) | (((uintptr_t)dst) & 15)) == 0;
    if (vec) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
            const us8 v = ((const us8 *)src)[i];
            ((f4 *)dst)[2 * i] = f4{widen(v[0]), widen(v[1]), widen(v[2]), widen(v[3])};
            ((f4 *)dst)[2 * i + 1] = f4{widen(v[4]), widen(v[5]), widen(v[6]), widen(v[7])};
        }
    }
    for (size_t i = (vec ? 8 * n8 : 0) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = widen(src[i]);
}
__global__ __launch_bounds__(256) void k_narrow(const float *__restrict__ src, bf16_t *__restrict__ dst, size_t n)
{
    const size_t n8 = n / 8, stride = (size_t)gridDim.x * 256;
    const bool vec = ((((uintptr_t)src) & 15) | (((uintptr_t)dst) & 15)) == 0;
    if (vec) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
            const f4 a = ((const f4 *)src)[2 * i], b = ((const f4 *)src)[2 * i + 1];
            ((us8 *)dst)[i] = us8{narrow_bf16(a[0]), narrow_bf16(a[1]), narrow_bf16(a[2]), narrow_bf16(a[3]),
                                  narrow_bf16(b[0]), narrow_bf16(b[1]), narrow_bf16(b[2]), narrow_bf16(b[3])};
        }
    }
    for (size_t i = (vec ? 8 * n8 : 0) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = narrow_bf16(src[i]);
}

static unsigned stream_grid(size_t n)
{
    const size_t wg = (n / 8 + 255) / 256;
    return (unsigned)(wg < 1 ? 1 : (wg > 2048 ? 2048 : wg));
}
static void widen_to(const uint16_t *src0, float *dst0, const uint16_t *src1, float *dst1, const uint16_t *src2, float *dst2, size_t n,
                     hipStream_t s)
{
    WidenJobs jobs;
    int nj = 0;
    const uint16_t *srcs[3] = {src0, src1, src2};
    float *dsts[3] = {dst0, dst1, dst2};
    for (int k = 0; k < 3; ++k)
        if (srcs[k]) {
            jobs.src[nj] = srcs[k];
            jobs.dst[nj++] = dsts[k];
        }
    for (int k = nj; k < 3; ++k) jobs.src[k] = nullptr, jobs.dst[k] = nullptr;
    if (nj) hipLaunchKernelGGL(k_widen, dim3(stream_grid(n), nj), dim3(256), 0, s, jobs, n);
}
static void narrow_to(const float *src, uint16_t *dst, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_narrow, dim3(stream_grid(n)), dim3(256), 0, s, src, dst, n);
}

static size_t staged_bytes(int B, int C, int H, int W)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return align_up((size_t)B * C * H * W * sizeof(float), 256);
}
static int staged_count(int op) { return op == IFL_OP_BACKWARD ? 4 : 2; } // backward: g, z, x, dx

// the front of the caller's workspace holds the fp32 images, the rest is the f32 entry point's own workspace
struct Staging {
    char *p;
    size_t left;
    bool ok = true;
    Staging(void *ws, size_t bytes)
    {
        const uintptr_t a = ((uintptr_t)ws + 255) & ~(uintptr_t)255;
        const size_t skip = a - (uintptr_t)ws;
        p = (char *)a;
        left = ws && bytes > skip ? bytes - skip : 0;
    }
    float *take(size_t bytes)
    {
        if (bytes > left) {
            ok = false;
            return nullptr;
        }
        float *r = (float *)p;
        p += bytes;
        left -= bytes;
        return r;
    }
};

} // namespace ifl

using namespace ifl;

extern "C" {

size_t ifl_workspace_bytes_bf16(int op, int B, int C, int H, int W, int KH, int KW, unsigned flags)
{
    return 256 + (size_t)staged_count(op) * staged_bytes(B, C, H, W) + ifl_workspace_bytes(op, B, C, H, W, KH, KW, flags);
}

int ifl_inverse_bf16(const uint16_t *x, const float *w, uint16_t *z, int B, int C, int H, int W, int KH, int KW, int order,
                     unsigned flags, void *ws, size_t ws_bytes, void *carry, void *scan_state, ifl_stream_t stream)
{
    clear_error();
    if (B == 0) return ifl_inverse_f32(nullptr, w, nullptr, B, C, H, W, KH, KW, order, flags, ws, ws_bytes, carry, scan_state, stream);
    if (B < 0 || C < 1 || H < 1 || W < 1) IFL_FAIL(IFL_EINVAL, "ifl_inverse_bf16: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    if (!x || !w || !z) IFL_FAIL(IFL_EINVAL, "ifl_inverse_bf16: null tensor pointer");
    if (ws_bytes < ifl_workspace_bytes_bf16(IFL_OP_INVERSE, B, C, H, W, KH, KW, flags) || !ws)
        IFL_FAIL(IFL_EWORKSPACE, "ifl_inverse_bf16: workspace of %zu bytes needed",
                 ifl_workspace_bytes_bf16(IFL_OP_INVERSE, B, C, H, W, KH, KW, flags));
    const size_t n = (size_t)B * C * H * W, nb = staged_bytes(B, C, H, W);
    Staging st(ws, ws_bytes);
    hipStream_t s = (hipStream_t)stream;
    if ((const void *)x == (const void *)z) IFL_FAIL(IFL_EINVAL, "ifl_inverse_bf16: z must not alias x");
    if (KH >= 1 && KW >= 1 && order >= 0 && order <= 3 && native_bf16_ok(B, C, H, W, KH, KW, order, flags, scan_state, x, z))
        return inverse_io(ScanIO{nullptr, x, nullptr, z}, w, B, C, H, W, KH, KW, order, flags, st.p, st.left, carry, scan_state, s);
    float *x32 = st.take(nb), *z32 = st.take(nb);
    widen_to(x, x32, nullptr, nullptr, nullptr, nullptr, n, s);
    if (int rc = ifl_inverse_f32(x32, w, z32, B, C, H, W, KH, KW, order, flags, st.p, st.left, carry, scan_state, stream)) return rc;
    narrow_to(z32, z, n, s);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

int ifl_forward_bf16(const uint16_t *z, const float *w, uint16_t *xhat, float *logdet, int B, int C, int H, int W, int KH, int KW,
                     int order, unsigned flags, void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    if (B == 0) return ifl_forward_f32(nullptr, w, nullptr, logdet, B, C, H, W, KH, KW, order, flags, ws, ws_bytes, stream);
    if (B < 0 || C < 1 || H < 1 || W < 1) IFL_FAIL(IFL_EINVAL, "ifl_forward_bf16: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    if (!z || !w || !xhat) IFL_FAIL(IFL_EINVAL, "ifl_forward_bf16: null tensor pointer");
    if (ws_bytes < ifl_workspace_bytes_bf16(IFL_OP_FORWARD, B, C, H, W, KH, KW, flags) || !ws)
        IFL_FAIL(IFL_EWORKSPACE, "ifl_forward_bf16: workspace of %zu bytes needed",
                 ifl_workspace_bytes_bf16(IFL_OP_FORWARD, B, C, H, W, KH, KW, flags));
    const size_t n = (size_t)B * C * H * W, nb = staged_bytes(B, C, H, W);
    Staging st(ws, ws_bytes);
    float *z32 = st.take(nb), *x32 = st.take(nb);
    hipStream_t s = (hipStream_t)stream;
    widen_to(z, z32, nullptr, nullptr, nullptr, nullptr, n, s);
    if (int rc = ifl_forward_f32(z32, w, x32, logdet, B, C, H, W, KH, KW, order, flags, st.p, st.left, stream)) return rc;
    narrow_to(x32, xhat, n, s);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

int ifl_backward_bf16(const uint16_t *gout, const uint16_t *z, const uint16_t *x, const float *w, uint16_t *dx, float *dw,
                      float recon_weight, float *recon_loss, int B, int C, int H, int W, int KH, int KW, int order,
                      unsigned flags, void *ws, size_t ws_bytes, void *carry, void *scan_state, ifl_stream_t stream)
{
    clear_error();
    if (B == 0)
        return ifl_backward_f32(nullptr, nullptr, nullptr, w, nullptr, dw, recon_weight, recon_loss, B, C, H, W, KH, KW, order, flags,
                                ws, ws_bytes, carry, scan_state, stream);
    if (B < 0 || C < 1 || H < 1 || W < 1) IFL_FAIL(IFL_EINVAL, "ifl_backward_bf16: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    if (!gout || !w) IFL_FAIL(IFL_EINVAL, "ifl_backward_bf16: null tensor pointer");
    // (nothing was asked for: as ifl_backward_f32, no launch)
    if (!dx && !dw) {
        if (recon_loss) IFL_HIP(hipMemsetAsync(recon_loss, 0, sizeof(float), (hipStream_t)stream));
        return IFL_OK;
    }
    if (ws_bytes < ifl_workspace_bytes_bf16(IFL_OP_BACKWARD, B, C, H, W, KH, KW, flags) || !ws)
        IFL_FAIL(IFL_EWORKSPACE, "ifl_backward_bf16: workspace of %zu bytes needed",
                 ifl_workspace_bytes_bf16(IFL_OP_BACKWARD, B, C, H, W, KH, KW, flags));
    const size_t n = (size_t)B * C * H * W, nb = staged_bytes(B, C, H, W);
    Staging st(ws, ws_bytes);
    float *g32 = st.take(nb), *z32 = st.take(nb), *x32 = st.take(nb), *dx32 = st.take(nb);
    hipStream_t s = (hipStream_t)stream;
    if (dw && !z) IFL_FAIL(IFL_EINVAL, "ifl_backward_bf16: dw requested but z is null");
    if (dx && (const void *)dx == (const void *)gout) IFL_FAIL(IFL_EINVAL, "ifl_backward_bf16: dx must not alias g");
    if (KH >= 1 && KW >= 1 && order >= 0 && order <= 3 && native_bf16_ok(B, C, H, W, KH, KW, order, flags, scan_state, gout, dx ? dx : gout)) {
        // g is read and dx written in bf16 by the adjoint scan itself; the weight gradient wants fp32 z (and x for the recon
        // term): one streaming pass; its fp32 dx comes out of the same scan launch
        const bool recon = dw && x && recon_weight != 0.0f;
        if (dw) widen_to(z, z32, recon ? x : nullptr, x32, nullptr, nullptr, n, s);
        return backward_io(ScanIO{nullptr, gout, dw ? dx32 : nullptr, dx}, dw ? z32 : nullptr, recon ? x32 : nullptr, w, dw, recon_weight,
                           recon_loss, B, C, H, W, KH, KW, order, flags, st.p, st.left, carry, scan_state, s);
    }
    widen_to(gout, g32, z, z32, x, x32, n, s); // (one launch)
    // (the weight gradient contracts the fp32 dx with z: dx32 is kept even when the caller does not want dx)
    if (int rc = ifl_backward_f32(g32, z ? z32 : nullptr, x ? x32 : nullptr, w, dx32, dw, recon_weight, recon_loss, B, C, H, W, KH, KW,
                                  order, flags, st.p, st.left, carry, scan_state, stream))
        return rc;
    if (dx) narrow_to(dx32, dx, n, s);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

} // extern "C"
