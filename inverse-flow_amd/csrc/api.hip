// C-ABI entry points of libinvflow_hip.so (see include/invflow.h for the contract and the
// reference interfaces each one replaces).  Argument validation, workspace carving and
// kernel selection only -- no device allocation, no synchronisation, no environment variable, and no state besides the
// thread-local error text / profiling records and the once-per-device LDS opt-in latches (ifl_common.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "ifl_common.h"

namespace ifl {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void clear_error() { g_err[0] = 0; }

// ---- optional per-kernel timing (bench.py's roofline leg) -------------------------------------
// When enabled for the calling thread, every launch of a tagged kernel is bracketed by a pair of
// hipEvents recorded on the stream the kernel is launched on; ifl_profile_collect() synchronises
// on them and returns the summed device time per tag.  Disabled (the default) it costs nothing.
struct ProfRec {
    hipEvent_t a, b;
    int tag;
};
static thread_local bool g_prof_on = false;
static thread_local std::vector<ProfRec> g_prof_live;
static thread_local std::vector<ProfRec> g_prof_pool;

struct ProfScope {
    hipStream_t s;
    ProfRec r;
    bool on;
    ProfScope(int tag, hipStream_t stream) : s(stream), on(g_prof_on)
    {
        if (!on) return;
        if (!g_prof_pool.empty()) {
            r = g_prof_pool.back();
            g_prof_pool.pop_back();
        } else if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
            on = false;
            return;
        }
        r.tag = tag;
        (void)hipEventRecord(r.a, s);
    }
    ~ProfScope()
    {
        if (!on) return;
        (void)hipEventRecord(r.b, s);
        g_prof_live.push_back(r);
    }
};

struct Carver {
    char *base;
    size_t off, cap;
    Carver(void *ws, size_t bytes) : base((char *)ws), off(0), cap(bytes) {}
    template <typename T> T *take(size_t n)
    {
        off = align_up(off, 256);
        T *p = (T *)(base + off);
        off += n * sizeof(T);
        return p;
    }
    bool ok() const { return off <= cap && (base != nullptr || off == 0); }
};

static int check_shape(const char *fn, int B, int C, int H, int W, int KH, int KW, int order)
{
    if (B < 0 || C < 1 || H < 1 || W < 1 || KH < 1 || KW < 1)
        IFL_FAIL(IFL_EINVAL, "%s: bad shape B=%d C=%d H=%d W=%d K=%dx%d", fn, B, C, H, W, KH, KW);
    if (order < IFL_ORDER_TL || order > IFL_ORDER_BR) IFL_FAIL(IFL_EINVAL, "%s: unknown order: %d", fn, order);
    if ((size_t)B * C * H * W >= ((size_t)1 << 31))
        IFL_FAIL(IFL_EUNSUPPORTED, "%s: tensor has >= 2^31 elements", fn);
    return IFL_OK;
}

// buffers that may hold MFMA-packed weights are sized for the padded channel count (scan_mfma.hip)
static size_t cpad(int C) { return C <= 64 ? (size_t)mfma_padded_channels(C) : (size_t)C; }

static size_t fold_bytes(int B, int C, int H, int W, int KH, int KW)
{
    const size_t cp = cpad(C);
    const Geom g = make_geom(B, C, H, W, KH, KW, IFL_ORDER_TL, 0);
    return align_up(cp * cp * sizeof(double), 256) + 2 * align_up((size_t)KH * KW * cp * cp * sizeof(float), 256) +
           align_up(((size_t)B + 1) * sizeof(int), 256) + 1024 + align_up(scan_team_ws_bytes(g), 256);
}

// out = conv(in, w) + bias, same-size or general: the MFMA kernel where it applies (needs `pack` bytes of
// conv_pack_bytes), the direct kernel otherwise
static size_t conv_pack_bytes(int Ci, int Co, int KH, int KW)
{
    return Ci == Co ? align_up(conv_mfma_pack_bytes(Ci, KH, KW), 256) + 256 : 0;
}
static int run_conv(const float *in, const float *w, const float *bias, float *out, int B, int Ci, int Co, int H, int W,
                    int OH, int OW, int KH, int KW, int pt, int pl, void *pack, hipStream_t s)
{
    if (pack && conv_mfma_supported(Ci, Co, H, W, OH, OW, KH, KW, pt, pl))
        return launch_conv_mfma(in, w, bias, out, pack, B, Ci, H, W, KH, KW, pt, pl, s);
    return launch_conv_direct(in, w, bias, out, B, Ci, Co, H, W, OH, OW, KH, KW, pt, pl, s);
}

// xhat = A z with the layer's effective weight (and log|det A| per image when `logdet`): two launches where the MFMA
// convolution applies (effective weight + fp16 pack + log-det in one, then the convolution), the direct kernels otherwise
static bool conv_eff_on_mfma(const Geom &g, unsigned flags, void *pack)
{
    const int pt = g.flipH ? 0 : g.KH - 1, pl = g.flipW ? 0 : g.KW - 1;
    return pack && !(flags & (IFL_FLAG_NO_MFMA | IFL_FLAG_EXACT_F32)) && conv_mfma_supported(g.C, g.C, g.H, g.W, g.H, g.W, g.KH, g.KW, pt, pl);
}
static int run_conv_eff(const float *z, const float *w, float *weff, float *xhat, float *logdet, const Geom &g, unsigned flags,
                        void *pack, hipStream_t s, const ConvMix *mix = nullptr)
{
    int rc;
    const int pt = g.flipH ? 0 : g.KH - 1, pl = g.flipW ? 0 : g.KW - 1; // padding corner; also the diagonal tap's stored position
    if (pack && !(flags & (IFL_FLAG_NO_MFMA | IFL_FLAG_EXACT_F32)) && conv_mfma_supported(g.C, g.C, g.H, g.W, g.H, g.W, g.KH, g.KW, pt, pl)) {
        const ConvEff eff{weff, logdet, pt, pl, g.general_diag, g.B, g.H, g.W};
        ProfScope ps(IFL_PROF_CONV, s);
        return launch_conv_mfma(z, w, nullptr, xhat, pack, g.B, g.C, g.H, g.W, g.KH, g.KW, pt, pl, s, &eff, mix);
    }
    if ((rc = launch_effw(w, weff, g, s))) return rc;
    {
        ProfScope ps(IFL_PROF_CONV, s);
        if ((rc = launch_conv_direct(z, weff, nullptr, xhat, g.B, g.C, g.C, g.H, g.W, g.H, g.W, g.KH, g.KW, pt, pl, s))) return rc;
    }
    if (logdet && (rc = launch_logdet(w, logdet, g, s))) return rc;
    return IFL_OK;
}

// Forward -> backward side channel ("carry", caller-owned, ifl_carry_bytes): the folded + packed weights of
// the adjoint, produced by the forward call's single fold launch, and two words collecting max|z| / max|dx|
// from the scans (the weight-gradient kernel's power-of-two prescale), so that the backward needs neither
// a fold launch nor a pass over z and dx.
struct CarryView {
    unsigned *zmax, *dxmax;
    void *adj_pack;
    float *adj_wf32;
};
static size_t carry_bytes(int C, int KH, int KW)
{
    const size_t cp = cpad(C);
    return 256 + 2 * align_up((size_t)KH * KW * cp * cp * sizeof(float), 256);
}
static CarryView carry_view(void *carry, int C, int KH, int KW)
{
    char *p = (char *)carry;
    CarryView v;
    v.zmax = (unsigned *)p;
    v.dxmax = (unsigned *)(p + 64);
    v.adj_pack = p + 256;
    v.adj_wf32 = (float *)(p + 256 + align_up((size_t)KH * KW * cpad(C) * cpad(C) * sizeof(float), 256));
    return v;
}
// the carry is used iff the *shape* can take the MFMA scan: both calls evaluate this on the host
static bool carry_usable(const Geom &g, unsigned flags)
{
    static float dummy_aligned __attribute__((aligned(16)));
    return !(flags & (IFL_FLAG_NO_MFMA | IFL_FLAG_EXACT_F32)) &&
           (scan_mfma_supported(g, &dummy_aligned, &dummy_aligned) || scan_team_supported(g));
}

// fold + scan (shared by inverse and dx): z = scan(x) for the operator or its adjoint.
//   carry_out (forward): also fold the adjoint into the carry and collect max|z| there.
//   carry_in (backward, transposed): take the packed adjoint from the carry, collect max|dx| there.
// *amax_valid tells the caller whether the scan wrote max|output| into the carry word.
static int run_scan(const ScanIO &io, const float *w, const Geom &g, int transposed, unsigned flags,
                    Carver &cv, void *carry_out, void *carry_in, bool *amax_valid, void *scan_state, hipStream_t s)
{
    // (bf16 storage -- x16 / z16 -- is taken by the duo scan only: native_bf16_ok() is what the *_bf16 entry points ask first)
    const float *x = io.x32;
    float *z = io.z32;
    const void *xa = io.x16 ? (const void *)io.x16 : (const void *)io.x32;
    const void *za = io.z32 ? (const void *)io.z32 : (const void *)io.z16;
    const size_t cp = cpad(g.C);
    double *linv = cv.take<double>(cp * cp);
    float *wf = cv.take<float>((size_t)g.KH * g.KW * cp * cp); // folded taps (fp32) or packed fp16 hi/lo fragments
    float *wf2 = cv.take<float>((size_t)g.KH * g.KW * cp * cp); // fp32 copy of the folded taps (in-kernel fp32 redo)
    int *ovf = cv.take<int>((size_t)g.B + 1);                    // per-image overflow flags of the MFMA scan
    if (!cv.ok()) IFL_FAIL(IFL_EWORKSPACE, "workspace too small: need %zu bytes, have %zu", cv.off, cv.cap);
    int rc;
    if (amax_valid) *amax_valid = false;
    const int rh = g.flipH ^ (transposed ? 1 : 0), rw = g.flipW ^ (transposed ? 1 : 0);
    const bool usable = carry_usable(g, flags);
    const bool team_shape = usable && scan_team_supported(g);
    const bool mfma = usable && !team_shape && scan_mfma_supported(g, xa, za) && !(((uintptr_t)io.z16) & 15);
    if ((io.x16 || io.z16) && !(mfma && scan_duo_route(g, scan_state, (flags & IFL_FLAG_WHOLE_IMAGE) != 0)))
        IFL_FAIL(IFL_EUNSUPPORTED, "bf16 storage: this shape does not take the duo scan");
    CarryView co{}, ci{};
    if (carry_out && usable) co = carry_view(carry_out, g.C, g.KH, g.KW);
    if (carry_in && usable) ci = carry_view(carry_in, g.C, g.KH, g.KW);
    if (team_shape) {
        // wide layers: the fold of this file's route serves both its resident scan and the launch-per-diagonal one
        char *extra = cv.take<char>(scan_team_ws_bytes(g));
        if (!cv.ok()) IFL_FAIL(IFL_EWORKSPACE, "workspace too small: need %zu bytes, have %zu", cv.off, cv.cap);
        const void *pack = wf;
        const float *pack32 = wf2;
        {
            ProfScope ps(IFL_PROF_FOLD, s);
            if (co.zmax) {
                if ((rc = launch_fold_team(w, extra, g, transposed, 2, wf, wf2, co.adj_pack, co.adj_wf32, co.zmax, co.dxmax, s)))
                    return rc;
            } else if (ci.zmax) {
                pack = ci.adj_pack;
                pack32 = ci.adj_wf32;
            } else if ((rc = launch_fold_team(w, extra, g, transposed, 1, wf, wf2, nullptr, nullptr, nullptr, nullptr, s))) {
                return rc;
            }
        }
        ProfScope ps(IFL_PROF_SCAN, s);
        if (scan_state && !(flags & IFL_FLAG_WHOLE_IMAGE) && x != z) {
            if ((uintptr_t)scan_state & 255) IFL_FAIL(IFL_EINVAL, "scan_state must be 256-byte aligned");
            unsigned *amax = co.zmax ? co.zmax : (ci.zmax ? ci.dxmax : nullptr);
            if (amax_valid) *amax_valid = amax != nullptr;
            void *wide_state = (char *)scan_state + align_up(scan_duo_state_bytes(), 256);
            if ((rc = launch_scan_team(x, pack, z, g, rh, rw, extra, wide_state, ovf, amax, s))) return rc;
            // (a no-op unless the launch gave up or a value left the fp16 range: then exact fp32 from x)
            return launch_scan_general(x, pack32, z, g, rh, rw, s, ovf, 0);
        }
        if (scan_wide_supported(g) && x != z) return launch_scan_wide(x, pack32, z, g, rh, rw, s);
        return launch_scan_general(x, pack32, z, g, rh, rw, s);
    }
    if (co.zmax) {
        // one launch folds both directions (needed even if this call's pointers force the general scan)
        ProfScope ps(IFL_PROF_FOLD, s);
        if ((rc = launch_foldpack_mfma(w, wf, wf2, co.adj_pack, co.adj_wf32, g, transposed, 2, co.zmax, co.dxmax, s)))
            return rc;
    }
    if (mfma) {
        const void *pack = wf;
        const float *pack32 = wf2;
        if (ci.zmax) {
            pack = ci.adj_pack;
            pack32 = ci.adj_wf32;
        } else if (!co.zmax) {
            ProfScope ps(IFL_PROF_FOLD, s);
            if ((rc = launch_foldpack_mfma(w, wf, wf2, nullptr, nullptr, g, transposed, 1, nullptr, nullptr, s))) return rc;
        }
        unsigned *amax = co.zmax ? co.zmax : (ci.zmax ? ci.dxmax : nullptr);
        if (amax_valid) *amax_valid = amax != nullptr;
        // (an image whose r leaves the fp16 range is redone in exact fp32 inside the same launch)
        ProfScope ps(IFL_PROF_SCAN, s);
        return launch_scan_mfma(io, pack, g, rh, rw, ovf, pack32, amax, scan_state, (flags & IFL_FLAG_WHOLE_IMAGE) != 0, s);
    }
    {
        ProfScope ps(IFL_PROF_FOLD, s);
        if ((rc = launch_linv_fold(w, linv, wf, g, transposed, s))) return rc;
    }
    ProfScope ps(IFL_PROF_SCAN, s);
    if (scan_resident_supported(g)) return launch_scan_resident(x, wf, z, g, rh, rw, s);
    if (scan_wide_supported(g) && x != z) return launch_scan_wide(x, wf, z, g, rh, rw, s);
    return launch_scan_general(x, wf, z, g, rh, rw, s);
}

// padding corner of the stored-layout convolution for an order (inf/layers/inv_conv.py:126-144)
static void order_pads(const Geom &g, int &pt, int &pl, int &dkh, int &dkw)
{
    pt = g.flipH ? 0 : g.KH - 1;
    pl = g.flipW ? 0 : g.KW - 1;
    dkh = g.flipH ? 0 : g.KH - 1; // stored position of the diagonal tap
    dkw = g.flipW ? 0 : g.KW - 1;
}

} // namespace ifl

using namespace ifl;

namespace ifl {

bool native_bf16_ok(int B, int C, int H, int W, int KH, int KW, int order, unsigned flags, const void *scan_state, const void *a,
                    const void *b)
{
    if (B < 1 || (((uintptr_t)a | (uintptr_t)b) & 15)) return false;
    const Geom g = make_geom(B, C, H, W, KH, KW, order, flags);
    return carry_usable(g, flags) && !scan_team_supported(g) && scan_duo_route(g, scan_state, (flags & IFL_FLAG_WHOLE_IMAGE) != 0);
}

// z = A^-1 x on whatever storage `io` names (shape and pointers checked by the entry point)
int inverse_io(const ScanIO &io, const float *w, int B, int C, int H, int W, int KH, int KW, int order, unsigned flags, void *ws,
               size_t ws_bytes, void *carry, void *scan_state, hipStream_t stream)
{
    Geom g = make_geom(B, C, H, W, KH, KW, order, flags);
    Carver cv(ws, ws_bytes);
    bool amax_ok = false;
    int rc = run_scan(io, w, g, 0, flags, cv, carry, nullptr, &amax_ok, scan_state, stream);
    if (rc) return rc;
    if (carry && carry_usable(g, flags) && !amax_ok && io.z32) {
        // the general scan ran (unaligned pointers): collect max|z| with a streaming pass (rare path)
        const CarryView cvw = carry_view(carry, C, KH, KW);
        if ((rc = launch_absmax(io.z32, (size_t)B * C * H * W, cvw.zmax, stream))) return rc;
    }
    return IFL_OK;
}

} // namespace ifl

extern "C" {

int ifl_version(void) { return 2100; }

void ifl_profile_enable(int on)
{
    g_prof_on = on != 0;
}

int ifl_profile_collect(int tag, double *total_ms, int *launches)
{
    double tot = 0.0;
    int n = 0;
    std::vector<ProfRec> keep;
    for (ProfRec &r : g_prof_live) {
        if (r.tag != tag) {
            keep.push_back(r);
            continue;
        }
        float ms = 0.f;
        IFL_HIP(hipEventSynchronize(r.b));
        IFL_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        tot += ms;
        ++n;
        g_prof_pool.push_back(r);
    }
    g_prof_live.swap(keep);
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return IFL_OK;
}

const char *ifl_last_error(void) { return g_err; }

size_t ifl_workspace_bytes(int op, int B, int C, int H, int W, int KH, int KW, unsigned flags)
{
    (void)flags;
    if (B < 0 || C < 1 || H < 1 || W < 1 || KH < 1 || KW < 1) return 0;
    const size_t n = align_up((size_t)B * C * H * W * sizeof(float), 256) + 256;
    const size_t wbytes = align_up((size_t)KH * KW * C * C * sizeof(float), 256) + 256 + conv_pack_bytes(C, C, KH, KW);
    switch (op) {
    case IFL_OP_INVERSE:
    case IFL_OP_DY:
        return fold_bytes(B, C, H, W, KH, KW);
    case IFL_OP_FORWARD:
        return wbytes;
    case IFL_OP_BACKWARD:
        // fold + (dx when the caller passes none) + (A z and mixed gradient for the recon term) + dW partials
        return fold_bytes(B, C, H, W, KH, KW) + wbytes + 3 * n + wgrad_mfma_workspace_bytes(B, C, H, KH, KW) +
               wgrad_small_workspace_bytes(B < 0 ? 0 : B, C, KH, KW) + 512;
    case IFL_OP_DW:
        return wgrad_mfma_workspace_bytes(B, C, H, KH, KW) + wgrad_small_workspace_bytes(B < 0 ? 0 : B, C, KH, KW) + 512;
    default:
        return 0;
    }
}

size_t ifl_scan_state_bytes(void) { return align_up(scan_duo_state_bytes(), 256) + scan_wide_state_bytes(); }

size_t ifl_scan_state_voided_offset(void) { return align_up(scan_duo_state_bytes(), 256) + scan_wide_voided_offset(); }

size_t ifl_carry_bytes(int C, int KH, int KW)
{
    if (C < 1 || KH < 1 || KW < 1) return 0;
    return carry_bytes(C, KH, KW);
}

int ifl_inverse_f32(const float *x, const float *w, float *z, int B, int C, int H, int W, int KH, int KW, int order,
                    unsigned flags, void *ws, size_t ws_bytes, void *carry, void *scan_state, ifl_stream_t stream)
{
    clear_error();
    int rc = check_shape("ifl_inverse_f32", B, C, H, W, KH, KW, order);
    if (rc) return rc;
    if (B == 0) return IFL_OK;
    if (!x || !w || !z) IFL_FAIL(IFL_EINVAL, "ifl_inverse_f32: null tensor pointer");
    if (x == z) IFL_FAIL(IFL_EINVAL, "ifl_inverse_f32: z must not alias x (an image that leaves the fp16 range is redone from x)");
    return ifl::inverse_io(scan_io_f32(x, z), w, B, C, H, W, KH, KW, order, flags, ws, ws_bytes, carry, scan_state, (hipStream_t)stream);
}

int ifl_forward_f32(const float *z, const float *w, float *xhat, float *logdet, int B, int C, int H, int W, int KH,
                    int KW, int order, unsigned flags, void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    int rc = check_shape("ifl_forward_f32", B, C, H, W, KH, KW, order);
    if (rc) return rc;
    if (B == 0) return IFL_OK;
    if (!z || !w || !xhat) IFL_FAIL(IFL_EINVAL, "ifl_forward_f32: null tensor pointer");
    hipStream_t s = (hipStream_t)stream;
    Geom g = make_geom(B, C, H, W, KH, KW, order, flags);
    Carver cv(ws, ws_bytes);
    float *weff = cv.take<float>((size_t)KH * KW * C * C);
    void *pack = cv.take<unsigned char>(conv_pack_bytes(C, C, KH, KW));
    if (!cv.ok()) IFL_FAIL(IFL_EWORKSPACE, "ifl_forward_f32: workspace too small: need %zu bytes, have %zu", cv.off, cv.cap);
    return run_conv_eff(z, w, weff, xhat, logdet, g, flags, pack, s);
}

static int dw_impl(const float *z, const float *dx, float *dw, int B, int C, int H, int W, int KH, int KW, int order,
                   unsigned flags, void *ws, size_t ws_bytes, const unsigned *amax_dx, const unsigned *amax_z,
                   hipStream_t s)
{
    Geom g = make_geom(B, C, H, W, KH, KW, order, flags);
    if (B == 0) {
        IFL_HIP(hipMemsetAsync(dw, 0, (size_t)C * C * KH * KW * sizeof(float), s));
        return IFL_OK;
    }
    int pt, pl, dkh, dkw;
    order_pads(g, pt, pl, dkh, dkw);
    ProfScope ps(IFL_PROF_WGRAD, s);
    if (!(flags & (IFL_FLAG_NO_MFMA | IFL_FLAG_EXACT_F32)) && wgrad_mfma_supported(B, C, H, W, KH, KW, pt, pl, dx, z)) {
        Carver cv(ws, ws_bytes);
        void *wws = cv.take<char>(wgrad_mfma_workspace_bytes(B, C, H, KH, KW));
        if (cv.ok())
            return launch_wgrad_mfma(dx, z, dw, wws, B, C, H, W, KH, KW, pt, pl, -1.0f, g.general_diag ? 2 : 1, dkh,
                                     dkw, amax_dx, amax_z, s);
        IFL_FAIL(IFL_EWORKSPACE, "ifl_dw_f32: workspace too small: need %zu bytes, have %zu", cv.off, cv.cap);
    }
    if (!(flags & (IFL_FLAG_NO_MFMA | IFL_FLAG_EXACT_F32)) && wgrad_w8_supported(B, C, H, W, KH, KW, pt, pl, dx, z)) {
        Carver cv(ws, ws_bytes);
        void *wws = cv.take<char>(256);
        if (cv.ok())
            return launch_wgrad_w8(dx, z, dw, wws, B, C, H, KH, KW, pt, pl, -1.0f, g.general_diag ? 2 : 1, dkh, dkw, amax_dx,
                                   amax_z, s);
    }
    if (wgrad_small_supported(B, C, H, W, KH, KW)) {
        Carver cv(ws, ws_bytes);
        void *wws = cv.take<char>(wgrad_small_workspace_bytes(B, C, KH, KW));
        if (cv.ok())
            return launch_wgrad_small(dx, z, dw, wws, B, C, H, W, KH, KW, pt, pl, -1.0f, g.general_diag ? 2 : 1, dkh, dkw, s);
    }
    return launch_wgrad_direct(dx, z, dw, B, C, C, H, W, H, W, KH, KW, pt, pl, -1.0f, g.general_diag ? 2 : 1, dkh, dkw,
                               s);
}

int ifl_dw_f32(const float *z, const float *dx, float *dw, int B, int C, int H, int W, int KH, int KW, int order,
               unsigned flags, void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    int rc = check_shape("ifl_dw_f32", B, C, H, W, KH, KW, order);
    if (rc) return rc;
    if (!dw || (B > 0 && (!z || !dx))) IFL_FAIL(IFL_EINVAL, "ifl_dw_f32: null tensor pointer");
    return dw_impl(z, dx, dw, B, C, H, W, KH, KW, order, flags, ws, ws_bytes, nullptr, nullptr, (hipStream_t)stream);
}

int ifl_backward_f32(const float *gout, const float *z, const float *x, const float *w, float *dx, float *dw,
                     float recon_weight, float *recon_loss, int B, int C, int H, int W, int KH, int KW, int order,
                     unsigned flags, void *ws, size_t ws_bytes, void *carry, void *scan_state, ifl_stream_t stream)
{
    clear_error();
    int rc = check_shape("ifl_backward_f32", B, C, H, W, KH, KW, order);
    if (rc) return rc;
    if (B > 0 && (!gout || !w)) IFL_FAIL(IFL_EINVAL, "ifl_backward_f32: null tensor pointer");
    if (B > 0 && dw && !z) IFL_FAIL(IFL_EINVAL, "ifl_backward_f32: dw requested but z is null");
    if (B > 0 && dx && dx == gout) IFL_FAIL(IFL_EINVAL, "ifl_backward_f32: dx must not alias g (an image that leaves the fp16 range is redone from g)");
    return ifl::backward_io(ScanIO{gout, nullptr, dx, nullptr}, z, x, w, dw, recon_weight, recon_loss, B, C, H, W, KH, KW, order, flags,
                            ws, ws_bytes, carry, scan_state, (hipStream_t)stream);
}

} // extern "C"

namespace ifl {

// (dx, dW) with g and dx on whatever storage `gio` names: x32 / x16 = g, z32 / z16 = dx (either may be NULL: the weight
// gradient contracts the fp32 dx, which then lives in the workspace).  z, x (recon term), dW are fp32.
int backward_io(const ScanIO &gio, const float *z, const float *x, const float *w, float *dw, float recon_weight, float *recon_loss,
                int B, int C, int H, int W, int KH, int KW, int order, unsigned flags, void *ws, size_t ws_bytes, void *carry,
                void *scan_state, hipStream_t s)
{
    int rc;
    if (!gio.z32 && !gio.z16 && !dw) return IFL_OK;
    Geom g = make_geom(B, C, H, W, KH, KW, order, flags);
    const size_t n = (size_t)B * C * H * W;
    if (B == 0) {
        if (dw) IFL_HIP(hipMemsetAsync(dw, 0, (size_t)C * C * KH * KW * sizeof(float), s));
        if (recon_loss) IFL_HIP(hipMemsetAsync(recon_loss, 0, sizeof(float), s));
        return IFL_OK;
    }
    const bool recon = dw && x && recon_weight != 0.0f;
    Carver cv(ws, ws_bytes);
    float *u = gio.z32 ? gio.z32 : ((dw || !gio.z16) ? cv.take<float>(n) : nullptr);
    bool dxmax_ok = false;
    if ((rc = run_scan(ScanIO{gio.x32, gio.x16, u, gio.z16}, w, g, 1, flags, cv, nullptr, carry, &dxmax_ok, scan_state, s))) return rc;
    if (!dw) return IFL_OK;
    const float *gsrc = u;
    if (recon) {
        float *weff = cv.take<float>((size_t)KH * KW * C * C);
        void *pack = cv.take<unsigned char>(conv_pack_bytes(C, C, KH, KW));
        float *az = cv.take<float>(n);
        float *mix = cv.take<float>(n);
        if (!cv.ok())
            IFL_FAIL(IFL_EWORKSPACE, "ifl_backward_f32: workspace too small: need %zu bytes, have %zu", cv.off, cv.cap);
        // d/dW of rw*mean_b||x - A z||^2 = -(2 rw / B) sum r (x) shifted z  -> fold into the dW reduction
        if (conv_eff_on_mfma(g, flags, pack)) {
            // ... in the epilogue of the convolution A z itself: the residual never exists as a tensor.  (The epilogue ADDS
            // the squared residuals into *recon_loss: cleared here -- the caller's buffer "receives" the loss, invflow.h)
            if (recon_loss) IFL_HIP(hipMemsetAsync(recon_loss, 0, sizeof(float), s));
            const ConvMix mp{u, x, 2.0f * recon_weight / (float)B, recon_loss, 1.0f / (float)B};
            if ((rc = run_conv_eff(z, w, weff, mix, nullptr, g, flags, pack, s, &mp))) return rc;
        } else {
            if ((rc = run_conv_eff(z, w, weff, az, nullptr, g, flags, pack, s))) return rc;
            if ((rc = launch_recon_mix(u, x, az, mix, 2.0f * recon_weight / (float)B, recon_loss, 1.0f / (float)B, n, s)))
                return rc;
        }
        gsrc = mix;
    } else if (recon_loss) {
        IFL_HIP(hipMemsetAsync(recon_loss, 0, sizeof(float), s));
    }
    {
        // hand the rest of the workspace to the dW reduction
        cv.off = align_up(cv.off, 256);
        char *rest = cv.base ? cv.base + cv.off : nullptr;
        const size_t rest_bytes = cv.cap > cv.off ? cv.cap - cv.off : 0;
        // max|dx| and max|z| collected by the two scans make the weight gradient's own pass over them unnecessary
        const unsigned *amax_dx = nullptr, *amax_z = nullptr;
        if (carry && dxmax_ok && !recon && carry_usable(g, flags)) {
            const CarryView cvw = carry_view(carry, C, KH, KW);
            amax_dx = cvw.dxmax;
            amax_z = cvw.zmax;
        }
        return dw_impl(z, gsrc, dw, B, C, H, W, KH, KW, order, flags, rest, rest_bytes, amax_dx, amax_z, s);
    }
}

} // namespace ifl

extern "C" {

// ---- the inverse-flow block: TL -> TR -> BL -> BR (inf/layers/inv_flow.py:13-53) ------------------------------

static const int kUnitOrder[4] = {IFL_ORDER_TL, IFL_ORDER_TR, IFL_ORDER_BL, IFL_ORDER_BR};

size_t ifl_unit_workspace_bytes(int op, int B, int C, int H, int W, int KH, int KW, unsigned flags)
{
    if (op == IFL_OP_INVERSE) return 4 * ifl_workspace_bytes(IFL_OP_INVERSE, B, C, H, W, KH, KW, flags) + 1024;
    if (op == IFL_OP_BACKWARD) {
        const size_t n = align_up((size_t)(B < 0 ? 0 : B) * C * H * W * sizeof(float), 256) + 256;
        return ifl_workspace_bytes(IFL_OP_BACKWARD, B, C, H, W, KH, KW, flags) + 2 * n + 1024;
    }
    return 0;
}

int ifl_unit_inverse_f32(const float *x, const float *const w[4], float *const z[4], int B, int C, int H, int W, int KH,
                         int KW, unsigned flags, void *ws, size_t ws_bytes, void *const carry[4], void *scan_state,
                         ifl_stream_t stream)
{
    clear_error();
    int rc = check_shape("ifl_unit_inverse_f32", B, C, H, W, KH, KW, IFL_ORDER_TL);
    if (rc) return rc;
    if (B == 0) return IFL_OK;
    if (!x || !w || !z) IFL_FAIL(IFL_EINVAL, "ifl_unit_inverse_f32: null pointer");
    for (int l = 0; l < 4; ++l)
        if (!w[l] || !z[l]) IFL_FAIL(IFL_EINVAL, "ifl_unit_inverse_f32: null tensor pointer (layer %d)", l);
    hipStream_t s = (hipStream_t)stream;
    const float *in[4] = {x, z[0], z[1], z[2]};
    Geom g[4];
    bool mfma = !(flags & (IFL_FLAG_NO_MFMA | IFL_FLAG_EXACT_F32));
    for (int l = 0; l < 4; ++l) {
        g[l] = make_geom(B, C, H, W, KH, KW, kUnitOrder[l], flags);
        mfma = mfma && scan_mfma_supported(g[l], in[l], z[l]);
    }
    const size_t per = ifl_workspace_bytes(IFL_OP_INVERSE, B, C, H, W, KH, KW, flags);
    if (!mfma) {
        // shapes the MFMA scan does not cover: the four layers one after the other
        if (ws_bytes < 4 * per) IFL_FAIL(IFL_EWORKSPACE, "ifl_unit_inverse_f32: workspace too small: need %zu bytes, have %zu", 4 * per, ws_bytes);
        for (int l = 0; l < 4; ++l)
            if ((rc = ifl_inverse_f32(in[l], w[l], z[l], B, C, H, W, KH, KW, kUnitOrder[l], flags, (char *)ws + l * per,
                                      per, carry ? carry[l] : nullptr, scan_state, stream)))
                return rc;
        return IFL_OK;
    }
    // one fold launch for the four layers (and, with carries, their adjoints), then the four scans back to back
    Carver cv(ws, ws_bytes);
    FoldJobs jobs;
    float *wf[4], *wf2[4];
    int *ovf[4];
    CarryView co[4];
    const bool with_carry = carry && carry[0] && carry[1] && carry[2] && carry[3];
    for (int l = 0; l < 4; ++l) {
        (void)cv.take<double>(cpad(C) * cpad(C));
        wf[l] = cv.take<float>((size_t)KH * KW * cpad(C) * cpad(C));
        wf2[l] = cv.take<float>((size_t)KH * KW * cpad(C) * cpad(C));
        ovf[l] = cv.take<int>((size_t)B + 1);
        co[l] = CarryView{};
        if (with_carry) co[l] = carry_view(carry[l], C, KH, KW);
        jobs.job[l] = FoldJob{w[l], wf[l], wf2[l], co[l].adj_pack, co[l].adj_wf32, g[l], 0, co[l].zmax, co[l].dxmax};
    }
    if (!cv.ok()) IFL_FAIL(IFL_EWORKSPACE, "ifl_unit_inverse_f32: workspace too small: need %zu bytes, have %zu", cv.off, cv.cap);
    {
        ProfScope ps(IFL_PROF_FOLD, s);
        if ((rc = launch_foldpack_jobs(jobs, 4, with_carry ? 2 : 1, s))) return rc;
    }
    for (int l = 0; l < 4; ++l) {
        ProfScope ps(IFL_PROF_SCAN, s);
        if (in[l] == z[l]) IFL_FAIL(IFL_EINVAL, "ifl_unit_inverse_f32: a layer's output must not alias its input");
        if ((rc = launch_scan_mfma(scan_io_f32(in[l], z[l]), wf[l], g[l], g[l].flipH, g[l].flipW, ovf[l], wf2[l], co[l].zmax, scan_state,
                                   (flags & IFL_FLAG_WHOLE_IMAGE) != 0, s)))
            return rc;
    }
    return IFL_OK;
}

int ifl_unit_backward_f32(const float *gout, const float *const z[4], const float *const w[4], float *dx, float *const dw[4],
                          int B, int C, int H, int W, int KH, int KW, unsigned flags, void *ws, size_t ws_bytes,
                          void *const carry[4], void *scan_state, ifl_stream_t stream)
{
    clear_error();
    int rc = check_shape("ifl_unit_backward_f32", B, C, H, W, KH, KW, IFL_ORDER_TL);
    if (rc) return rc;
    if (!gout || !z || !w || !dx || !dw) IFL_FAIL(IFL_EINVAL, "ifl_unit_backward_f32: null pointer");
    const size_t n = (size_t)B * C * H * W;
    Carver cv(ws, ws_bytes);
    float *t0 = cv.take<float>(n), *t1 = cv.take<float>(n);
    if (!cv.ok()) IFL_FAIL(IFL_EWORKSPACE, "ifl_unit_backward_f32: workspace too small: need %zu bytes, have %zu", cv.off, cv.cap);
    void *rest = (char *)ws + align_up(cv.off, 256);
    const size_t rest_bytes = ws_bytes > align_up(cv.off, 256) ? ws_bytes - align_up(cv.off, 256) : 0;
    // BR -> BL -> TR -> TL: the gradient w.r.t. a layer's input is the next one's output gradient
    const float *gcur = gout;
    for (int l = 3; l >= 0; --l) {
        float *gnext = l == 0 ? dx : ((l & 1) ? t0 : t1);
        if ((rc = ifl_backward_f32(gcur, z[l], nullptr, w[l], gnext, dw[l], 0.0f, nullptr, B, C, H, W, KH, KW, kUnitOrder[l],
                                   flags, rest, rest_bytes, carry ? carry[l] : nullptr, scan_state, stream)))
            return rc;
        gcur = gnext;
    }
    return IFL_OK;
}

// ---- SelfNormConv pieces -------------------------------------------------------------------

static int check_conv(const char *fn, int B, int Ci, int Co, int H, int W, int KH, int KW, int ph, int pw)
{
    if (B < 0 || Ci < 1 || Co < 1 || H < 1 || W < 1 || KH < 1 || KW < 1 || ph < 0 || pw < 0)
        IFL_FAIL(IFL_EINVAL, "%s: bad shape", fn);
    if (H + 2 * ph - KH + 1 < 1 || W + 2 * pw - KW + 1 < 1) IFL_FAIL(IFL_EINVAL, "%s: kernel larger than padded input", fn);
    return IFL_OK;
}

size_t ifl_conv2d_workspace_bytes(int B, int Ci, int Co, int H, int W, int KH, int KW, int ph, int pw)
{
    (void)B; (void)H; (void)W; (void)ph; (void)pw;
    if (Ci < 1 || Co < 1 || KH < 1 || KW < 1) return 0;
    return align_up((size_t)Ci * Co * KH * KW * sizeof(float), 256) + 256 + conv_pack_bytes(Ci, Co, KH, KW) +
           (Ci == Co ? wgrad_mfma_workspace_bytes(B, Ci, H, KH, KW) + 512 : 0);
}

int ifl_conv2d_f32(const float *x, const float *w, const float *bias, float *z, int B, int Ci, int Co, int H, int W,
                   int KH, int KW, int ph, int pw, void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    int rc = check_conv("ifl_conv2d_f32", B, Ci, Co, H, W, KH, KW, ph, pw);
    if (rc) return rc;
    if (!x || !w || !z) IFL_FAIL(IFL_EINVAL, "ifl_conv2d_f32: null tensor pointer");
    const int OH = H + 2 * ph - KH + 1, OW = W + 2 * pw - KW + 1;
    // the workspace is optional here: without it (or with too little) the direct kernel runs
    Carver cv(ws, ws_bytes);
    void *pack = ws ? cv.take<unsigned char>(conv_pack_bytes(Ci, Co, KH, KW)) : nullptr;
    if (!cv.ok() || conv_pack_bytes(Ci, Co, KH, KW) == 0) pack = nullptr;
    return run_conv(x, w, bias, z, B, Ci, Co, H, W, OH, OW, KH, KW, ph, pw, pack, (hipStream_t)stream);
}

int ifl_conv2d_wgrad_f32(const float *gz, const float *x, float *dw, int B, int Ci, int Co, int H, int W, int KH,
                         int KW, int ph, int pw, void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    int rc = check_conv("ifl_conv2d_wgrad_f32", B, Ci, Co, H, W, KH, KW, ph, pw);
    if (rc) return rc;
    if (!gz || !x || !dw) IFL_FAIL(IFL_EINVAL, "ifl_conv2d_wgrad_f32: null tensor pointer");
    const int OH = H + 2 * ph - KH + 1, OW = W + 2 * pw - KW + 1;
    if (B == 0) {
        IFL_HIP(hipMemsetAsync(dw, 0, (size_t)Ci * Co * KH * KW * sizeof(float), (hipStream_t)stream));
        return IFL_OK;
    }
    if (Ci == Co && OH == H && OW == W && wgrad_mfma_supported(B, Ci, H, W, KH, KW, ph, pw, gz, x)) {
        Carver cv(ws, ws_bytes);
        void *wws = cv.take<unsigned char>(wgrad_mfma_workspace_bytes(B, Ci, H, KH, KW));
        if (ws && cv.ok())
            return launch_wgrad_mfma(gz, x, dw, wws, B, Ci, H, W, KH, KW, ph, pw, 1.0f, 0, 0, 0, nullptr, nullptr,
                                     (hipStream_t)stream);
    }
    return launch_wgrad_direct(gz, x, dw, B, Ci, Co, H, W, OH, OW, KH, KW, ph, pw, 1.0f, 0, 0, 0, (hipStream_t)stream);
}

int ifl_conv2d_igrad_f32(const float *gz, const float *w, float *dx, int B, int Ci, int Co, int H, int W, int KH,
                         int KW, int ph, int pw, void *ws, size_t ws_bytes, ifl_stream_t stream)
{
    clear_error();
    int rc = check_conv("ifl_conv2d_igrad_f32", B, Ci, Co, H, W, KH, KW, ph, pw);
    if (rc) return rc;
    if (!gz || !w || !dx) IFL_FAIL(IFL_EINVAL, "ifl_conv2d_igrad_f32: null tensor pointer");
    if (ph > KH - 1 || pw > KW - 1) IFL_FAIL(IFL_EUNSUPPORTED, "ifl_conv2d_igrad_f32: padding larger than K-1");
    const int OH = H + 2 * ph - KH + 1, OW = W + 2 * pw - KW + 1;
    hipStream_t s = (hipStream_t)stream;
    Carver cv(ws, ws_bytes);
    float *wt = cv.take<float>((size_t)Ci * Co * KH * KW);
    void *pack = cv.take<unsigned char>(conv_pack_bytes(Ci, Co, KH, KW));
    if (!cv.ok())
        IFL_FAIL(IFL_EWORKSPACE, "ifl_conv2d_igrad_f32: workspace too small: need %zu bytes, have %zu", cv.off, cv.cap);
    if ((rc = launch_flip_kernel(w, wt, Co, Ci, KH, KW, s))) return rc;
    // dx = conv2d(gz, flip_kernel(w), padding = K-1-p): input (B,Co,OH,OW) -> output (B,Ci,H,W)
    return run_conv(gz, wt, nullptr, dx, B, Co, Ci, OH, OW, H, W, KH, KW, KH - 1 - ph, KW - 1 - pw,
                    conv_pack_bytes(Ci, Co, KH, KW) ? pack : nullptr, s);
}

} // extern "C"
