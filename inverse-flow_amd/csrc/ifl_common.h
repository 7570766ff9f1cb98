// Shared declarations of libinvflow_hip (gfx950).  Internal header: the public C ABI is
// include/invflow.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

#include "../../include/invflow.h"

namespace ifl {

// thread-local error text behind ifl_last_error()
void set_error(const char *fmt, ...);
void clear_error();

#define IFL_FAIL(code, ...)         \
    do {                            \
        ::ifl::set_error(__VA_ARGS__); \
        return (code);              \
    } while (0)

#define IFL_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) IFL_FAIL(IFL_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Opt-in of a kernel to more than 64 KiB of dynamic LDS: an attribute of the function on a device, set once per
// (kernel, device) behind std::call_once -- the only state the library keeps besides the thread-local error text, and it
// never changes after its first use.
static constexpr int IFL_MAX_DEVICES = 64;
struct LdsOptIn {
    std::once_flag once[IFL_MAX_DEVICES];
    hipError_t err[IFL_MAX_DEVICES];
};
static inline int lds_opt_in(LdsOptIn &st, const void *fn, int bytes)
{
    int dev = 0;
    IFL_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= IFL_MAX_DEVICES) {
        IFL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        return IFL_OK;
    }
    std::call_once(st.once[dev], [&] { st.err[dev] = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
    if (st.err[dev] != hipSuccess)
        IFL_FAIL(IFL_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize, %d) failed: %s", bytes, hipGetErrorString(st.err[dev]));
    return IFL_OK;
}

// Geometry of one inverse-conv problem in *logical* (TL-canonical) coordinates.
// A layer of order TR/BL/BR is the TL operator seen through a reflection of the pixel grid and
// of the kernel taps (inf/layers/conv.py:192-219 realises the same thing with torch.flip
// copies; here the reflection is folded into the addressing, nothing is copied).
struct Geom {
    int B, C, H, W, KH, KW;
    int flipH, flipW; // reflect rows / columns of pixels and taps between logical and stored
    int general_diag; // IFL_FLAG_GENERAL_DIAG
};

static inline Geom make_geom(int B, int C, int H, int W, int KH, int KW, int order, unsigned flags)
{
    Geom g;
    g.B = B; g.C = C; g.H = H; g.W = W; g.KH = KH; g.KW = KW;
    g.flipH = (order == IFL_ORDER_BL || order == IFL_ORDER_BR);
    g.flipW = (order == IFL_ORDER_TR || order == IFL_ORDER_BR);
    g.general_diag = (flags & IFL_FLAG_GENERAL_DIAG) ? 1 : 0;
    return g;
}

// ---- prep kernels (prep.hip) ---------------------------------------------------------------
// Linv = L^-1 (double, row-major CxC), L = diagonal-tap matrix of What (unit or general diagonal).
int launch_linv(const float *w, double *linv, const Geom &g, hipStream_t s);
int launch_linv_fold(const float *w, double *linv, float *wf, const Geom &g, int transposed, hipStream_t s); // both, one launch for C <= 32
// Folded taps Wf[t][kc][c] (float), t = dh*KW+dw:  t=0 -> L^-1 (x tap), t>0 -> L^-1 What_t.
// transposed!=0 builds the taps of A^T (for dx = A^-T g): L^-T and L^-T What_t^T.
int launch_fold(const float *w, const double *linv, float *wf, const Geom &g, int transposed, hipStream_t s);
// What in the stored layout (diagonal tap masked / unit) for the forward conv.
int launch_effw(const float *w, float *weff, const Geom &g, hipStream_t s);
// out[b] = H*W*sum_c log|w[c,c,diag tap]|  (0 for the unit diagonal)
int launch_logdet(const float *w, float *out, const Geom &g, hipStream_t s);
// wt[ci][co][kh][kw] = w[co][ci][KH-1-kh][KW-1-kw]   (flip_kernel, inf/layers/selfnorm.py:35-36)
int launch_flip_kernel(const float *w, float *wt, int Co, int Ci, int KH, int KW, hipStream_t s);
// t = dx + coef*(x - az);  *loss += scale * sum (x-az)^2   (recon term of ifl_backward_f32)
int launch_recon_mix(const float *dx, const float *x, const float *az, float *t, float coef, float *loss,
                     float loss_scale, size_t n, hipStream_t s);

// ---- MFMA scan (scan_mfma.hip): 9 <= C <= 64 (padded to 32 / 64), K in {2x2,3x3}, H <= 32, W % 4 == 0 ---------
int mfma_padded_channels(int C); // 32 or 64: what the MFMA kernels are instantiated for
bool scan_mfma_supported(const Geom &g, const void *x, const void *z);
size_t scan_mfma_pack_bytes(const Geom &g);
// mode 0: packed split-fp16 A fragments of the right fold (also zeroes the B overflow flags + the any-flag);
// mode 1: fp32 left fold wf[t][kc][c] for the general scan, skipped unless *gate_any != 0 (overflow fallback)
// ndir = 1: one direction (`transposed`) -> out0/wf0.  ndir = 2: the operator -> out0/wf0 and its adjoint ->
// out1/wf1 in one launch.  zero0/zero1: optional words cleared by the launch (absmax accumulators).
int launch_foldpack_mfma(const float *w, void *out0, float *wf0, void *out1, float *wf1, const Geom &g, int transposed,
                         int ndir, unsigned *zero0, unsigned *zero1, hipStream_t s);
// the same for up to four layers (an inverse-flow block, inf/layers/inv_flow.py:13-53) in ONE launch
struct FoldJob {
    const float *w;
    void *out0;
    float *wf0;
    void *out1;
    float *wf1;
    Geom g;
    int transposed;
    unsigned *zero0, *zero1;
};
struct FoldJobs {
    FoldJob job[4];
};
int launch_foldpack_jobs(const FoldJobs &jobs, int njobs, int ndir, hipStream_t s);
// The duo form of the scan (scan_duo.hip): one workgroup per 16-row tile, chain waves + helper waves.  An image of 17..32
// rows is two workgroups, the upper one handing the lower its last two rows of every diagonal through a mailbox in
// `state`: the caller's block of ifl_scan_state_bytes (include/invflow.h), or NULL.
struct SplitState {
    unsigned long long *mbox; // [image][80 lines][1 KiB]: 8-byte {value, tag} granules
    unsigned *gen;            // [image] launch generation: this launch's tags are gen + 1, gen + 2; the lower part advances it
};
size_t scan_duo_state_bytes();
int scan_duo_max_images();
bool scan_duo_supported(const Geom &g);
// What a scan reads and writes.  x32 or x16 (bf16 storage); z goes to z32 (fp32), to z16 (bf16, rounded to nearest even)
// or to both.  Only the duo scan takes the 16-bit forms; every other route wants x32 / z32.
struct ScanIO {
    const float *x32;
    const uint16_t *x16;
    float *z32;
    uint16_t *z16;
};
static inline ScanIO scan_io_f32(const float *x, float *z) { return ScanIO{x, nullptr, z, nullptr}; }
// whole_image: an image of more than 16 rows stays in ONE workgroup, which sweeps its two tiles in turn (IFL_FLAG_WHOLE_IMAGE)
int launch_scan_duo(const ScanIO &io, const void *apack, const Geom &g, int rh, int rw, int *flags, const float *wf32,
                    unsigned *amax, void *state, bool whole_image, hipStream_t s);
// true when launch_scan_mfma sends this shape to the duo scan with this scan-state block (the route that takes bf16 storage)
bool scan_duo_route(const Geom &g, const void *state, bool whole_image);
// ---- the layer's inverse and backward on whatever storage a ScanIO names (api.hip; the *_bf16 entry points of bf16_io.hip
//      call them with 16-bit activations when native_bf16_ok(), and go through fp32 staging otherwise) -------------------
bool native_bf16_ok(int B, int C, int H, int W, int KH, int KW, int order, unsigned flags, const void *scan_state, const void *a,
                    const void *b);
int inverse_io(const ScanIO &io, const float *w, int B, int C, int H, int W, int KH, int KW, int order, unsigned flags, void *ws,
               size_t ws_bytes, void *carry, void *scan_state, hipStream_t stream);
// gio: x32 / x16 = g, z32 / z16 = dx (either or both; fp32 dx lives in the workspace when the caller has no place for it)
int backward_io(const ScanIO &gio, const float *z, const float *x, const float *w, float *dw, float recon_weight, float *recon_loss,
                int B, int C, int H, int W, int KH, int KW, int order, unsigned flags, void *ws, size_t ws_bytes, void *carry,
                void *scan_state, hipStream_t s);
// amax: optional device word that receives max|z| (atomicMax of float bits; must be cleared beforehand)
// state: the caller's scan-state block or NULL; whole_image: keep one workgroup per image (IFL_FLAG_WHOLE_IMAGE)
int launch_scan_mfma(const ScanIO &io, const void *apack, const Geom &g, int rh, int rw, int *flags, const float *wf32,
                     unsigned *amax, void *state, bool whole_image, hipStream_t s);

// ---- MFMA weight gradient (wgrad_mfma.hip): C in {32,64}, W in {16,32}, K in {2x2,3x3}, corner pads ----
bool wgrad_mfma_supported(int B, int C, int H, int W, int KH, int KW, int pt, int pl, const void *gz, const void *x);
size_t wgrad_mfma_workspace_bytes(int B, int C, int H, int KH, int KW);
// *out = max|a| as float bits (out must not need clearing: the launch clears it first)
int launch_absmax(const float *a, size_t n, unsigned *out, hipStream_t s);
// amax_gz / amax_x: device words holding (an upper bound of) max|gz|, max|x| as float bits, or NULL to compute them
int launch_wgrad_mfma(const float *gz, const float *x, float *dw, void *ws, int B, int C, int H, int W, int KH, int KW,
                      int pt, int pl, float scale, int mask_mode, int mkh, int mkw, const unsigned *amax_gz,
                      const unsigned *amax_x, hipStream_t s);

// ---- MFMA dense same-size convolution (conv_mfma.hip): Ci = Co in {32,64}, K in {2x2,3x3}, W in {16,32} --------
bool conv_mfma_supported(int Ci, int Co, int H, int W, int OH, int OW, int KH, int KW, int pt, int pl);
size_t conv_mfma_pack_bytes(int C, int KH, int KW);
// apack: conv_mfma_pack_bytes of workspace (packed by the launch); w: (C, C, KH, KW) fp32.
// eff (optional): the layer's reverse pass -- the packing launch applies the mask of the diagonal tap at stored position
// (dkh, dkw), leaves the effective weight in eff->weff (C*C*KH*KW floats) and log|det A| per image in eff->logdet (or NULL)
struct ConvEff {
    float *weff;
    float *logdet;
    int dkh, dkw, general_diag;
    int B, H, W;
};
// mix (optional): the reconstruction term of ifl_backward_f32 in the convolution's epilogue -- instead of A z the launch
// stores  t = dx + coef * (x - A z)  (NaN residuals count as 0, inf/layers/selfnorm.py:212) and adds
// loss_scale * sum (x - A z)^2 to *loss: no A z tensor, no separate pass over three activations
struct ConvMix {
    const float *dx, *x;
    float coef;
    float *loss;
    float loss_scale;
};
int launch_conv_mfma(const float *in, const float *w, const float *bias, float *out, void *apack, int B, int C, int H,
                     int W, int KH, int KW, int pt, int pl, hipStream_t s, const ConvEff *eff = nullptr,
                     const ConvMix *mix = nullptr);

// ---- small layers (small_layers.hip): C <= 8, image + result resident in LDS ------------------------------------
bool scan_resident_supported(const Geom &g);
int launch_scan_resident(const float *x, const float *wf, float *z, const Geom &g, int rh, int rw, hipStream_t s);
bool wgrad_small_supported(int B, int C, int H, int W, int KH, int KW);
size_t wgrad_small_workspace_bytes(int B, int C, int KH, int KW);
int launch_wgrad_small(const float *gz, const float *x, float *dw, void *ws, int B, int C, int H, int W, int KH, int KW,
                       int pt, int pl, float scale, int mask_mode, int mkh, int mkw, hipStream_t s);

// ---- wide layers (scan_wide.hip): C > 64, one small GEMM per anti-diagonal over all images ------------------------
bool scan_wide_supported(const Geom &g);
int launch_scan_wide(const float *x, const float *wf, float *z, const Geom &g, int rh, int rw, hipStream_t s);

// ---- wide layers, one resident launch (wide.hip): 64 < C <= 256, C % 32 == 0, H <= 16, W <= 32, K in {2x2,3x3} --------
// A team of C/16 workgroups per tile of 16 columns sweeps all diagonals inside one launch, exchanging each diagonal's
// results through `ws`, synchronised by generation-numbered flags in the caller's scan state.
bool scan_team_supported(const Geom &g);
size_t scan_team_ws_bytes(const Geom &g); // compact L, block inverses, exchange buffer
size_t scan_wide_state_bytes();           // this route's part of the scan state (behind the duo scan's)
size_t scan_wide_voided_offset();         // byte offset, inside that part, of the 64-bit count of voided launches
// fold direction dir0 (0 operator, 1 adjoint) -> pack0 / wf0 (fp32 left fold wf[t][kc][c]); ndir = 2: also the other
// direction -> pack1 / wf1.  zero0 / zero1: optional words cleared by the launch.
int launch_fold_team(const float *w, void *ws, const Geom &g, int dir0, int ndir, void *pack0, float *wf0, void *pack1,
                     float *wf1, unsigned *zero0, unsigned *zero1, hipStream_t s);
// gate: B ints written by the launch (1 -> redo the batch with the general scan on the fp32 fold); state: the wide part
int launch_scan_team(const float *x, const void *pack, float *z, const Geom &g, int rh, int rw, void *ws, void *state,
                     int *gate, unsigned *amax, hipStream_t s);
// weight gradient for W = 8, C > 64 (C % 16 == 0); ws: 256 bytes
bool wgrad_w8_supported(int B, int C, int H, int W, int KH, int KW, int pt, int pl, const void *gz, const void *x);
int launch_wgrad_w8(const float *gz, const float *x, float *dw, void *ws, int B, int C, int H, int KH, int KW, int pt, int pl,
                    float scale, int mask_mode, int mkh, int mkw, const unsigned *amax_gz, const unsigned *amax_x,
                    hipStream_t s);
int launch_absmax2(const float *a, const float *b, size_t n, unsigned *out, hipStream_t s); // out[0], out[1]

// ---- general (any C, any K) VALU kernels (scan_general.hip, conv_general.hip) ----------------
size_t scan_general_lds_bytes(const Geom &g);
// z = scan(x) with folded taps wf; pixel reflection (rh, rw) applied to both x and z addressing.
int launch_scan_general(const float *x, const float *wf, float *z, const Geom &g, int rh, int rw, hipStream_t s,
                        const int *gate = nullptr, int rf = 0);

// out[b][co][oh][ow] = bias[co] + sum w[co][ci][kh][kw] * in[b][ci][oh-pt+kh][ow-pl+kw]
int launch_conv_direct(const float *in, const float *w, const float *bias, float *out, int B, int Ci, int Co,
                       int H, int W, int OH, int OW, int KH, int KW, int pt, int pl, hipStream_t s);
// dw[co][ci][kh][kw] = scale * sum_{b,oh,ow} gz[b][co][oh][ow] * x[b][ci][oh-pt+kh][ow-pl+kw]
// mask_mode 0: none; 1: zero [co][ci>=co] of tap (mkh,mkw); 2: zero [co][ci>co] of that tap.
int launch_wgrad_direct(const float *gz, const float *x, float *dw, int B, int Ci, int Co, int H, int W, int OH,
                        int OW, int KH, int KW, int pt, int pl, float scale, int mask_mode, int mkh, int mkw,
                        hipStream_t s);

} // namespace ifl
