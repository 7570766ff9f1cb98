/*
 * invflow.h -- C ABI of libinvflow_hip.so: the MI355X (gfx950) implementation of
 * Inverse-Flow's inverse-of-convolution hot path.
 *
 * This is the drop-in boundary.  Each entry point replaces one function of the reference's
 * pybind11 CUDA extension `inv_conv_with_bp`
 * (reference: inf/utils/inv_conv_cuda/inv_conv_with_bp_general.cpp:115-120) or one of the
 * ATen/cuDNN calls of inf/utils/convbackward/conv2d_backward.cpp; the citation above every
 * prototype names the reference interface it stands in for.  Plain pointers and sizes only:
 * no torch types, so cgo / JNI / ctypes / pybind can all bind it (INTEGRATION.md shows the
 * Python stub that the reference's inf/layers/inv_conv.py:21 `import inv_conv_with_bp` resolves to).
 *
 * Conventions
 *   - All tensors are device pointers on the *current HIP device* of the calling thread,
 *     contiguous NCHW fp32:  x, z, g, dx : (B, C, H, W);  w, dw : (C, C, KH, KW) = [c_out, c_in, kh, kw].
 *   - `order` is the padding corner of the layer (inf/layers/inv_conv.py:126-144) and `w` is the
 *     layer's *stored* weight for that order (inf/layers/inv_conv.py:172-179 stores it pre-flipped).
 *   - The operator is  A = conv2d(pad_order(.), What)  with What = w whose diagonal tap is
 *     forced to unit-lower-triangular (inf/utils/solve_mc.py:105-109), or -- with
 *     IFL_FLAG_GENERAL_DIAG -- lower-triangular with w's own diagonal
 *     (inf/layers/emerging/inverse_op_cython.pyx:64).
 *   - The library never allocates device memory, never synchronises the device and keeps no
 *     global mutable state: scratch (`ws`, `ws_bytes`, size from ifl_workspace_bytes) and the
 *     persistent block of the two-workgroup scan (`scan_state`, below) are arguments, work is
 *     enqueued on the caller's stream, no environment variable is read.  Re-entrant and
 *     thread-safe (the reference: legacy default stream + cudaDeviceSynchronize after every
 *     launch, inv_conv_with_bp_kernel_general.cu:113-124).
 *   - Return value: 0 on success, a negative IFL_E* code otherwise; ifl_last_error() gives
 *     the message for the calling thread.
 */
#ifndef INVFLOW_H
#define INVFLOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *ifl_stream_t; /* a hipStream_t (NULL = the null stream) */

enum { IFL_ORDER_TL = 0, IFL_ORDER_TR = 1, IFL_ORDER_BL = 2, IFL_ORDER_BR = 3 };

enum {
    IFL_FLAG_GENERAL_DIAG = 1u, /* divide by w[c,c,diag tap] instead of assuming 1 */
    IFL_FLAG_EXACT_F32 = 2u,    /* force the plain-fp32 arithmetic path (no split-precision MFMA) */
    IFL_FLAG_NO_MFMA = 4u,      /* force the general (any C, any K) VALU kernels */
    IFL_FLAG_WHOLE_IMAGE = 8u   /* one workgroup per image even when a scan_state block is given: with the block the same
                                   two sweeps through the same mailbox, bit-identical; without one the round-1 whole-image
                                   kernel (its own summation order: within the tolerance, not bit for bit) */
};

enum { IFL_OK = 0, IFL_EINVAL = -1, IFL_EUNSUPPORTED = -2, IFL_EWORKSPACE = -3, IFL_EHIP = -4 };

enum { IFL_OP_INVERSE = 0, IFL_OP_FORWARD = 1, IFL_OP_BACKWARD = 2, IFL_OP_DY = 3, IFL_OP_DW = 4 };

/* Library / ABI version (major*1000 + minor).
 * 2.0 (2000): every entry point that scans takes the caller's `scan_state` block as an argument (there is no registry inside
 *   the library); ifl_inverse_* rejects z aliasing x with IFL_EINVAL (the scans read x while z rows are already leaving);
 *   *recon_loss RECEIVES the loss on every route (the library clears it; it does not accumulate into the caller's value).
 * 2.1 (2100): + ifl_cond_* (the conditioner of the affine coupling), ifl_adam_flat_f32, ifl_rqspline_p_*; nothing else
 *   changes (ifl_activation_workspace_bytes returns one row more). */
int ifl_version(void);

/* Message of the last failing call made by this thread ("" if none). */
const char *ifl_last_error(void);

/*
 * Optional per-kernel device timing for the calling thread (used by bench.py's roofline leg; the
 * reference's only timing is a cuda.Event pair around a whole train batch,
 * inf/train/experiment.py:276-279).  While enabled, each launch of a tagged kernel is bracketed by
 * hipEvents on the caller's stream; ifl_profile_collect() waits for them, returns the summed
 * device milliseconds and the launch count for `tag`, and forgets them.
 */
enum { IFL_PROF_SCAN = 0, IFL_PROF_WGRAD = 1, IFL_PROF_CONV = 2, IFL_PROF_FOLD = 3, IFL_PROF_FALLBACK = 4 };
void ifl_profile_enable(int on);
int ifl_profile_collect(int tag, double *total_ms, int *launches);

/* Scratch bytes the given op needs for this shape (0 is possible).  Negative sizes -> 0. */
size_t ifl_workspace_bytes(int op, int B, int C, int H, int W, int KH, int KW, unsigned flags);

/*
 * Persistent block of the two-workgroup scan.  Layers of 32 or 64 channels on 32-pixel rows with 17..32 rows run two
 * workgroups per image (at most half as many images as compute units) that hand two rows per anti-diagonal over through
 * a mailbox in this block.  The caller owns it: ifl_scan_state_bytes() bytes, 256-byte aligned, ZERO-FILLED once, then
 * passed unchanged to every scan call of ONE stream (launches that use a block must not overlap in time; its contents --
 * the mailbox and per-image launch generations -- advance on the device, so it is valid under graph replay).
 * scan_state = NULL: one workgroup per image; results are bit-identical either way.
 *
 * The same block serves the wide layers (64 < C <= 256, H <= 16, W <= 32): there a team of C/16 workgroups sweeps all
 * anti-diagonals of 16 image columns inside ONE launch and synchronises through generation-numbered flags kept here;
 * scan_state = NULL (or IFL_FLAG_WHOLE_IMAGE): one launch per anti-diagonal instead.  A launch whose team cannot finish in
 * bounded time, or whose values leave the fp16 range, is redone by the exact fp32 scan within the same call; the block
 * counts such launches in a 64-bit word at byte offset ifl_scan_state_voided_offset() (telemetry, read-only for callers).
 */
size_t ifl_scan_state_bytes(void);
size_t ifl_scan_state_voided_offset(void);

/*
 * z = A^-1 x -- the layer's forward pass x -> z.
 * Replaces  inv_conv_with_bp.inverse(input, kernel, output)
 *   (inv_conv_with_bp_general.cpp:19-28 -> inv_conv_cuda_inverse, inv_conv_with_bp_kernel_general.cu:72-129),
 * called from inv_conv_.forward (inf/layers/inv_conv.py:46-60).  Exact semantics = solve_mc.py:88-114.
 * z must not alias x (the scan re-reads x when an image has to be redone in a wider arithmetic).
 */
int ifl_inverse_f32(const float *x, const float *w, float *z, int B, int C, int H, int W, int KH, int KW,
                    int order, unsigned flags, void *ws, size_t ws_bytes, void *carry, void *scan_state,
                    ifl_stream_t stream);

/*
 * Optional forward -> backward side channel (the analogue of ctx.save_for_backward, inf/layers/inv_conv.py:56):
 * a caller-owned device buffer of ifl_carry_bytes() that ifl_inverse_f32 fills with the folded weights of
 * the adjoint operator and max|z|, and that ifl_backward_f32 of the SAME step (same w, shape, order, flags)
 * reads instead of folding again and scanning z / dx for their maxima.  Pass NULL to either call to
 * disable it; results are identical either way.
 */
size_t ifl_carry_bytes(int C, int KH, int KW);

/*
 * xhat = A z (+ log|det A| per image) -- the layer's reverse / sampling / reconstruction pass.
 * Replaces  inv_conv_with_bp.forward(input, kernel, output)
 *   (inv_conv_with_bp_general.cpp:44-53 -> inv_conv_fwd_cuda_inverse, .cu:203-264),
 * called from inv_flow_*.reverse (inf/layers/inv_conv.py:249-267,442-460).
 * `logdet` (B floats) may be NULL; it receives H*W*sum_c log|w[c,c,diag tap]|
 * (inf/layers/emerging/emerging_module.py:26-32) or 0 for the unit diagonal
 * (inf/layers/inv_conv.py:221).
 */
int ifl_forward_f32(const float *z, const float *w, float *xhat, float *logdet, int B, int C, int H, int W,
                    int KH, int KW, int order, unsigned flags, void *ws, size_t ws_bytes, ifl_stream_t stream);

/*
 * Fused backward:  dx = A^-T g,  dw = -(sum_b,h,w dx (x) shifted z) * mask   [+ recon term].
 * Replaces the pair inv_conv_with_bp.dy(...) + inv_conv_with_bp.dw(...)
 *   (inv_conv_with_bp_general.cpp:70-81,99-112; .cu:408-483,660-735) as called from
 *   inv_conv_.backward (inf/layers/inv_conv.py:62-81), and the mask of
 *   inv_flow_with_pad.reset_gradients (inf/layers/inv_conv.py:223-230).
 * `z` is the saved layer output.  If recon_weight != 0 and `x` != NULL the gradient of
 *   recon_weight * mean_b ||x - A z||^2   w.r.t. w (z detached; cf. add_recon_grad,
 *   inf/layers/selfnorm.py:187-229) is accumulated in the same dW reduction and
 *   `recon_loss` (1 float, may be NULL) receives mean_b ||x - A z||^2.
 * `dx` may be NULL (weights-only) or `dw` may be NULL (input-gradient only); dx must not alias g.
 */
int ifl_backward_f32(const float *g, const float *z, const float *x, const float *w, float *dx, float *dw,
                     float recon_weight, float *recon_loss, int B, int C, int H, int W, int KH, int KW,
                     int order, unsigned flags, void *ws, size_t ws_bytes, void *carry, void *scan_state,
                     ifl_stream_t stream);

/*
 * bf16 storage variants (the reference dispatches its kernels on the tensor's dtype, inv_conv_with_bp_kernel_general.cu:112;
 * "bf16 storage / fp32 accumulate").  Activations -- x, z, g, dx, xhat -- are bf16 (their 16-bit patterns);
 * weights, weight gradients, log-determinants, the recon loss and all arithmetic are as in the f32 entry points: a call
 * returns exactly the round-to-nearest-even bf16 of what the f32 call returns on the widened inputs.  Arguments as there;
 * workspace from ifl_workspace_bytes_bf16 (it also holds the fp32 images the matrix-pipe kernels work on).
 */
size_t ifl_workspace_bytes_bf16(int op, int B, int C, int H, int W, int KH, int KW, unsigned flags);
int ifl_inverse_bf16(const uint16_t *x, const float *w, uint16_t *z, int B, int C, int H, int W, int KH, int KW,
                     int order, unsigned flags, void *ws, size_t ws_bytes, void *carry, void *scan_state,
                     ifl_stream_t stream);
int ifl_forward_bf16(const uint16_t *z, const float *w, uint16_t *xhat, float *logdet, int B, int C, int H, int W,
                     int KH, int KW, int order, unsigned flags, void *ws, size_t ws_bytes, ifl_stream_t stream);
int ifl_backward_bf16(const uint16_t *g, const uint16_t *z, const uint16_t *x, const float *w, uint16_t *dx, float *dw,
                      float recon_weight, float *recon_loss, int B, int C, int H, int W, int KH, int KW,
                      int order, unsigned flags, void *ws, size_t ws_bytes, void *carry, void *scan_state,
                      ifl_stream_t stream);

/*
 * The inverse-flow block: four layers of orders TL -> TR -> BL -> BR (Inv_FlowUnit, inf/layers/inv_flow.py:13-53).
 *   z[0] = A_TL^-1 x, z[1] = A_TR^-1 z[0], z[2] = A_BL^-1 z[1], z[3] = A_BR^-1 z[2]   (all kept: the backward needs them)
 * ONE fold launch serves the four layers (and, with carries, their adjoints), the four scans follow back to back.
 * w[l], z[l], dw[l], carry[l] (NULL or four buffers of ifl_carry_bytes) are host arrays of four device pointers.
 * ifl_unit_backward_f32: dx = dL/dx and dw[l] = dL/dw[l] from gout = dL/dz[3] (BR -> BL -> TR -> TL).
 * Workspace: ifl_unit_workspace_bytes(IFL_OP_INVERSE / IFL_OP_BACKWARD, ...).
 */
size_t ifl_unit_workspace_bytes(int op, int B, int C, int H, int W, int KH, int KW, unsigned flags);
int ifl_unit_inverse_f32(const float *x, const float *const w[4], float *const z[4], int B, int C, int H, int W,
                         int KH, int KW, unsigned flags, void *ws, size_t ws_bytes, void *const carry[4],
                         void *scan_state, ifl_stream_t stream);
int ifl_unit_backward_f32(const float *gout, const float *const z[4], const float *const w[4], float *dx,
                          float *const dw[4], int B, int C, int H, int W, int KH, int KW, unsigned flags, void *ws,
                          size_t ws_bytes, void *const carry[4], void *scan_state, ifl_stream_t stream);

/*
 * Weight gradient from a precomputed dx:  dw = -(sum dx (x) shifted z) * mask.
 * Second half of inv_conv_with_bp.dw (inv_conv_with_bp_general.cpp:99-112).
 */
int ifl_dw_f32(const float *z, const float *dx, float *dw, int B, int C, int H, int W, int KH, int KW,
               int order, unsigned flags, void *ws, size_t ws_bytes, ifl_stream_t stream);

/* ---- dense convolution pieces of SelfNormConv (inf/layers/selfnorm.py:39-90) ------------------
 * stride 1, dilation 1, groups 1, symmetric zero padding (ph, pw):
 *   x : (B, Ci, H, W)   w : (Co, Ci, KH, KW)   z, gz : (B, Co, OH, OW),  OH = H + 2ph - KH + 1.
 */

/* z = conv2d(x, w) + bias  (bias may be NULL) -- F.conv2d at inf/layers/selfnorm.py:43. */
int ifl_conv2d_f32(const float *x, const float *w, const float *bias, float *z, int B, int Ci, int Co, int H,
                   int W, int KH, int KW, int ph, int pw, void *ws, size_t ws_bytes, ifl_stream_t stream);

/* dw = cudnn_convolution_backward_weight(gz, x) -- inf/utils/convbackward/conv2d_backward.cpp:7-28. */
int ifl_conv2d_wgrad_f32(const float *gz, const float *x, float *dw, int B, int Ci, int Co, int H, int W,
                         int KH, int KW, int ph, int pw, void *ws, size_t ws_bytes, ifl_stream_t stream);

/* dx = cudnn_convolution_backward_input(gz, w) -- inf/utils/convbackward/conv2d_backward.cpp:32-53. */
int ifl_conv2d_igrad_f32(const float *gz, const float *w, float *dx, int B, int Ci, int Co, int H, int W,
                         int KH, int KW, int ph, int pw, void *ws, size_t ws_bytes, ifl_stream_t stream);

/* scratch for the three calls above (ifl_conv2d_f32 also runs without: ws = NULL selects the direct kernel) */
size_t ifl_conv2d_workspace_bytes(int B, int Ci, int Co, int H, int W, int KH, int KW, int ph, int pw);

/* ---- Glow-step neighbours of the layer: one pass over an NCHW activation each ------------
 * Replaces the eager torch elementwise graphs of inf/layers/actnorm.py, squeeze.py and coupling.py (the
 * conditioner network of the coupling stays a library convolution stack: it is handed in as h).
 */
size_t ifl_glow_workspace_bytes(int B, int C); /* scratch of the calls below that take ws */

/* ActNorm (inf/layers/actnorm.py:18-69).  forward: y = (x - t_c) exp(-ls_c), logdet[b] = -H W sum_c ls_c
 * (logdet may be NULL); reverse != 0: y = x exp(ls_c) + t_c (logdet untouched).  y may alias x. */
int ifl_actnorm_f32(const float *x, const float *translation, const float *log_scale, float *y, float *logdet, int B,
                    int C, int H, int W, int reverse, ifl_stream_t stream);
/* backward of the forward direction: gx = gy exp(-ls); g_translation_c = -exp(-ls_c) sum gy;
 * g_log_scale_c = -sum gy y - H W sum_b g_logdet[b]  (g_logdet, g_translation, g_log_scale may be NULL). */
int ifl_actnorm_backward_f32(const float *gy, const float *g_logdet, const float *x, const float *translation,
                             const float *log_scale, float *gx, float *g_translation, float *g_log_scale, int B, int C,
                             int H, int W, void *ws, size_t ws_bytes, ifl_stream_t stream);
/* data-dependent initialisation (actnorm.py:21-28): mean_c, log(std_c + 1e-8) over (B, H, W), std unbiased. */
int ifl_actnorm_stats_f32(const float *x, float *mean, float *log_std, int B, int C, int H, int W, void *ws,
                          size_t ws_bytes, ifl_stream_t stream);

/* Squeeze (inf/layers/squeeze.py:5-25).  (C, H, W) is the LARGE layout.  reverse = 0: space_to_depth,
 * x (B,C,H,W) -> y (B,4C,H/2,W/2); reverse != 0: depth_to_space, x (B,4C,H/2,W/2) -> y (B,C,H,W). */
int ifl_squeeze_f32(const float *x, float *y, int B, int C, int H, int W, int reverse, ifl_stream_t stream);

/* Coupling, affine part (inf/layers/coupling.py:66-98).  h = net(x1) (B,C,H,W): h_s = h[:,0::2], t = h[:,1::2],
 * log_s = 2 tanh(h_s/2).  forward: y = cat(x1, x2 exp(log_s) + t), logdet[b] = sum log_s (may be NULL);
 * reverse != 0: y = cat(x1, (x2 - t) exp(-log_s)).  y may alias x. */
int ifl_coupling_f32(const float *x, const float *h, float *y, float *logdet, int B, int C, int H, int W, int reverse,
                     void *ws, size_t ws_bytes, ifl_stream_t stream);
/* backward of the forward direction: gx = cat(gy1, gy2 exp(log_s)) (the path through the net is the caller's),
 * gh[:,1::2] = gy2, gh[:,0::2] = (gy2 x2 exp(log_s) + g_logdet[b]) (1 - tanh^2(h_s/2)).  g_logdet may be NULL. */
int ifl_coupling_backward_f32(const float *gy, const float *g_logdet, const float *x, const float *h, float *gx, float *gh,
                              int B, int C, int H, int W, ifl_stream_t stream);

/* bf16 storage variants of the five calls above: activations (x, y, h, gy, gx, gh) as bf16, parameters, their
 * gradients and log-determinants fp32, arithmetic and summation order as in the f32 calls (a call returns exactly the
 * rounded result of the f32 call on the widened inputs; log-determinants and parameter gradients are bit-identical).
 * One pass each, at half the bytes.  (ifl_actnorm_stats_f32 -- one call in a model's lifetime -- has no bf16 form.) */
int ifl_actnorm_bf16(const uint16_t *x, const float *translation, const float *log_scale, uint16_t *y, float *logdet, int B,
                     int C, int H, int W, int reverse, ifl_stream_t stream);
int ifl_actnorm_backward_bf16(const uint16_t *gy, const float *g_logdet, const uint16_t *x, const float *translation,
                              const float *log_scale, uint16_t *gx, float *g_translation, float *g_log_scale, int B, int C,
                              int H, int W, void *ws, size_t ws_bytes, ifl_stream_t stream);
int ifl_squeeze_bf16(const uint16_t *x, uint16_t *y, int B, int C, int H, int W, int reverse, ifl_stream_t stream);
int ifl_coupling_bf16(const uint16_t *x, const uint16_t *h, uint16_t *y, float *logdet, int B, int C, int H, int W,
                      int reverse, void *ws, size_t ws_bytes, ifl_stream_t stream);
int ifl_coupling_backward_bf16(const uint16_t *gy, const float *g_logdet, const uint16_t *x, const uint16_t *h, uint16_t *gx,
                               uint16_t *gh, int B, int C, int H, int W, ifl_stream_t stream);

/* ---- the conditioner of the affine coupling (inf/layers/coupling.py:47-62 `Coupling.net`, :9-45 Conv2dZero) --------
 * h = (W3 * relu(W2 relu(W1 * x1)) + b3) exp(logscale_factor logs): 3x3 conv C/2 -> width (no bias), ReLU, 1x1 conv
 * width -> C (no bias), ReLU, 3x3 conv C -> C with bias and a per-channel gain, zero padding 1.  x1 = the first C/2
 * channels of x (an NCHW tensor of x_channels >= C/2 channels: the coupling's input as it is, no slice copy); everything
 * fp32 NCHW.  Two launches forward, six backward; a model's step
 * calls this per coupling in place of some fifty library-convolution launches.  C in {4, 8, 12, 16, 24, 32, 48}, width a
 * multiple of 16 (ifl_cond_supported); other shapes: IFL_EUNSUPPORTED, and the host layer keeps its convolutions.
 *
 * ifl_cond_prep_f32: once per weight update -- transposed copies of the kernels and the gain, ifl_cond_weights_floats(C,
 * width) floats (w1 [width][C/2][3][3], w2 [C][width], w3 [C][C][3][3], logs [C]).
 * ifl_cond_forward_f32 (wt from ifl_cond_prep_f32 plus the kernels w1, w2 as they are): a2 [B][C][H][W] (the second ReLU's output, kept for the backward) and h [B][C][H][W].
 * ifl_cond_backward_f32 (dh = gradient of h):
 *   dx[:, :C/2] += the input gradient (dx: [B][x_channels][H][W], already holding the coupling's direct part);
 *   grads [ifl_cond_grads_floats(C, width)] = dW1 [width][C/2][3][3] | dW2 [C][width] | dW3 [C][C][3][3] | d logs [C] | d b3 [C];
 *   ws: ifl_cond_backward_workspace_bytes(...) bytes, 256-aligned (the operand matrices of the weight-gradient products,
 *   pixel-major, and the partial sums).  operands_f32 = 0: those matrices are bf16 (fp32 accumulate: the precision of a
 *   bf16 autocast step); 1: fp32 (an fp32 step keeps fp32-accurate gradients).  Six launches.
 * No float atomics anywhere: results are reproducible bit for bit. */
int ifl_cond_supported(int C, int width);
size_t ifl_cond_weights_floats(int C, int width);
int ifl_cond_pixels_padded(int B, int H, int W);
int ifl_cond_prep_f32(const float *w1, const float *w2, const float *w3, const float *logs, float *wt, int C, int width,
                      float logscale_factor, ifl_stream_t stream);

/* The same for the couplings of a whole model in ONE launch: `jobs` is a DEVICE array of n_jobs entries (the pointers of
 * an entry as for ifl_cond_prep_f32; every (C, width) must pass ifl_cond_supported -- the caller checks, the kernel
 * cannot), max_weights_floats = the largest ifl_cond_weights_floats among them.  A training step calls it once before its
 * forward pass; the table is built once (the parameters' addresses do not move). */
typedef struct ifl_cond_prep_job {
    const float *w1, *w2, *w3, *logs;
    float *wt;
    int C, width;
    float logscale_factor;
    int reserved;
} ifl_cond_prep_job;
int ifl_cond_prep_many_f32(const ifl_cond_prep_job *jobs, int n_jobs, size_t max_weights_floats, ifl_stream_t stream);
int ifl_cond_forward_f32(const float *x, int x_channels, const float *wt, const float *w1, const float *w2, const float *b3, float *a2,
                         float *h, int B, int C, int H, int W, int width, ifl_stream_t stream);
size_t ifl_cond_backward_workspace_bytes(int B, int C, int H, int W, int width, int operands_f32);
size_t ifl_cond_grads_floats(int C, int width);
int ifl_cond_backward_f32(const float *x, int x_channels, const float *dh, const float *h, const float *a2, const float *wt,
                          const float *w1, int operands_f32, void *ws, size_t ws_bytes, float *grads, float *dx, int B, int C, int H,
                          int W, int width, float logscale_factor, ifl_stream_t stream);

/* ---- optimizer step over a flat parameter buffer (the train step: inf/train/experiment.py:272-311 calls optimizer.step()) ----
 * Adam (decoupled = 0: weight decay added to the gradient) / AdamW (decoupled = 1) with torch.optim's arithmetic on n
 * contiguous floats: p, m = exp_avg, v = exp_avg_sq updated in place from g.  *lr and *step (the number of this step, from 1,
 * as a float) are read on the device.  One elementwise pass where a multi-tensor optimizer takes a launch per 36 tensors. */
int ifl_adam_flat_f32(float *p, const float *g, float *m, float *v, size_t n, const float *lr, const float *step, float beta1,
                      float beta2, float eps, float weight_decay, int decoupled, ifl_stream_t stream);

/* ---- activations of the Glow step (inf/layers/activations.py) ----------------------------------------------------- */
size_t ifl_activation_workspace_bytes(int B, int C, int n_bins); /* scratch of the calls below (n_bins = 0: SmoothLeakyRelu) */

/* SmoothLeakyRelu (activations.py:37-54): y = alpha x + (1 - alpha) log(1 + e^x), logdet[b] = sum log y' (may be NULL);
 * reverse != 0: the reference's Newton-Raphson inverse (100 iterations from x0 = y, slope clamped at 1e-2:
 * activations.py:27-34), iterated in registers.  y may alias x. */
int ifl_slr_f32(const float *x, float *y, float *logdet, int B, int C, int H, int W, float alpha, int reverse, void *ws,
                size_t ws_bytes, ifl_stream_t stream);
/* gx = gy y' + g_logdet[b] y'' / y'  (g_logdet may be NULL) */
int ifl_slr_backward_f32(const float *gy, const float *g_logdet, const float *x, float *gx, int B, int C, int H, int W,
                         float alpha, ifl_stream_t stream);

/* SplineActivation with shared weights (activations.py:126-217): the monotone rational-quadratic spline with linear
 * tails of inf/layers/splines/rational_quadratic.py:20-175 on knot tables cw, ch (positions, cw[0] = ch[0] =
 * -tail_bound, cw[n_bins] = ch[n_bins] = +tail_bound) and dv (derivatives): DEVICE arrays of n_bins + 1 floats, computed
 * from the layer's parameters by the caller (rational_quadratic.py:97-116; no host round trip).  n_bins <= 16 (per-element form: <= 8).
 * inverse = 0: y = spline(x), logdet[b] = sum log|dy/dx| (may be NULL);  inverse != 0: y = spline^-1(x), logdet[b] =
 * sum log|dy/dx| of the inverse map. */
/* the knot tables from the layer's parameters (DEVICE vectors: unnormalized widths and heights of n_bins entries,
 * derivatives of n_bins - 1), rational_quadratic.py:35-46,97-116, and the gradients of the parameters from those of the
 * tables (g_tables as ifl_rqspline_backward_f32 returns them) */
int ifl_rqspline_tables_f32(const float *uw, const float *uh, const float *ud, int n_bins, float tail_bound, float *cw,
                            float *ch, float *dv, ifl_stream_t stream);
int ifl_rqspline_tables_backward_f32(const float *g_tables, const float *uw, const float *uh, const float *ud, int n_bins,
                                     float tail_bound, float *g_uw, float *g_uh, float *g_ud, ifl_stream_t stream);
int ifl_rqspline_f32(const float *x, const float *cw, const float *ch, const float *dv, int n_bins, float tail_bound, float *y,
                     float *logdet, int B, int C, int H, int W, int inverse, void *ws, size_t ws_bytes, ifl_stream_t stream);
/* backward of the forward direction: gx and g_tables = d loss / d (cw, ch, dv): DEVICE array of 3 (n_bins + 1) floats */
int ifl_rqspline_backward_f32(const float *gy, const float *g_logdet, const float *x, const float *cw, const float *ch,
                              const float *dv, int n_bins, float tail_bound, float *gx, float *g_tables, int B, int C, int H,
                              int W, void *ws, size_t ws_bytes, ifl_stream_t stream);

/* The same spline straight from its parameters (unnormalized widths / heights: n_bins floats, derivatives: n_bins - 1): the
 * knot tables are computed inside the launch -- one launch less each way than ifl_rqspline_tables_f32 + ifl_rqspline_f32 --
 * and kept in `tables` (3 (n_bins + 1) floats: cw | ch | dv) for the backward, which returns the parameters' gradients.
 * Same arithmetic, same bits as the two-call form. */
int ifl_rqspline_p_f32(const float *x, const float *uw, const float *uh, const float *ud, int n_bins, float tail_bound, float *y,
                       float *logdet, float *tables, int B, int C, int H, int W, int inverse, void *ws, size_t ws_bytes,
                       ifl_stream_t stream);
int ifl_rqspline_p_backward_f32(const float *gy, const float *g_logdet, const float *x, const float *tables, const float *uw,
                                const float *uh, const float *ud, int n_bins, float tail_bound, float *gx, float *g_uw, float *g_uh,
                                float *g_ud, int B, int C, int H, int W, void *ws, size_t ws_bytes, ifl_stream_t stream);

/* SplineActivation with INDIVIDUAL weights (activations.py:135-144: one set of knots per element; parameters of shape
 * (1, C, H, W, n_bins) -- here P = C*H*W element positions, DEVICE arrays uw, uh of P*n_bins floats and ud of P*(n_bins-1)).
 * x, y: (B, P).  The knot tables are built inside the kernels.  inverse / logdet as for ifl_rqspline_f32.
 * Backward of the forward direction: gx (B, P) and g_params = [g_uw (P*n_bins) | g_uh (P*n_bins) | g_ud (P*(n_bins-1))],
 * summed over the batch in a fixed order.  ws: ifl_rqspline_pe_workspace_bytes. */
size_t ifl_rqspline_pe_workspace_bytes(int B, int P, int n_bins);
int ifl_rqspline_pe_f32(const float *x, const float *uw, const float *uh, const float *ud, int n_bins, float tail_bound, float *y,
                        float *logdet, int B, int P, int inverse, void *ws, size_t ws_bytes, ifl_stream_t stream);
int ifl_rqspline_pe_backward_f32(const float *gy, const float *g_logdet, const float *x, const float *uw, const float *uh,
                                 const float *ud, int n_bins, float tail_bound, float *gx, float *g_params, int B, int P, void *ws,
                                 size_t ws_bytes, ifl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* INVFLOW_H */
