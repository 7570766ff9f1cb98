/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the inverse-of-convolution hot path.
 * Included twice by invflow_oracle.c with REAL/SUF defined (float/_f32, double/_f64).
 *
 * Every loop below restates the reference's *exact CPU* algorithm; the citation next to
 * each function names the reference file:line it follows (paths relative to the
 * reference checkout).  Nothing here is ever linked into, imported by, or called from the
 * product path (inverse-flow_amd/): only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * Conventions:
 *   x, z, g : (B, C, H, W) contiguous NCHW
 *   w       : (C, C, KH, KW) = [c_out, c_in, kh, kw]
 *   diag    : 0 = unit diagonal tap, upper-channel part of the last tap ignored
 *                 (solve_mc.py:105-109 `continue` / `break`)
 *             1 = general diagonal: divide by w[c,c,KH-1,KW-1]
 *                 (inverse_op_cython.pyx:64), upper part still ignored (it multiplies
 *                 not-yet-written zeros in inverse_op_cython.pyx:62).
 *   All functions are the TL ("top-left padded") order; the other orders are obtained by
 *   the caller through flips exactly as inf/layers/conv.py:192-219 does.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* effective weight of the diagonal tap: Ŵ[c,kc,KH-1,KW-1] */
static inline REAL FN(eff_diag_w)(const REAL *w, int C, int KH, int KW, int c, int kc, int diag)
{
    if (kc > c) return (REAL)0;
    if (kc == c) return diag ? w[(((size_t)c * C + kc) * KH + (KH - 1)) * KW + (KW - 1)] : (REAL)1;
    return w[(((size_t)c * C + kc) * KH + (KH - 1)) * KW + (KW - 1)];
}

/*
 * z = A^-1 x by raster-order back-substitution.
 * Loop order b,h,w,c,k_h,k_w,k_c with the same `break`/`continue` structure as
 * inf/utils/solve_mc.py:88-114 (`solve`); diag=1 adds the final division of
 * inf/layers/emerging/inverse_op_cython.pyx:62-64.
 */
void FN(orc_inverse)(const REAL *x, const REAL *w, REAL *z, int B, int C, int H, int W,
                     int KH, int KW, int diag, int nthreads)
{
    const size_t img = (size_t)C * H * W;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int b = 0; b < B; ++b) {
        const REAL *xb = x + b * img;
        REAL *y = z + b * img;
        for (int h = 0; h < H; ++h)
            for (int ww = 0; ww < W; ++ww)
                for (int c = 0; c < C; ++c) {
                    REAL acc = xb[((size_t)c * H + h) * W + ww];
                    for (int kh = 0; kh < KH; ++kh) {
                        if (h - kh < 0) break;
                        for (int kw = 0; kw < KW; ++kw) {
                            if (ww - kw < 0) break;
                            for (int kc = 0; kc < C; ++kc) {
                                if (kh == 0 && kw == 0) {
                                    if (kc == c) continue;
                                    if (c - kc < 0) break;
                                }
                                acc -= y[((size_t)kc * H + (h - kh)) * W + (ww - kw)] *
                                       w[(((size_t)c * C + kc) * KH + (KH - kh - 1)) * KW + (KW - kw - 1)];
                            }
                        }
                    }
                    if (diag) acc /= w[(((size_t)c * C + c) * KH + (KH - 1)) * KW + (KW - 1)];
                    y[((size_t)c * H + h) * W + ww] = acc;
                }
    }
}

/*
 * xhat = A z : TL-padded masked convolution,
 * F.conv2d(F.pad(z,(KW-1,0,KH-1,0)), What) -- inf/layers/conv.py:103-108 (pad tuple
 * conv.py:42-44) with What the masked weight of inf/layers/inv_conv.py:233-248.
 */
void FN(orc_forward)(const REAL *zin, const REAL *w, REAL *xo, int B, int C, int H, int W,
                     int KH, int KW, int diag, int nthreads)
{
    const size_t img = (size_t)C * H * W;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int b = 0; b < B; ++b) {
        const REAL *zb = zin + b * img;
        REAL *xb = xo + b * img;
        for (int c = 0; c < C; ++c)
            for (int h = 0; h < H; ++h)
                for (int ww = 0; ww < W; ++ww) {
                    REAL acc = 0;
                    for (int kh = 0; kh < KH && h - kh >= 0; ++kh)
                        for (int kw = 0; kw < KW && ww - kw >= 0; ++kw)
                            for (int kc = 0; kc < C; ++kc) {
                                REAL wt = (kh == 0 && kw == 0)
                                              ? FN(eff_diag_w)(w, C, KH, KW, c, kc, diag)
                                              : w[(((size_t)c * C + kc) * KH + (KH - kh - 1)) * KW + (KW - kw - 1)];
                                acc += wt * zb[((size_t)kc * H + (h - kh)) * W + (ww - kw)];
                            }
                    xb[((size_t)c * H + h) * W + ww] = acc;
                }
    }
}

/*
 * u = A^-T g : the adjoint of `solve` (autograd backward of inv_conv_.forward,
 * inf/layers/inv_conv.py:62-74 *as the math requires*, SURVEY section 0 row `dy`):
 * anti-causal back-substitution in reversed raster order,
 *   u[c,h,w] = (g[c,h,w] - sum What[kc,c,KH-1-dh,KW-1-dw] u[kc,h+dh,w+dw]) / What[c,c,last].
 * Verified against torch.autograd through a dense torch.linalg.solve of A
 * (the reference's compute_expensive recipe, inf/layers/selfnorm.py:175-180).
 */
void FN(orc_dy)(const REAL *g, const REAL *w, REAL *u, int B, int C, int H, int W, int KH,
                int KW, int diag, int nthreads)
{
    const size_t img = (size_t)C * H * W;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int b = 0; b < B; ++b) {
        const REAL *gb = g + b * img;
        REAL *ub = u + b * img;
        for (int h = H - 1; h >= 0; --h)
            for (int ww = W - 1; ww >= 0; --ww)
                for (int c = C - 1; c >= 0; --c) {
                    REAL acc = gb[((size_t)c * H + h) * W + ww];
                    for (int kh = 0; kh < KH && h + kh < H; ++kh)
                        for (int kw = 0; kw < KW && ww + kw < W; ++kw)
                            for (int kc = 0; kc < C; ++kc) {
                                REAL wt;
                                if (kh == 0 && kw == 0) {
                                    if (kc <= c) continue; /* What[kc,c] needs c < kc */
                                    wt = w[(((size_t)kc * C + c) * KH + (KH - 1)) * KW + (KW - 1)];
                                } else {
                                    wt = w[(((size_t)kc * C + c) * KH + (KH - kh - 1)) * KW + (KW - kw - 1)];
                                }
                                acc -= wt * ub[((size_t)kc * H + (h + kh)) * W + (ww + kw)];
                            }
                    if (diag) acc /= w[(((size_t)c * C + c) * KH + (KH - 1)) * KW + (KW - 1)];
                    ub[((size_t)c * H + h) * W + ww] = acc;
                }
    }
}

/*
 * dW[c,kc,KH-1-dh,KW-1-dw] = - sum_{b,h,w} u[b,c,h,w] z[b,kc,h-dh,w-dw], then the mask of
 * inf/layers/inv_conv.py:223-248 (`reset_gradients`/`get_mask`): the diagonal tap keeps
 * only kc<c (diag=0) or kc<=c (diag=1).  SURVEY section 0 row `dw`.
 * Accumulated in double regardless of REAL; per-thread partials are reduced in a fixed
 * order so the result does not depend on nthreads scheduling.
 */
void FN(orc_dw)(const REAL *zin, const REAL *u, REAL *dw, int B, int C, int H, int W, int KH,
                int KW, int diag, int nthreads)
{
    const size_t img = (size_t)C * H * W;
    const size_t nw = (size_t)C * C * KH * KW;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int c = 0; c < C; ++c)
        for (int kc = 0; kc < C; ++kc)
            for (int kh = 0; kh < KH; ++kh)
                for (int kw = 0; kw < KW; ++kw) {
                    double acc = 0;
                    int masked = 0;
                    if (kh == 0 && kw == 0) masked = diag ? (kc > c) : (kc >= c);
                    if (!masked)
                        for (int b = 0; b < B; ++b) {
                            const REAL *zb = zin + b * img + (size_t)kc * H * W;
                            const REAL *ub = u + b * img + (size_t)c * H * W;
                            for (int h = kh; h < H; ++h)
                                for (int ww = kw; ww < W; ++ww)
                                    acc += (double)ub[(size_t)h * W + ww] *
                                           (double)zb[(size_t)(h - kh) * W + (ww - kw)];
                        }
                    dw[(((size_t)c * C + kc) * KH + (KH - kh - 1)) * KW + (KW - kw - 1)] = (REAL)(-acc);
                }
    (void)nw;
}

/*
 * log|det A| per image = H*W*sum_c log|W[c,c,KH-1,KW-1]| (0 for the unit diagonal) --
 * inf/layers/emerging/emerging_module.py:26-32 (`delta_ldj`); inv_flow_* return 0.0
 * (inf/layers/inv_conv.py:221,440).
 */
double FN(orc_logdet)(const REAL *w, int C, int H, int W, int KH, int KW, int diag)
{
    if (!diag) return 0.0;
    double s = 0;
    for (int c = 0; c < C; ++c) {
        double d = (double)w[(((size_t)c * C + c) * KH + (KH - 1)) * KW + (KW - 1)];
        s += log(d < 0 ? -d : d);
    }
    return s * H * W;
}

/* ------------------------------------------------------------------------------------ *
 * Self-normalising convolution: dense "same"-style conv with
 * symmetric zero padding (ph,pw), stride 1, dilation 1, groups 1.
 * ------------------------------------------------------------------------------------ */

/* z = conv2d(x, W) + b  -- inf/layers/selfnorm.py:42-50 (SelfNormConvFunc.forward). */
void FN(orc_conv2d)(const REAL *x, const REAL *w, const REAL *bias, REAL *z, int B, int Ci, int Co,
                    int H, int W, int KH, int KW, int ph, int pw, int nthreads)
{
    const int OH = H + 2 * ph - KH + 1, OW = W + 2 * pw - KW + 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Co; ++co)
            for (int oh = 0; oh < OH; ++oh)
                for (int ow = 0; ow < OW; ++ow) {
                    double acc = bias ? (double)bias[co] : 0.0;
                    for (int ci = 0; ci < Ci; ++ci)
                        for (int kh = 0; kh < KH; ++kh) {
                            int ih = oh - ph + kh;
                            if (ih < 0 || ih >= H) continue;
                            for (int kw = 0; kw < KW; ++kw) {
                                int iw = ow - pw + kw;
                                if (iw < 0 || iw >= W) continue;
                                acc += (double)w[(((size_t)co * Ci + ci) * KH + kh) * KW + kw] *
                                       (double)x[(((size_t)b * Ci + ci) * H + ih) * W + iw];
                            }
                        }
                    z[(((size_t)b * Co + co) * OH + oh) * OW + ow] = (REAL)acc;
                }
}

/*
 * dW[co,ci,kh,kw] = sum_{b,oh,ow} gz[b,co,oh,ow] x[b,ci,oh-ph+kh,ow-pw+kw]
 * = cudnn_convolution_backward_weight -- inf/utils/convbackward/conv2d_backward.cpp:7-28,
 * called at inf/layers/selfnorm.py:63-66,77-80.
 */
void FN(orc_conv2d_wgrad)(const REAL *gz, const REAL *x, REAL *dw, int B, int Ci, int Co, int H,
                          int W, int KH, int KW, int ph, int pw, int nthreads)
{
    const int OH = H + 2 * ph - KH + 1, OW = W + 2 * pw - KW + 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci)
            for (int kh = 0; kh < KH; ++kh)
                for (int kw = 0; kw < KW; ++kw) {
                    double acc = 0;
                    for (int b = 0; b < B; ++b)
                        for (int oh = 0; oh < OH; ++oh) {
                            int ih = oh - ph + kh;
                            if (ih < 0 || ih >= H) continue;
                            for (int ow = 0; ow < OW; ++ow) {
                                int iw = ow - pw + kw;
                                if (iw < 0 || iw >= W) continue;
                                acc += (double)gz[(((size_t)b * Co + co) * OH + oh) * OW + ow] *
                                       (double)x[(((size_t)b * Ci + ci) * H + ih) * W + iw];
                            }
                        }
                    dw[(((size_t)co * Ci + ci) * KH + kh) * KW + kw] = (REAL)acc;
                }
}

/*
 * dx[b,ci,ih,iw] = sum_{co,kh,kw} W[co,ci,kh,kw] gz[b,co,ih+ph-kh,iw+pw-kw]
 * = cudnn_convolution_backward_input -- inf/utils/convbackward/conv2d_backward.cpp:32-53,
 * called at inf/layers/selfnorm.py:73-76.
 */
void FN(orc_conv2d_igrad)(const REAL *gz, const REAL *w, REAL *dx, int B, int Ci, int Co, int H,
                          int W, int KH, int KW, int ph, int pw, int nthreads)
{
    const int OH = H + 2 * ph - KH + 1, OW = W + 2 * pw - KW + 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int ci = 0; ci < Ci; ++ci)
            for (int ih = 0; ih < H; ++ih)
                for (int iw = 0; iw < W; ++iw) {
                    double acc = 0;
                    for (int co = 0; co < Co; ++co)
                        for (int kh = 0; kh < KH; ++kh) {
                            int oh = ih + ph - kh;
                            if (oh < 0 || oh >= OH) continue;
                            for (int kw = 0; kw < KW; ++kw) {
                                int ow = iw + pw - kw;
                                if (ow < 0 || ow >= OW) continue;
                                acc += (double)w[(((size_t)co * Ci + ci) * KH + kh) * KW + kw] *
                                       (double)gz[(((size_t)b * Co + co) * OH + oh) * OW + ow];
                            }
                        }
                    dx[(((size_t)b * Ci + ci) * H + ih) * W + iw] = (REAL)acc;
                }
}

#undef FN
#undef CAT
#undef CAT_
