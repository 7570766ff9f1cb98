"""ctypes/numpy front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package (inverse-flow_amd/).

All functions take/return numpy arrays (float32 or float64, C-contiguous NCHW) and follow the
reference's exact CPU semantics; see invflow_oracle_impl.h for the file:line citations.
Orders TR/BL/BR are realised by explicit flips of input, weight and output exactly like
inf/layers/conv.py:192-219 (`reverse_cuda`) -- deliberately different from the in-kernel
index reflection the HIP path uses, so the two are independent statements.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ORDERS = ("TL", "TR", "BL", "BR")


def build(force=False):
    """Compile liboracle.so (gcc) and, if the reference checkout exists, oracle/_ref."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("invflow_oracle.c", "invflow_oracle_impl.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so", "-B"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/inf"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.orc_logdet_f32.restype = ctypes.c_double
        _LIB.orc_logdet_f64.restype = ctypes.c_double
    return _LIB


def _suf(a):
    if a.dtype == np.float32:
        return "_f32"
    if a.dtype == np.float64:
        return "_f64"
    raise TypeError("oracle handles float32/float64 only, got %s" % a.dtype)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype=None):
    return np.ascontiguousarray(a, dtype=dtype)


def _flip_axes(order):
    if order not in ORDERS:
        raise ValueError("unknown order: {}".format(order))
    ax = []
    if order[0] == "B":
        ax.append(2)
    if order[1] == "R":
        ax.append(3)
    return tuple(ax)


def _flip(a, order):
    ax = _flip_axes(order)
    return _c(np.flip(a, ax)) if ax else a


def _run4(name, a, w, diag, order, nthreads):
    a = _c(a)
    w = _c(w, a.dtype)
    B, C, H, W = (int(v) for v in a.shape)
    KH, KW = (int(v) for v in w.shape[2:])
    assert w.shape[:2] == (C, C)
    af, wf = _flip(a, order), _flip(w, order)
    out = np.empty_like(af)
    getattr(lib(), name + _suf(a))(_p(af), _p(wf), _p(out), B, C, H, W, KH, KW, int(diag), int(nthreads))
    return _flip(out, order)


def inverse(x, w, diag=0, order="TL", nthreads=1):
    """z = A^-1 x (solve_mc.py:88-114; diag=1: inverse_op_cython.pyx:17-66)."""
    return _run4("orc_inverse", x, w, diag, order, nthreads)


def forward(z, w, diag=0, order="TL", nthreads=1):
    """xhat = A z (conv.py:103-108 with the mask of inv_conv.py:233-248)."""
    return _run4("orc_forward", z, w, diag, order, nthreads)


def dy(g, w, diag=0, order="TL", nthreads=1):
    """dL/dx = A^-T g."""
    return _run4("orc_dy", g, w, diag, order, nthreads)


def dw(z, u, kernel_size, diag=0, order="TL", nthreads=1):
    """dL/dW = -sum u (x) shifted z, masked (inv_conv.py:223-248)."""
    z = _c(z)
    u = _c(u, z.dtype)
    B, C, H, W = (int(v) for v in z.shape)
    KH, KW = (int(v) for v in kernel_size)
    zf, uf = _flip(z, order), _flip(u, order)
    out = np.empty((C, C, KH, KW), dtype=z.dtype)
    getattr(lib(), "orc_dw" + _suf(z))(_p(zf), _p(uf), _p(out), B, C, H, W, KH, KW, int(diag), int(nthreads))
    return _flip(out, order)


def logdet(w, H, W, diag=0, order="TL"):
    """log|det A| per image (emerging_module.py:26-32); 0 for the unit diagonal."""
    w = _flip(_c(w), order)
    C = w.shape[0]
    KH, KW = w.shape[2:]
    return float(getattr(lib(), "orc_logdet" + _suf(w))(_p(w), int(C), int(H), int(W), int(KH), int(KW), int(diag)))


def mask(C, KH, KW, diag=0, order="TL", dtype=np.float32):
    """Gradient mask of inf/layers/inv_conv.py:233-248 (`get_mask`); diag=1 keeps the diagonal."""
    m = np.ones((C, C, KH, KW), dtype=dtype)
    for c in range(C):
        if not diag:
            m[c, c, -1, -1] = 0
        m[c, c + 1:, -1, -1] = 0
    return _flip(m, order)


def effective_weight(w, diag=0, order="TL"):
    """What: the weight the exact solver actually uses (solve_mc.py:105-109)."""
    w = _flip(_c(w).copy(), order)
    C = w.shape[0]
    for c in range(C):
        if not diag:
            w[c, c, -1, -1] = 1
        w[c, c + 1:, -1, -1] = 0
    return _flip(w, order)


def dense_operator(w, H, W, diag=0, order="TL"):
    """Dense matrix of A (rows/cols indexed (c,h,w) NCHW-flat) -- independent of the C loops.

    Built by pushing unit impulses through a numpy restatement of
    F.conv2d(F.pad(., order-pad), What)  (inv_conv.py:126-144 pads, conv.py:103-108).
    """
    w = np.asarray(w, dtype=np.float64)
    C, _, KH, KW = w.shape
    we = effective_weight(w, diag, order)
    n = C * H * W
    A = np.zeros((n, n))
    top = KH - 1 if order[0] == "T" else 0
    left = KW - 1 if order[1] == "L" else 0
    for c in range(C):
        for h in range(H):
            for x in range(W):
                row = (c * H + h) * W + x
                for kc in range(C):
                    for kh in range(KH):
                        ih = h + kh - top
                        if ih < 0 or ih >= H:
                            continue
                        for kw in range(KW):
                            iw = x + kw - left
                            if iw < 0 or iw >= W:
                                continue
                            A[row, (kc * H + ih) * W + iw] += we[c, kc, kh, kw]
    return A


# ---- self-normalising conv (selfnorm.py) ------------------------------------------------

def conv2d(x, w, bias=None, padding=(0, 0), nthreads=1):
    x = _c(x)
    w = _c(w, x.dtype)
    B, Ci, H, W = x.shape
    Co, _, KH, KW = w.shape
    ph, pw = padding
    OH, OW = H + 2 * ph - KH + 1, W + 2 * pw - KW + 1
    out = np.empty((B, Co, OH, OW), dtype=x.dtype)
    b = _c(bias, x.dtype) if bias is not None else None
    getattr(lib(), "orc_conv2d" + _suf(x))(_p(x), _p(w), _p(b) if b is not None else None, _p(out),
                                           B, Ci, Co, H, W, KH, KW, ph, pw, int(nthreads))
    return out


def conv2d_wgrad(gz, x, wshape, padding=(0, 0), nthreads=1):
    x = _c(x)
    gz = _c(gz, x.dtype)
    B, Ci, H, W = x.shape
    Co, _, KH, KW = wshape
    ph, pw = padding
    out = np.empty(tuple(wshape), dtype=x.dtype)
    getattr(lib(), "orc_conv2d_wgrad" + _suf(x))(_p(gz), _p(x), _p(out), B, Ci, Co, H, W, KH, KW, ph, pw,
                                                 int(nthreads))
    return out


def conv2d_igrad(gz, w, xshape, padding=(0, 0), nthreads=1):
    gz = _c(gz)
    w = _c(w, gz.dtype)
    B, Ci, H, W = xshape
    Co, _, KH, KW = w.shape
    ph, pw = padding
    out = np.empty(tuple(xshape), dtype=gz.dtype)
    getattr(lib(), "orc_conv2d_igrad" + _suf(gz))(_p(gz), _p(w), _p(out), B, Ci, Co, H, W, KH, KW, ph, pw,
                                                  int(nthreads))
    return out


def flip_kernel(w):
    """inf/layers/selfnorm.py:35-36."""
    return _c(np.flip(w, (2, 3)).transpose(1, 0, 2, 3))


def selfnorm_grads(x, W, bw, R, gz, padding):
    """SelfNormConvFunc.backward, inf/layers/selfnorm.py:52-90 (stride 1, dilation 1, groups 1).

    Returns (input_grad, weight_grad_fwd, bias_grad, weight_grad_inv).
    """
    x = _c(x)
    B = x.shape[0]
    z = conv2d(x, W, bw, padding)
    # _compute_weight_multiple, selfnorm.py:24-32
    multiple = conv2d_wgrad(np.ones_like(z), np.ones_like(x), W.shape, padding) / B
    delta_z_xt = conv2d_wgrad(gz, x, W.shape, padding)
    wg_fwd = (delta_z_xt - flip_kernel(R) * multiple) / 2.0
    input_grad = conv2d_igrad(gz, W, x.shape, padding)
    Wx = z - bw.reshape(1, -1, 1, 1) if bw is not None else z
    neg = conv2d_wgrad(-input_grad, Wx, R.shape, padding)
    wg_inv = (neg + flip_kernel(W) * flip_kernel(multiple)) / 2.0
    bg = gz.reshape(gz.shape[0], gz.shape[1], -1).sum(-1).sum(0) if bw is not None else None
    return input_grad, wg_fwd.astype(x.dtype), bg, wg_inv.astype(x.dtype)


# ---- Glow-step neighbours of the layer: numpy restatements, float64 inside -------------------
def actnorm_forward(x, translation, log_scale):
    """inf/layers/actnorm.py:18-38,59-67: ((x - t) exp(-ls), -H W sum ls)."""
    x = np.asarray(x, np.float64)
    t = np.asarray(translation, np.float64).reshape(1, -1, 1, 1)
    ls = np.asarray(log_scale, np.float64).reshape(1, -1, 1, 1)
    y = (x - t) * np.exp(-ls)
    ld = np.full(x.shape[0], -ls.sum() * x.shape[2] * x.shape[3])
    return y, ld


def actnorm_reverse(y, translation, log_scale):
    """inf/layers/actnorm.py:40-54."""
    t = np.asarray(translation, np.float64).reshape(1, -1, 1, 1)
    ls = np.asarray(log_scale, np.float64).reshape(1, -1, 1, 1)
    return np.asarray(y, np.float64) * np.exp(ls) + t


def actnorm_backward(gy, g_logdet, x, translation, log_scale):
    """Gradients of actnorm_forward w.r.t. (x, translation, log_scale) for upstream (gy, g_logdet)."""
    gy = np.asarray(gy, np.float64)
    y, _ = actnorm_forward(x, translation, log_scale)
    sc = np.exp(-np.asarray(log_scale, np.float64)).reshape(1, -1, 1, 1)
    gx = gy * sc
    gt = -(gy * sc).sum(axis=(0, 2, 3))
    gls = -(gy * y).sum(axis=(0, 2, 3))
    if g_logdet is not None:
        gls = gls - x.shape[2] * x.shape[3] * np.asarray(g_logdet, np.float64).sum()
    return gx, gt, gls


def actnorm_stats(x):
    """inf/layers/actnorm.py:21-26: mean and log(std + 1e-8) over (B, H, W), std unbiased (torch.std default)."""
    x = np.asarray(x, np.float64)
    return x.mean(axis=(0, 2, 3)), np.log(x.std(axis=(0, 2, 3), ddof=1) + 1e-8)


def space_to_depth(x):
    """inf/layers/squeeze.py:5-13."""
    B, C, H, W = x.shape
    x = x.reshape(B, C, H // 2, 2, W // 2, 2).transpose(0, 1, 3, 5, 2, 4)
    return np.ascontiguousarray(x).reshape(B, C * 4, H // 2, W // 2)


def depth_to_space(x):
    """inf/layers/squeeze.py:16-25."""
    B, C, H, W = x.shape
    x = x.reshape(B, C // 4, 2, 2, H, W).transpose(0, 1, 4, 2, 5, 3)
    return np.ascontiguousarray(x).reshape(B, C // 4, H * 2, W * 2)


def _coupling_parts(x, h):
    x = np.asarray(x, np.float64)
    h = np.asarray(h, np.float64)
    ch = x.shape[1] // 2
    return x[:, :ch], x[:, ch:], 2.0 * np.tanh(h[:, 0::2] / 2.0), h[:, 1::2]


def coupling_forward(x, h):
    """inf/layers/coupling.py:66-89 given h = net(x1): (cat(x1, x2 exp(log_s) + t), sum log_s)."""
    x1, x2, log_s, t = _coupling_parts(x, h)
    return np.concatenate([x1, x2 * np.exp(log_s) + t], axis=1), log_s.reshape(len(x1), -1).sum(-1)


def coupling_reverse(y, h):
    """inf/layers/coupling.py:92-98."""
    x1, x2, log_s, t = _coupling_parts(y, h)
    return np.concatenate([x1, (x2 - t) * np.exp(-log_s)], axis=1)


def coupling_backward(gy, g_logdet, x, h):
    """Gradients of coupling_forward w.r.t. x (direct part: h held fixed) and h."""
    gy = np.asarray(gy, np.float64)
    x1, x2, log_s, t = _coupling_parts(x, h)
    ch = x1.shape[1]
    g1, g2 = gy[:, :ch], gy[:, ch:]
    e = np.exp(log_s)
    gl = 0.0 if g_logdet is None else np.asarray(g_logdet, np.float64).reshape(-1, 1, 1, 1)
    th = log_s / 2.0
    gh = np.empty(np.asarray(h).shape, np.float64)
    gh[:, 0::2] = (g2 * x2 * e + gl) * (1.0 - th * th)
    gh[:, 1::2] = g2
    return np.concatenate([g1, g2 * e], axis=1), gh


# ---- activations of the Glow step (inf/layers/activations.py, splines/rational_quadratic.py), float64 -----------------
def slr_forward(x, alpha):
    """SmoothLeakyRelu (activations.py:37-54): (alpha x + (1 - alpha) log(1 + e^x), sum log y')."""
    x = np.asarray(x, np.float64)
    y = alpha * x + (1 - alpha) * np.logaddexp(0.0, x)
    d = alpha + (1 - alpha) / (1 + np.exp(-x))
    return y, np.log(d).reshape(len(x), -1).sum(-1)


def slr_backward(gy, g_logdet, x, alpha):
    x = np.asarray(x, np.float64)
    s = 1 / (1 + np.exp(-x))
    d1 = alpha + (1 - alpha) * s
    d2 = (1 - alpha) * s * (1 - s)
    gl = 0.0 if g_logdet is None else np.asarray(g_logdet, np.float64).reshape(-1, 1, 1, 1)
    return np.asarray(gy, np.float64) * d1 + gl * d2 / d1


def slr_reverse(y, alpha, n_iter=100):
    """newton_raphson_inverse (activations.py:27-34): x0 = y, slope clamped at 1e-2."""
    y = np.asarray(y, np.float64)
    x = y.copy()
    for _ in range(n_iter):
        fp = np.maximum(alpha + (1 - alpha) / (1 + np.exp(-x)), 1e-2)
        x = x - (alpha * x + (1 - alpha) * np.logaddexp(0.0, x) - y) / fp
    return x


def spline_tables(uw, uh, ud, tail_bound, min_size=1e-6):
    """Knot tables (cumwidths, cumheights, derivatives) of the shared-weight spline with linear tails
    (rational_quadratic.py:35-46,97-116)."""
    uw, uh, ud = (np.asarray(a, np.float64) for a in (uw, uh, ud))
    nb = len(uw)
    const = np.log(np.exp(1 - min_size) - 1)
    udp = np.pad(ud, (1, 1)) + const

    def knots(u):
        v = np.exp(u - u.max())
        v = v / v.sum()
        v = min_size + (1 - min_size * nb) * v
        cum = np.concatenate([[0.0], np.cumsum(v)]) * 2 * tail_bound - tail_bound
        cum[0], cum[-1] = -tail_bound, tail_bound
        return cum

    return knots(uw), knots(uh), min_size + np.logaddexp(0.0, udp)


def rqspline_individual(x, uw, uh, ud, tail_bound, inverse=False):
    """The spline with one set of knots per element (SplineActivation(individual_weights=True), activations.py:135-144):
    parameters of shape (1, C, H, W, n_bins[-1]); element by element with the shared-weight functions above.
    Returns (y, logabsdet) like rqspline."""
    x = np.asarray(x, np.float64)
    uw, uh, ud = (np.asarray(a, np.float64).reshape(-1, np.shape(a)[-1]) for a in (uw, uh, ud))
    xf = x.reshape(x.shape[0], -1)
    y, lad = np.empty_like(xf), np.empty_like(xf)
    for e in range(xf.shape[1]):
        cw, ch, dv = spline_tables(uw[e], uh[e], ud[e], tail_bound)
        y[:, e], lad[:, e] = rqspline(xf[:, e], cw, ch, dv, tail_bound, inverse)
    return y.reshape(x.shape), lad.reshape(x.shape)


def rqspline(x, cw, ch, dv, tail_bound, inverse=False):
    """Elementwise spline / inverse spline and its log-derivative (rational_quadratic.py:20-175); returns (y, logabsdet)."""
    x = np.asarray(x, np.float64)
    cw, ch, dv = (np.asarray(a, np.float64) for a in (cw, ch, dv))
    nb = len(cw) - 1
    inside = (x >= -tail_bound) & (x <= tail_bound)
    edges = (ch if inverse else cw).copy()
    edges[-1] += 1e-6
    k = np.clip((x[..., None] >= edges).sum(-1) - 1, 0, nb - 1)
    a, b, c, e, d0, d1 = cw[k], cw[k + 1], ch[k], ch[k + 1], dv[k], dv[k + 1]
    w, h = b - a, e - c
    delta = h / w
    with np.errstate(all="ignore"):
        if inverse:
            r = x - c
            s = d0 + d1 - 2 * delta
            qa = r * s + h * (delta - d0)
            qb = h * d0 - r * s
            qc = -delta * r
            root = (2 * qc) / (-qb - np.sqrt(qb ** 2 - 4 * qa * qc))
            out = root * w + a
            t1 = root * (1 - root)
            den = delta + s * t1
            dnum = delta ** 2 * (d1 * root ** 2 + 2 * delta * t1 + d0 * (1 - root) ** 2)
            lad = -(np.log(dnum) - 2 * np.log(den))
        else:
            th = (x - a) / w
            t1 = th * (1 - th)
            num = h * (delta * th ** 2 + d0 * t1)
            den = delta + (d0 + d1 - 2 * delta) * t1
            out = c + num / den
            dnum = delta ** 2 * (d1 * th ** 2 + 2 * delta * t1 + d0 * (1 - th) ** 2)
            lad = np.log(dnum) - 2 * np.log(den)
    return np.where(inside, out, x), np.where(inside, lad, 0.0)
