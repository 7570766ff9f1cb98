"""Host-side binding of the C ABI (include/invflow.h) for PyTorch-ROCm tensors.

PyTorch is plumbing here: device memory, the current stream and the caching allocator for the
scratch the library asks for.  All arithmetic happens in libinvflow_hip.so (hand-written HIP
for gfx950).  There is deliberately NO CPU / eager fallback: if the library is missing or a
tensor is not on the GPU the call raises.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libinvflow_hip.so")

ORDERS = {"TL": 0, "TR": 1, "BL": 2, "BR": 3}
FLAG_GENERAL_DIAG = 1
FLAG_EXACT_F32 = 2
FLAG_NO_MFMA = 4
FLAG_WHOLE_IMAGE = 8
OP_INVERSE, OP_FORWARD, OP_BACKWARD, OP_DY, OP_DW = range(5)

_lib = None

_vp, _i, _u, _sz, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_size_t, ctypes.c_float

# symbol -> (restype, argtypes); mirrors include/invflow.h one to one
SIGNATURES = {
    "ifl_version": (_i, []),
    "ifl_last_error": (ctypes.c_char_p, []),
    "ifl_profile_enable": (None, [_i]),
    "ifl_profile_collect": (_i, [_i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i)]),
    "ifl_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _u]),
    "ifl_inverse_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _u, _vp, _sz, _vp, _vp, _vp]),
    "ifl_workspace_bytes_bf16": (_sz, [_i, _i, _i, _i, _i, _i, _i, _u]),
    "ifl_inverse_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _u, _vp, _sz, _vp, _vp, _vp]),
    "ifl_forward_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _u, _vp, _sz, _vp]),
    "ifl_backward_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _i, _i, _u, _vp, _sz, _vp, _vp, _vp]),
    "ifl_unit_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _u]),
    "ifl_unit_inverse_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _u, _vp, _sz, _vp, _vp, _vp]),
    "ifl_unit_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _u, _vp, _sz, _vp, _vp, _vp]),
    "ifl_carry_bytes": (_sz, [_i, _i, _i]),
    "ifl_scan_state_bytes": (_sz, []),
    "ifl_scan_state_voided_offset": (_sz, []),
    "ifl_forward_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _u, _vp, _sz, _vp]),
    "ifl_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _i, _i, _u, _vp, _sz, _vp, _vp, _vp]),
    "ifl_dw_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _u, _vp, _sz, _vp]),
    "ifl_conv2d_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ifl_conv2d_wgrad_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ifl_conv2d_igrad_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ifl_conv2d_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "ifl_glow_workspace_bytes": (_sz, [_i, _i]),
    "ifl_actnorm_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ifl_actnorm_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ifl_actnorm_stats_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ifl_squeeze_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ifl_coupling_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ifl_coupling_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ifl_actnorm_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ifl_actnorm_backward_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ifl_squeeze_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ifl_adam_flat_f32": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _f, _f, _f, _f, _i, _vp]),
    "ifl_cond_supported": (_i, [_i, _i]),
    "ifl_cond_weights_floats": (_sz, [_i, _i]),
    "ifl_cond_pixels_padded": (_i, [_i, _i, _i]),
    "ifl_cond_prep_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "ifl_cond_prep_many_f32": (_i, [_vp, _i, _sz, _vp]),
    "ifl_cond_forward_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ifl_cond_backward_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "ifl_cond_grads_floats": (_sz, [_i, _i]),
    "ifl_cond_backward_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "ifl_coupling_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ifl_coupling_backward_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ifl_activation_workspace_bytes": (_sz, [_i, _i, _i]),
    "ifl_slr_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "ifl_slr_backward_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "ifl_rqspline_tables_f32": (_i, [_vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    "ifl_rqspline_tables_backward_f32": (_i, [_vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    "ifl_rqspline_f32": (_i, [_vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ifl_rqspline_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ifl_rqspline_p_f32": (_i, [_vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ifl_rqspline_p_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ifl_rqspline_pe_workspace_bytes": (_sz, [_i, _i, _i]),
    "ifl_rqspline_pe_f32": (_i, [_vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "ifl_rqspline_pe_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _i, _i, _vp, _sz, _vp]),
}


def lib():
    """Load libinvflow_hip.so (built by inverse-flow_amd/build.py).  Fails loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libinvflow_hip.so not found at %s -- build it with `python inverse-flow_amd/build.py` "
                "(there is no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


PROF_TAGS = {"scan": 0, "wgrad": 1, "conv": 2, "fold": 3, "fallback": 4}


def profile_enable(on=True):
    """Bracket every tagged kernel launch of this thread with hipEvents (bench.py roofline leg)."""
    lib().ifl_profile_enable(1 if on else 0)


def profile_collect():
    """-> {tag: (total device ms, launches)}; waits for the recorded events."""
    out = {}
    for name, tag in PROF_TAGS.items():
        ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
        _check(lib().ifl_profile_collect(tag, ctypes.byref(ms), ctypes.byref(n)), "ifl_profile_collect")
        out[name] = (ms.value, n.value)
    return out


def _check(rc, what):
    if rc != 0:
        msg = lib().ifl_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg))


def _chk_tensor(t, name, dtype=torch.float32):
    # wording follows the reference's CHECK_CUDA / CHECK_CONTIGUOUS
    # (inf/utils/inv_conv_cuda/inv_conv_with_bp_general.cpp:15-17)
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s (got %s)" % (name, dtype, t.dtype))


def _storage(t, name):
    """("f32" | "bf16", dtype): the storage format of an activation picks the entry point, as the reference's kernels are
    dispatched on the tensor's dtype (inv_conv_with_bp_kernel_general.cu:112)."""
    if isinstance(t, torch.Tensor) and t.dtype == torch.bfloat16:
        return "bf16", torch.bfloat16
    return "f32", torch.float32


def _same_device(*ts):
    dev = ts[0].device
    for t in ts[1:]:
        if t is not None and t.device != dev:
            raise RuntimeError("all tensors must be on the same device (%s vs %s)" % (dev, t.device))
    return dev


def _order(order):
    if isinstance(order, str):
        if order not in ORDERS:
            raise ValueError("unknown order: {}".format(order))
        return ORDERS[order]
    return int(order)


def _shape5(x, w):
    if x.dim() != 4 or w.dim() != 4:
        raise RuntimeError("expected x (B,C,H,W) and kernel (C,C,KH,KW)")
    B, C, H, W = x.shape
    if w.shape[0] != C or w.shape[1] != C:
        raise RuntimeError("kernel shape %s does not match %d channels" % (tuple(w.shape), C))
    return B, C, H, W, w.shape[2], w.shape[3]


_get_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _raw_stream():
    """handle of the current stream of the current device (the private accessor Triton / inductor use costs ~0.3 us;
    torch.cuda.current_stream().cuda_stream ~5 us, which matters for the launch-bound small layers)"""
    if _get_raw_stream is not None:
        return _get_raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on(dev):
    """device guard, free when `dev` is already current (the common case)"""
    idx = dev.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(dev)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


def _ptr(t):
    return None if t is None else t.data_ptr()


# Persistent block of the two-workgroup scan (include/invflow.h, ifl_scan_state_bytes): an argument of every scan call.  The
# library never allocates, so this host layer owns one zero-filled block per (device, stream): launches that share a block
# must not overlap, and launches of one stream do not.
_scan_states = {}


def scan_state(dev):
    """The block of the current stream of `dev`, or None while a stream without one is being captured (the scan then runs
    one workgroup per image on the round-1 kernel: the same result within the tolerance)."""
    stream = _raw_stream()  # (called under the device guard: the current device is `dev`)
    key = (torch.cuda.current_device(), stream)
    st = _scan_states.get(key)
    if st is not None:
        return st
    if torch.cuda.is_current_stream_capturing():
        return None
    nb = int(lib().ifl_scan_state_bytes())
    with _on(dev):
        st = torch.zeros(nb, dtype=torch.uint8, device=dev)
        torch.cuda.current_stream(dev).synchronize()  # (the zero fill is complete before any launch can see the block)
    _scan_states[key] = st
    return st


def scan_voided(dev):
    """How many wide-layer launches of this (device, stream)'s scan state were redone by the exact fp32 scan (telemetry
    word of the block, include/invflow.h); 0 when no block exists yet."""
    st = scan_state(dev)
    if st is None:
        return 0
    off = int(lib().ifl_scan_state_voided_offset())
    return int(st[off:off + 8].view(torch.int64).item())


def new_carry(w):
    """Buffer for the forward -> backward side channel of one step (see ifl_carry_bytes in invflow.h)."""
    C, _, KH, KW = w.shape
    return torch.empty(int(lib().ifl_carry_bytes(C, KH, KW)), dtype=torch.uint8, device=w.device)


def inverse(x, w, order="TL", flags=0, out=None, carry=None):
    """z = A^-1 x.  Replaces inv_conv_with_bp.inverse (inv_conv_with_bp_general.cpp:19-28).

    `carry` (from new_carry) is filled for the backward call of the same step."""
    sfx, dt = _storage(x, "input")
    _chk_tensor(x, "input", dt)
    _chk_tensor(w, "kernel")
    B, C, H, W, KH, KW = _shape5(x, w)
    if out is None:
        out = torch.empty_like(x)
    else:
        _chk_tensor(out, "output", dt)
        if out.shape != x.shape:
            raise RuntimeError("output shape mismatch")
    dev = _same_device(x, w, out)
    L = lib()
    with _on(dev):
        nb = (L.ifl_workspace_bytes_bf16 if sfx == "bf16" else L.ifl_workspace_bytes)(OP_INVERSE, B, C, H, W, KH, KW, flags)
        ws = _ws(nb, dev)
        rc = getattr(L, "ifl_inverse_" + sfx)(_ptr(x), _ptr(w), _ptr(out), B, C, H, W, KH, KW, _order(order), flags, _ptr(ws),
                                              nb, _ptr(carry), _ptr(scan_state(dev)), _raw_stream())
    _check(rc, "ifl_inverse_" + sfx)
    return out


def _ptr4(ts):
    """host array of four device pointers (NULL entries for None)"""
    arr = (ctypes.c_void_p * 4)()
    for i, t in enumerate(ts):
        arr[i] = _ptr(t)
    return arr


def unit_inverse(x, ws4, flags=0, carries=None):
    """The inverse-flow block TL -> TR -> BL -> BR (inf/layers/inv_flow.py:13-53): returns the four layer outputs
    [z_TL, z_TR, z_BL, z_BR]; one fold launch for the four layers (ifl_unit_inverse_f32)."""
    _chk_tensor(x, "input")
    if len(ws4) != 4:
        raise RuntimeError("unit_inverse needs four kernels (TL, TR, BL, BR)")
    for w in ws4:
        _chk_tensor(w, "kernel")
    B, C, H, W, KH, KW = _shape5(x, ws4[0])
    for w in ws4[1:]:
        if tuple(w.shape) != tuple(ws4[0].shape):
            raise RuntimeError("the four kernels of a block must have one shape")
    dev = _same_device(x, *ws4)
    zs = [torch.empty_like(x) for _ in range(4)]
    L = lib()
    with _on(dev):
        nb = L.ifl_unit_workspace_bytes(OP_INVERSE, B, C, H, W, KH, KW, flags)
        ws = _ws(nb, dev)
        wp, zp = _ptr4(ws4), _ptr4(zs)
        cp = _ptr4(carries) if carries is not None else None
        rc = L.ifl_unit_inverse_f32(_ptr(x), wp, zp, B, C, H, W, KH, KW, flags, _ptr(ws), nb, cp,
                                    _ptr(scan_state(dev)), _raw_stream())
    _check(rc, "ifl_unit_inverse_f32")
    return zs


def unit_backward(g, zs, ws4, flags=0, carries=None):
    """dL/dx and [dL/dw_TL, .., dL/dw_BR] of the block from g = dL/dz_BR (ifl_unit_backward_f32)."""
    _chk_tensor(g, "grad_output")
    B, C, H, W, KH, KW = _shape5(g, ws4[0])
    dev = _same_device(g, *zs, *ws4)
    dx = torch.empty_like(g)
    dws = [torch.empty_like(w) for w in ws4]
    L = lib()
    with _on(dev):
        nb = L.ifl_unit_workspace_bytes(OP_BACKWARD, B, C, H, W, KH, KW, flags)
        ws = _ws(nb, dev)
        cp = _ptr4(carries) if carries is not None else None
        rc = L.ifl_unit_backward_f32(_ptr(g), _ptr4(zs), _ptr4(ws4), _ptr(dx), _ptr4(dws), B, C, H, W, KH, KW, flags,
                                     _ptr(ws), nb, cp, _ptr(scan_state(dev)), _raw_stream())
    _check(rc, "ifl_unit_backward_f32")
    return dx, dws


def forward(z, w, order="TL", flags=0, out=None, want_logdet=False):
    """xhat = A z (and log|det A| per image).  Replaces inv_conv_with_bp.forward (…general.cpp:44-53)."""
    sfx, dt = _storage(z, "input")
    _chk_tensor(z, "input", dt)
    _chk_tensor(w, "kernel")
    B, C, H, W, KH, KW = _shape5(z, w)
    if out is None:
        out = torch.empty_like(z)
    else:
        _chk_tensor(out, "output", dt)
        if out.shape != z.shape:
            raise RuntimeError("output shape mismatch")
    dev = _same_device(z, w, out)
    ld = torch.empty(B, dtype=torch.float32, device=dev) if want_logdet else None
    L = lib()
    with _on(dev):
        nb = (L.ifl_workspace_bytes_bf16 if sfx == "bf16" else L.ifl_workspace_bytes)(OP_FORWARD, B, C, H, W, KH, KW, flags)
        ws = _ws(nb, dev)
        rc = getattr(L, "ifl_forward_" + sfx)(_ptr(z), _ptr(w), _ptr(out), _ptr(ld), B, C, H, W, KH, KW, _order(order), flags,
                                              _ptr(ws), nb, _raw_stream())
    _check(rc, "ifl_forward_" + sfx)
    return (out, ld) if want_logdet else out


def backward(g, z, w, order="TL", flags=0, x=None, recon_weight=0.0, need_dx=True, need_dw=True,
             dx_out=None, dw_out=None, carry=None):
    """Fused backward: dx = A^-T g, dw = -(sum dx (x) shifted z)*mask [+ recon term].

    Replaces inv_conv_with_bp.dy + inv_conv_with_bp.dw (…general.cpp:70-112).  Returns
    (dx or None, dw or None, recon_loss tensor or None).
    """
    sfx, dt = _storage(g, "output_grad")
    _chk_tensor(g, "output_grad", dt)
    _chk_tensor(w, "kernel")
    B, C, H, W, KH, KW = _shape5(g, w)
    if need_dw:
        _chk_tensor(z, "z", dt)
        if z.shape != g.shape:
            raise RuntimeError("z shape mismatch")
    recon = need_dw and x is not None and recon_weight != 0.0
    if recon:
        _chk_tensor(x, "x", dt)
    dev = _same_device(g, w, z if need_dw else None, x if recon else None)
    dx = None
    if need_dx:
        dx = dx_out if dx_out is not None else torch.empty_like(g)
        _chk_tensor(dx, "dx", dt)
    dw = None
    if need_dw:
        dw = dw_out if dw_out is not None else torch.empty_like(w)
        _chk_tensor(dw, "dw")
    rl = torch.empty(1, dtype=torch.float32, device=dev) if recon else None  # (the library clears it: the call's loss, not a sum)
    L = lib()
    with _on(dev):
        nb = L.ifl_workspace_bytes(OP_BACKWARD, B, C, H, W, KH, KW, flags)
        if need_dx and not recon:  # no activation-sized temporaries needed: fold (+ dW partials)
            nb = L.ifl_workspace_bytes(OP_DY, B, C, H, W, KH, KW, flags)
            if need_dw:
                nb += L.ifl_workspace_bytes(OP_DW, B, C, H, W, KH, KW, flags) + 512
        if sfx == "bf16":
            nb = L.ifl_workspace_bytes_bf16(OP_BACKWARD, B, C, H, W, KH, KW, flags)
        ws = _ws(nb, dev)
        rc = getattr(L, "ifl_backward_" + sfx)(_ptr(g), _ptr(z) if need_dw else None, _ptr(x) if recon else None, _ptr(w), _ptr(dx),
                                _ptr(dw), float(recon_weight) if recon else 0.0, _ptr(rl), B, C, H, W, KH, KW,
                                _order(order), flags, _ptr(ws), nb, _ptr(carry), _ptr(scan_state(dev)),
                                _raw_stream())
    _check(rc, "ifl_backward_" + sfx)
    return dx, dw, rl


def dw_from(z, dx, kernel_size, order="TL", flags=0, out=None):
    """dw = -(sum dx (x) shifted z) * mask from a precomputed dx (second half of …general.cpp:99-112)."""
    _chk_tensor(z, "z")
    _chk_tensor(dx, "dx")
    B, C, H, W = z.shape
    KH, KW = kernel_size
    if out is None:
        out = torch.empty(C, C, KH, KW, dtype=torch.float32, device=z.device)
    _chk_tensor(out, "output")
    dev = _same_device(z, dx, out)
    L = lib()
    with _on(dev):
        nb = L.ifl_workspace_bytes(OP_DW, B, C, H, W, KH, KW, flags)
        ws = _ws(nb, dev)
        rc = L.ifl_dw_f32(_ptr(z), _ptr(dx), _ptr(out), B, C, H, W, KH, KW, _order(order), flags, _ptr(ws), nb,
                          _raw_stream())
    _check(rc, "ifl_dw_f32")
    return out


# ---- SelfNormConv pieces (inf/layers/selfnorm.py:42-82, inf/utils/convbackward/conv2d_backward.cpp) ----

def _conv_dims(x, wshape, padding):
    B, Ci, H, W = x.shape
    Co, Ci2, KH, KW = wshape
    if Ci2 != Ci:
        raise RuntimeError("weight in_channels %d != input channels %d" % (Ci2, Ci))
    ph, pw = padding
    return B, Ci, Co, H, W, KH, KW, int(ph), int(pw)


def conv2d(x, w, bias=None, padding=(0, 0)):
    _chk_tensor(x, "input")
    _chk_tensor(w, "weight")
    if bias is not None:
        _chk_tensor(bias, "bias")
    B, Ci, Co, H, W, KH, KW, ph, pw = _conv_dims(x, w.shape, padding)
    dev = _same_device(x, w, bias)
    out = torch.empty(B, Co, H + 2 * ph - KH + 1, W + 2 * pw - KW + 1, dtype=torch.float32, device=dev)
    L = lib()
    with _on(dev):
        nb = L.ifl_conv2d_workspace_bytes(B, Ci, Co, H, W, KH, KW, ph, pw)
        ws = _ws(nb, dev)
        rc = L.ifl_conv2d_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), B, Ci, Co, H, W, KH, KW, ph, pw, _ptr(ws), nb,
                              _raw_stream())
    _check(rc, "ifl_conv2d_f32")
    return out


def conv2d_wgrad(gz, x, wshape, padding=(0, 0)):
    _chk_tensor(gz, "grad_output")
    _chk_tensor(x, "input")
    B, Ci, Co, H, W, KH, KW, ph, pw = _conv_dims(x, wshape, padding)
    dev = _same_device(gz, x)
    out = torch.empty(tuple(wshape), dtype=torch.float32, device=dev)
    L = lib()
    with _on(dev):
        nb = L.ifl_conv2d_workspace_bytes(B, Ci, Co, H, W, KH, KW, ph, pw)
        ws = _ws(nb, dev)
        rc = L.ifl_conv2d_wgrad_f32(_ptr(gz), _ptr(x), _ptr(out), B, Ci, Co, H, W, KH, KW, ph, pw, _ptr(ws), nb,
                                    _raw_stream())
    _check(rc, "ifl_conv2d_wgrad_f32")
    return out


def conv2d_igrad(gz, w, xshape, padding=(0, 0)):
    _chk_tensor(gz, "grad_output")
    _chk_tensor(w, "weight")
    B, Ci, H, W = xshape
    Co, _, KH, KW = w.shape
    ph, pw = int(padding[0]), int(padding[1])
    dev = _same_device(gz, w)
    out = torch.empty(tuple(xshape), dtype=torch.float32, device=dev)
    L = lib()
    with _on(dev):
        nb = L.ifl_conv2d_workspace_bytes(B, Ci, Co, H, W, KH, KW, ph, pw)
        ws = _ws(nb, dev)
        rc = L.ifl_conv2d_igrad_f32(_ptr(gz), _ptr(w), _ptr(out), B, Ci, Co, H, W, KH, KW, ph, pw, _ptr(ws), nb,
                                    _raw_stream())
    _check(rc, "ifl_conv2d_igrad_f32")
    return out


# ---- Glow-step neighbours of the layer (csrc/glow_step.hip) -------------------------------------
def _stream():
    return _raw_stream()


def _glow_ws(B, C, dev):
    nb = lib().ifl_glow_workspace_bytes(B, C)
    return _ws(nb, dev), nb


def _chk4(x, name, dtype=torch.float32):
    _chk_tensor(x, name, dtype)
    if x.dim() != 4:
        raise RuntimeError("%s must be (B, C, H, W)" % name)
    return x.shape


def actnorm(x, translation, log_scale, reverse=False, want_logdet=True):
    """(y, logdet) = ActNorm.forward (inf/layers/actnorm.py:18-38,59-67), or y = ActNorm.reverse (actnorm.py:40-54)."""
    sfx, dt = _storage(x, "input")
    B, C, H, W = _chk4(x, "input", dt)
    _chk_tensor(translation, "translation")
    _chk_tensor(log_scale, "log_scale")
    if translation.numel() != C or log_scale.numel() != C:
        raise RuntimeError("translation / log_scale must have %d entries" % C)
    dev = _same_device(x, translation, log_scale)
    y = torch.empty_like(x)
    ld = torch.empty(B, dtype=torch.float32, device=dev) if (want_logdet and not reverse) else None
    with _on(dev):
        rc = getattr(lib(), "ifl_actnorm_" + sfx)(_ptr(x), _ptr(translation), _ptr(log_scale), _ptr(y), _ptr(ld), B, C, H, W,
                                                  1 if reverse else 0, _stream())
    _check(rc, "ifl_actnorm_" + sfx)
    return y if reverse else (y, ld)


def actnorm_backward(gy, g_logdet, x, translation, log_scale):
    """(gx, g_translation, g_log_scale) of ActNorm.forward."""
    sfx, dt = _storage(x, "input")
    B, C, H, W = _chk4(x, "input", dt)
    _chk_tensor(gy, "grad_output", dt)
    if g_logdet is not None:
        _chk_tensor(g_logdet, "grad_logdet")
    dev = _same_device(x, gy, translation, log_scale, g_logdet)
    gx = torch.empty_like(x)
    gt = torch.empty(C, dtype=torch.float32, device=dev)
    gls = torch.empty(C, dtype=torch.float32, device=dev)
    with _on(dev):
        ws, nb = _glow_ws(B, C, dev)
        rc = getattr(lib(), "ifl_actnorm_backward_" + sfx)(_ptr(gy), _ptr(g_logdet), _ptr(x), _ptr(translation), _ptr(log_scale), _ptr(gx),
                                            _ptr(gt), _ptr(gls), B, C, H, W, _ptr(ws), nb, _stream())
    _check(rc, "ifl_actnorm_backward_" + sfx)
    return gx, gt, gls


def actnorm_stats(x):
    """(mean_c, log(std_c + 1e-8)): the data-dependent initialisation of ActNorm (actnorm.py:21-28)."""
    B, C, H, W = _chk4(x, "input")
    dev = x.device
    mean = torch.empty(C, dtype=torch.float32, device=dev)
    lstd = torch.empty(C, dtype=torch.float32, device=dev)
    with _on(dev):
        ws, nb = _glow_ws(B, C, dev)
        rc = lib().ifl_actnorm_stats_f32(_ptr(x), _ptr(mean), _ptr(lstd), B, C, H, W, _ptr(ws), nb, _stream())
    _check(rc, "ifl_actnorm_stats_f32")
    return mean, lstd


def space_to_depth(x):
    """inf/layers/squeeze.py:5-13."""
    sfx, dt = _storage(x, "input")
    B, C, H, W = _chk4(x, "input", dt)
    y = torch.empty(B, 4 * C, H // 2, W // 2, dtype=dt, device=x.device)
    with _on(x.device):
        rc = getattr(lib(), "ifl_squeeze_" + sfx)(_ptr(x), _ptr(y), B, C, H, W, 0, _stream())
    _check(rc, "ifl_squeeze_" + sfx)
    return y


def depth_to_space(x):
    """inf/layers/squeeze.py:16-25."""
    sfx, dt = _storage(x, "input")
    B, C4, H2, W2 = _chk4(x, "input", dt)
    if C4 % 4:
        raise RuntimeError("depth_to_space needs a multiple of 4 channels")
    y = torch.empty(B, C4 // 4, 2 * H2, 2 * W2, dtype=dt, device=x.device)
    with _on(x.device):
        rc = getattr(lib(), "ifl_squeeze_" + sfx)(_ptr(x), _ptr(y), B, C4 // 4, 2 * H2, 2 * W2, 1, _stream())
    _check(rc, "ifl_squeeze_" + sfx)
    return y


def coupling(x, h, reverse=False, want_logdet=True):
    """The affine part of Coupling.forward / .reverse (inf/layers/coupling.py:66-98) given h = net(x1)."""
    sfx, dt = _storage(x, "input")
    B, C, H, W = _chk4(x, "input", dt)
    _chk_tensor(h, "h", dt)
    if h.shape != x.shape:
        raise RuntimeError("h must have the shape of the input")
    dev = _same_device(x, h)
    y = torch.empty_like(x)
    ld = torch.empty(B, dtype=torch.float32, device=dev) if (want_logdet and not reverse) else None
    with _on(dev):
        ws, nb = _glow_ws(B, C, dev)
        rc = getattr(lib(), "ifl_coupling_" + sfx)(_ptr(x), _ptr(h), _ptr(y), _ptr(ld), B, C, H, W, 1 if reverse else 0, _ptr(ws), nb,
                                    _stream())
    _check(rc, "ifl_coupling_" + sfx)
    return y if reverse else (y, ld)


def coupling_backward(gy, g_logdet, x, h):
    """(gx_direct, gh) of the affine part of Coupling.forward; the caller backpropagates gh through the net."""
    sfx, dt = _storage(x, "input")
    B, C, H, W = _chk4(x, "input", dt)
    _chk_tensor(gy, "grad_output", dt)
    _chk_tensor(h, "h", dt)
    if g_logdet is not None:
        _chk_tensor(g_logdet, "grad_logdet")
    dev = _same_device(x, gy, h, g_logdet)
    gx = torch.empty_like(x)
    gh = torch.empty_like(h)
    with _on(dev):
        rc = getattr(lib(), "ifl_coupling_backward_" + sfx)(_ptr(gy), _ptr(g_logdet), _ptr(x), _ptr(h), _ptr(gx), _ptr(gh), B, C, H, W,
                                             _stream())
    _check(rc, "ifl_coupling_backward_" + sfx)
    return gx, gh


def adam_flat(p, g, m, v, lr, step, beta1, beta2, eps, weight_decay=0.0, decoupled=False):
    """one Adam / AdamW step over flat fp32 buffers, in place (lr, step: one-element device tensors)"""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (lr, "lr"), (step, "step")):
        _chk_tensor(t, n)
    if not (p.numel() == g.numel() == m.numel() == v.numel()):
        raise RuntimeError("adam_flat: buffers of different sizes")
    dev = _same_device(p, g, m, v, lr, step)
    with _on(dev):
        rc = lib().ifl_adam_flat_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(lr), _ptr(step), float(beta1), float(beta2),
                                     float(eps), float(weight_decay), 1 if decoupled else 0, _stream())
    _check(rc, "ifl_adam_flat_f32")


# ---- the coupling's conditioner (csrc/conditioner.hip) -----------------------------------------------------------------
def cond_supported(C, width):
    """shapes the fused conditioner covers (others stay on the layer's convolutions)"""
    return bool(lib().ifl_cond_supported(int(C), int(width)))


def cond_prep(w1, w2, w3, logs, logscale_factor):
    """transposed copies of the three kernels + the gain exp(logscale_factor logs): once per weight update"""
    for t, n in ((w1, "w1"), (w2, "w2"), (w3, "w3"), (logs, "logs")):
        _chk_tensor(t, n)
    width, C = w1.shape[0], w3.shape[0]
    if tuple(w1.shape) != (width, C // 2, 3, 3) or tuple(w2.shape[:2]) != (C, width) or w2.numel() != C * width or \
            tuple(w3.shape) != (C, C, 3, 3) or logs.numel() != C:
        raise RuntimeError("conditioner kernels must be (width, C/2, 3, 3), (C, width, 1, 1), (C, C, 3, 3)")
    dev = _same_device(w1, w2, w3, logs)
    wt = torch.empty(lib().ifl_cond_weights_floats(C, width), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = lib().ifl_cond_prep_f32(_ptr(w1), _ptr(w2), _ptr(w3), _ptr(logs), _ptr(wt), C, width, float(logscale_factor), _stream())
    _check(rc, "ifl_cond_prep_f32")
    return wt


class CondPrepTable:
    """The job table of ifl_cond_prep_many_f32 for a list of couplings' (w1, w2, w3, logs, logscale_factor): the weight
    images `wt[i]` (persistent tensors, one per entry) and the device array of ifl_cond_prep_job.  `run()` refills every
    image in one launch; `stale()` says whether a parameter has moved since the table was built."""
    _JOB = np.dtype([("w1", "<u8"), ("w2", "<u8"), ("w3", "<u8"), ("logs", "<u8"), ("wt", "<u8"), ("C", "<i4"), ("width", "<i4"),
                     ("logscale", "<f4"), ("reserved", "<i4")])

    def __init__(self, entries):
        assert self._JOB.itemsize == 56
        self.n = len(entries)
        self.wt, self._ptrs, self._keep = [], [], []
        jobs = np.zeros(self.n, dtype=self._JOB)
        self.max_floats, dev = 0, None
        for i, (w1, w2, w3, logs, logscale) in enumerate(entries):
            for t, n in ((w1, "w1"), (w2, "w2"), (w3, "w3"), (logs, "logs")):
                _chk_tensor(t, n)
            width, C = w1.shape[0], w3.shape[0]
            if tuple(w1.shape) != (width, C // 2, 3, 3) or w2.numel() != C * width or tuple(w3.shape) != (C, C, 3, 3) or \
                    logs.numel() != C or not cond_supported(C, width):
                raise RuntimeError("CondPrepTable: entry %d is no supported conditioner (C=%d, width=%d)" % (i, C, width))
            d = _same_device(w1, w2, w3, logs)
            if dev is not None and d != dev:
                raise RuntimeError("CondPrepTable: the couplings live on different devices")
            dev = d
            nfl = int(lib().ifl_cond_weights_floats(C, width))
            wt = torch.empty(nfl, dtype=torch.float32, device=dev)
            self.wt.append(wt)
            self._keep.append((w1, w2, w3, logs))
            self._ptrs.append((w1.data_ptr(), w2.data_ptr(), w3.data_ptr(), logs.data_ptr()))
            jobs[i] = (w1.data_ptr(), w2.data_ptr(), w3.data_ptr(), logs.data_ptr(), wt.data_ptr(), C, width, float(logscale), 0)
            self.max_floats = max(self.max_floats, nfl)
        self.device = dev
        self.jobs = torch.from_numpy(jobs.view(np.uint8).copy()).to(dev) if self.n else None

    def stale(self, entries):
        return len(entries) != self.n or any(
            (e[0].data_ptr(), e[1].data_ptr(), e[2].data_ptr(), e[3].data_ptr()) != p for e, p in zip(entries, self._ptrs))

    def run(self):
        if not self.n:
            return
        with _on(self.device):
            rc = lib().ifl_cond_prep_many_f32(_ptr(self.jobs), self.n, self.max_floats, _stream())
        _check(rc, "ifl_cond_prep_many_f32")


def cond_forward(x, wt, w1, w2, b3, C, width):
    """(a2, h): h = Coupling.net(x[:, :C/2]) (inf/layers/coupling.py:47-62), a2 = the second ReLU's output (for the backward)"""
    B, Cx, H, W = _chk4(x, "input")
    _chk_tensor(wt, "weights")
    _chk_tensor(w1, "w1")
    _chk_tensor(w2, "w2")
    if b3 is not None:
        _chk_tensor(b3, "bias")
    dev = _same_device(x, wt, b3)
    a2 = torch.empty(B, C, H, W, dtype=torch.float32, device=dev)
    h = torch.empty(B, C, H, W, dtype=torch.float32, device=dev)
    with _on(dev):
        rc = lib().ifl_cond_forward_f32(_ptr(x), Cx, _ptr(wt), _ptr(w1), _ptr(w2), _ptr(b3), _ptr(a2), _ptr(h), B, C, H, W, width,
                                        _stream())
    _check(rc, "ifl_cond_forward_f32")
    return a2, h


def cond_backward(x, dh, h, a2, wt, w1, dx, width, logscale_factor, low_precision=False):
    """dx[:, :C/2] += the input gradient (in place); returns (dW1, dW2, dW3, d logs, d b3) (views of one buffer).
    low_precision (the bf16 autocast step): the operand matrices of the three weight-gradient products are bf16."""
    B, Cx, H, W = _chk4(x, "input")
    C = h.shape[1]
    for t, n in ((dh, "grad_h"), (h, "h"), (a2, "a2"), (wt, "weights"), (w1, "w1"), (dx, "grad_input")):
        _chk_tensor(t, n)
    if dh.shape != h.shape or a2.shape != h.shape or dx.shape != x.shape:
        raise RuntimeError("conditioner backward: shapes do not match the forward's")
    dev = _same_device(x, dh, h, a2, wt, w1, dx)
    f32 = 0 if low_precision else 1
    nb = lib().ifl_cond_backward_workspace_bytes(B, C, H, W, width, f32)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)  # (fresh: lives until the kernels of this call have run)
    grads = torch.empty(lib().ifl_cond_grads_floats(C, width), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = lib().ifl_cond_backward_f32(_ptr(x), Cx, _ptr(dh), _ptr(h), _ptr(a2), _ptr(wt), _ptr(w1), f32, _ptr(ws), nb,
                                         _ptr(grads),
                                         _ptr(dx), B, C, H, W, width, float(logscale_factor), _stream())
    _check(rc, "ifl_cond_backward_f32")
    K1 = 9 * (C // 2)
    dw1, dw2, dw3, dlogs, db3 = grads.split([width * K1, C * width, 9 * C * C, C, C])
    return dw1.view(width, C // 2, 3, 3), dw2.view(C, width, 1, 1), dw3.view(C, C, 3, 3), dlogs, db3


# ---- activations of the Glow step (csrc/glow_step.hip) --------------------------------------------------------------
def _act_ws(B, C, n_bins, dev):
    nb = lib().ifl_activation_workspace_bytes(B, C, n_bins)
    return _ws(nb, dev), nb


def slr(x, alpha, reverse=False, want_logdet=True):
    """SmoothLeakyRelu.forward -> (y, logdet) / .reverse -> x (inf/layers/activations.py:37-54)."""
    B, C, H, W = _chk4(x, "input")
    dev = x.device
    y = torch.empty_like(x)
    ld = torch.empty(B, dtype=torch.float32, device=dev) if (want_logdet and not reverse) else None
    with _on(dev):
        ws, nb = _act_ws(B, C, 0, dev)
        rc = lib().ifl_slr_f32(_ptr(x), _ptr(y), _ptr(ld), B, C, H, W, float(alpha), 1 if reverse else 0, _ptr(ws), nb,
                               _stream())
    _check(rc, "ifl_slr_f32")
    return y if reverse else (y, ld)


def slr_backward(gy, g_logdet, x, alpha):
    B, C, H, W = _chk4(x, "input")
    _chk_tensor(gy, "grad_output")
    gx = torch.empty_like(x)
    with _on(x.device):
        rc = lib().ifl_slr_backward_f32(_ptr(gy), _ptr(g_logdet), _ptr(x), _ptr(gx), B, C, H, W, float(alpha), _stream())
    _check(rc, "ifl_slr_backward_f32")
    return gx


def _tables(cw, ch, dv):
    """knot tables: three contiguous fp32 device vectors of n_bins + 1 entries"""
    ts = [t.detach().to(torch.float32).contiguous() for t in (cw, ch, dv)]
    n = ts[0].numel()
    if ts[1].numel() != n or ts[2].numel() != n or n < 2 or not all(t.is_cuda for t in ts):
        raise RuntimeError("knot tables must be CUDA vectors of n_bins + 1 entries each")
    return ts, n - 1


def rqspline(x, cw, ch, dv, tail_bound, inverse=False, want_logdet=True):
    """The rational-quadratic spline with linear tails (inf/layers/splines/rational_quadratic.py:20-175) on shared
    knot tables: (y, logdet) of the forward map, or of the inverse map."""
    B, C, H, W = _chk4(x, "input")
    dev = x.device
    (a, b, c), nbins = _tables(cw, ch, dv)
    y = torch.empty_like(x)
    ld = torch.empty(B, dtype=torch.float32, device=dev) if want_logdet else None
    with _on(dev):
        ws, nb = _act_ws(B, C, 0, dev)
        rc = lib().ifl_rqspline_f32(_ptr(x), _ptr(a), _ptr(b), _ptr(c), nbins, float(tail_bound), _ptr(y), _ptr(ld), B, C, H, W,
                                    1 if inverse else 0, _ptr(ws), nb, _stream())
    _check(rc, "ifl_rqspline_f32")
    return y, ld


def rqspline_backward(gy, g_logdet, x, cw, ch, dv, tail_bound):
    """(gx, g_cw, g_ch, g_dv) of the forward direction."""
    B, C, H, W = _chk4(x, "input")
    _chk_tensor(gy, "grad_output")
    dev = x.device
    (a, b, c), nbins = _tables(cw, ch, dv)
    gx = torch.empty_like(x)
    gt = torch.empty(3 * (nbins + 1), dtype=torch.float32, device=dev)
    with _on(dev):
        ws, nb = _act_ws(B, C, nbins, dev)
        rc = lib().ifl_rqspline_backward_f32(_ptr(gy), _ptr(g_logdet), _ptr(x), _ptr(a), _ptr(b), _ptr(c), nbins, float(tail_bound), _ptr(gx),
                                             _ptr(gt), B, C, H, W, _ptr(ws), nb, _stream())
    _check(rc, "ifl_rqspline_backward_f32")
    gt = gt.view(3, nbins + 1)
    return gx, gt[0], gt[1], gt[2]


def rqspline_p(x, uw, uh, ud, tail_bound, inverse=False, want_logdet=True):
    """the shared-weight spline from its parameters, tables computed in the launch: (y, logdet, tables)"""
    B, C, H, W = _chk4(x, "input")
    for t, n in ((uw, "unnormalized_widths"), (uh, "unnormalized_heights"), (ud, "unnormalized_derivatives")):
        _chk_tensor(t, n)
    nbins = uw.numel()
    if uh.numel() != nbins or ud.numel() != nbins - 1:
        raise RuntimeError("spline parameters: n_bins widths, n_bins heights, n_bins - 1 derivatives")
    dev = _same_device(x, uw, uh, ud)
    y = torch.empty_like(x)
    ld = torch.empty(B, dtype=torch.float32, device=dev) if want_logdet else None
    tables = torch.empty(3 * (nbins + 1), dtype=torch.float32, device=dev)
    with _on(dev):
        ws, nb = _act_ws(B, C, 0, dev)
        rc = lib().ifl_rqspline_p_f32(_ptr(x), _ptr(uw), _ptr(uh), _ptr(ud), nbins, float(tail_bound), _ptr(y), _ptr(ld), _ptr(tables),
                                      B, C, H, W, 1 if inverse else 0, _ptr(ws), nb, _stream())
    _check(rc, "ifl_rqspline_p_f32")
    return y, ld, tables


def rqspline_p_backward(gy, g_logdet, x, tables, uw, uh, ud, tail_bound):
    """(gx, g_uw, g_uh, g_ud) of the forward direction"""
    B, C, H, W = _chk4(x, "input")
    _chk_tensor(gy, "grad_output")
    _chk_tensor(tables, "tables")
    nbins = uw.numel()
    dev = _same_device(x, gy, tables, uw, uh, ud)
    gx = torch.empty_like(x)
    guw, guh, gud = torch.empty_like(uw), torch.empty_like(uh), torch.empty_like(ud)
    with _on(dev):
        ws, nb = _act_ws(B, C, nbins, dev)
        rc = lib().ifl_rqspline_p_backward_f32(_ptr(gy), _ptr(g_logdet), _ptr(x), _ptr(tables), _ptr(uw), _ptr(uh), _ptr(ud), nbins,
                                               float(tail_bound), _ptr(gx), _ptr(guw), _ptr(guh), _ptr(gud), B, C, H, W, _ptr(ws), nb,
                                               _stream())
    _check(rc, "ifl_rqspline_p_backward_f32")
    return gx, guw, guh, gud


def _pe_params(x, uw, uh, ud):
    """(B, P, n_bins) of a per-element spline call: x (B, ...), parameters (1, ..., n_bins) / (1, ..., n_bins - 1)"""
    for t, name in ((x, "input"), (uw, "unnormalized_widths"), (uh, "unnormalized_heights"), (ud, "unnormalized_derivatives")):
        _chk_tensor(t, name)
    B, P, nbins = x.shape[0], x[0].numel() if x.shape[0] else int(uw.numel() // uw.shape[-1]), uw.shape[-1]
    if uw.numel() != P * nbins or uh.numel() != P * nbins or ud.numel() != P * (nbins - 1):
        raise RuntimeError("per-element spline: parameter shapes do not match the input's element count")
    return B, P, nbins


def rqspline_pe(x, uw, uh, ud, tail_bound, inverse=False, want_logdet=True):
    """The spline with one set of knots per element (SplineActivation(individual_weights=True), activations.py:135-144):
    (y, logdet) of the forward or of the inverse map; the knot tables are built inside the kernel."""
    B, P, nbins = _pe_params(x, uw, uh, ud)
    dev = _same_device(x, uw, uh, ud)
    y = torch.empty_like(x)
    ld = torch.empty(B, dtype=torch.float32, device=dev) if want_logdet else None
    with _on(dev):
        nb = int(lib().ifl_rqspline_pe_workspace_bytes(B, P, nbins))
        ws = _ws(nb, dev)
        rc = lib().ifl_rqspline_pe_f32(_ptr(x), _ptr(uw), _ptr(uh), _ptr(ud), nbins, float(tail_bound), _ptr(y), _ptr(ld), B, P,
                                       1 if inverse else 0, _ptr(ws), nb, _stream())
    _check(rc, "ifl_rqspline_pe_f32")
    return y, ld


def rqspline_pe_backward(gy, g_logdet, x, uw, uh, ud, tail_bound):
    """(gx, g_uw, g_uh, g_ud) of the forward direction, the parameter gradients summed over the batch."""
    B, P, nbins = _pe_params(x, uw, uh, ud)
    _chk_tensor(gy, "grad_output")
    dev = _same_device(x, gy, uw, uh, ud)
    gx = torch.empty_like(x)
    gp = torch.empty(P * (3 * nbins - 1), dtype=torch.float32, device=dev)
    with _on(dev):
        nb = int(lib().ifl_rqspline_pe_workspace_bytes(B, P, nbins))
        ws = _ws(nb, dev)
        rc = lib().ifl_rqspline_pe_backward_f32(_ptr(gy), _ptr(g_logdet), _ptr(x), _ptr(uw), _ptr(uh), _ptr(ud), nbins,
                                                float(tail_bound), _ptr(gx), _ptr(gp), B, P, _ptr(ws), nb, _stream())
    _check(rc, "ifl_rqspline_pe_backward_f32")
    return (gx, gp[:P * nbins].view(uw.shape), gp[P * nbins:2 * P * nbins].view(uh.shape), gp[2 * P * nbins:].view(ud.shape))


def rqspline_tables(uw, uh, ud, tail_bound):
    """(cw, ch, dv) knot tables from the three parameter vectors (one launch; rational_quadratic.py:35-46,97-116)."""
    nb = uw.numel()
    dev = uw.device
    cw, ch, dv = (torch.empty(nb + 1, dtype=torch.float32, device=dev) for _ in range(3))
    with _on(dev):
        rc = lib().ifl_rqspline_tables_f32(_ptr(uw), _ptr(uh), _ptr(ud), nb, float(tail_bound), _ptr(cw), _ptr(ch), _ptr(dv),
                                           _stream())
    _check(rc, "ifl_rqspline_tables_f32")
    return cw, ch, dv


def rqspline_tables_backward(g_tables, uw, uh, ud, tail_bound):
    """gradients of the parameter vectors from the (3, n_bins + 1) gradients of the tables"""
    nb = uw.numel()
    dev = uw.device
    guw, guh = torch.empty_like(uw), torch.empty_like(uh)
    gud = torch.empty_like(ud)
    with _on(dev):
        rc = lib().ifl_rqspline_tables_backward_f32(_ptr(g_tables), _ptr(uw), _ptr(uh), _ptr(ud), nb, float(tail_bound),
                                                    _ptr(guw), _ptr(guh), _ptr(gud), _stream())
    _check(rc, "ifl_rqspline_tables_backward_f32")
    return guw, guh, gud
