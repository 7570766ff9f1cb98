"""Drop-in for the reference's pybind11 CUDA extension of the same name.

The reference's layer file does `import inv_conv_with_bp` (inf/layers/inv_conv.py:21) and calls
`inverse / forward / dy / dw` (inf/utils/inv_conv_cuda/inv_conv_with_bp_general.cpp:115-120).
Putting this directory on sys.path makes that import resolve here; the four functions keep the
reference signatures (caller-allocated `output`, scratch `M` accepted and ignored, a list whose
element 0 *is* `output` is returned) and run the exact semantics of the reference's CPU solver
(inf/utils/solve_mc.py:88-114) on the MI355X through libinvflow_hip.so.

Deviations from the reference *CUDA* behaviour: full CxC channel coupling for any
C >= 1 (the CUDA kernel is depthwise and returns zeros for C < 4), `dy` is the true adjoint
A^-T g (the CUDA kernel computes A^-1 g), `dw` is the true, batch-reduced, masked gradient.
"""
import invflow_hip as _h

__all__ = ["inverse", "forward", "dy", "dw"]


def inverse(input, kernel, output):
    """z = A^-1 x written into `output` (inv_conv_with_bp_general.cpp:19-28)."""
    _h.inverse(input, kernel, "TL", 0, out=output)
    return [output]


def forward(input, kernel, output):
    """xhat = A z written into `output` (inv_conv_with_bp_general.cpp:44-53)."""
    _h.forward(input, kernel, "TL", 0, out=output)
    return [output]


def dy(input, kernel, M, output):
    """dL/dx = A^-T (dL/dz) written into `output`; `M` is unused scratch (…general.cpp:70-81)."""
    _h._chk_tensor(M, "M", M.dtype if hasattr(M, "dtype") else None)
    _h.backward(input, None, kernel, "TL", 0, need_dx=True, need_dw=False, dx_out=output)
    return [output]


def dw(input, kernel, loss, M, output):
    """dL/dW written into `output` (…general.cpp:99-112).

    As in the reference call site (inf/layers/inv_conv.py:79) `input` is the layer *input* x and
    `loss` is dL/dz; z = A^-1 x and A^-T loss are recomputed here.  Callers that already hold z
    should use invflow_hip.backward (one fused call) instead.
    """
    _h._chk_tensor(M, "M", M.dtype if hasattr(M, "dtype") else None)
    z = _h.inverse(input, kernel, "TL", 0)
    _h.backward(loss, z, kernel, "TL", 0, need_dx=False, need_dw=True, dw_out=output)
    return [output]
