"""Generate the golden vectors in tests/golden/*.npz from the REFERENCE's exact CPU code.

Run in the build container only (needs /root/reference and oracle/_ref):

    make -C oracle && python tests/golden/make_golden.py

What is imported from the reference (PYTHONPATH=/root/reference, nothing is copied):
  * inf.utils.solve_mc.solve                      -- exact raster-order inverse (fp32 torch loops)
  * inf.utils.toeplitz.get_toeplitz_idxs / get_sparse_toeplitz -- dense operator for slogdet
  * oracle/_ref/inverse_op_cython.inverse_conv    -- compiled from inf/layers/emerging/inverse_op_cython.pyx
  * oracle/_ref/solve_parallel_mc.solve_parallel  -- compiled (serial) from inf/utils/fastflow_inverse/solve_parallel_mc.pyx
Gradients come from torch.autograd through a dense torch.linalg.solve of the operator built with
F.conv2d(F.pad(.)) -- the reference's own `compute_expensive` recipe (inf/layers/selfnorm.py:175-180).
inf.layers.inv_conv / inf.layers.selfnorm themselves cannot be imported here (ModuleNotFoundError:
wandb; CUDA extension) -- ordinary Python errors, see SURVEY 8c -- so SelfNormConv vectors restate
selfnorm.py:52-90,187-229 with torch.nn.grad.conv2d_weight/conv2d_input.

The fixtures are data only (inputs + expected outputs).
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))

from inf.utils.solve_mc import solve as ref_solve  # noqa: E402
from inf.utils import toeplitz as ref_toeplitz  # noqa: E402
import inverse_op_cython  # noqa: E402
import solve_parallel_mc  # noqa: E402

ORDER_FLIPS = {"TL": [], "TR": [3], "BL": [2], "BR": [2, 3]}
PADS = {  # inf/layers/inv_conv.py:126-144  (left, right, top, bottom)
    "TL": lambda kh, kw: (kw - 1, 0, kh - 1, 0),
    "TR": lambda kh, kw: (0, kw - 1, kh - 1, 0),
    "BL": lambda kh, kw: (kw - 1, 0, 0, kh - 1),
    "BR": lambda kh, kw: (0, kw - 1, 0, kh - 1),
}


def flip(t, order):
    ax = ORDER_FLIPS[order]
    return torch.flip(t, ax) if ax else t


def ref_init_weight(C, K, gen):
    """inf/layers/inv_conv.py:153-170 (TL): dirac + xavier_normal(gain=0.01), W[c,-1,-1,-1]=1."""
    w_eye = torch.nn.init.dirac_(torch.empty(C, C, K, K))
    fan = C * K * K
    std = 0.01 * (2.0 / (fan + fan)) ** 0.5
    w = w_eye + torch.randn(C, C, K, K, generator=gen) * std
    for c in range(C):
        w[c, -1, -1, -1] = 1.0
    return w


def finc_init_weight(C, K, gen, std=0.05):
    """inf/layers/conv.py:67-74 (PaddedConv2d.reset_parameters): N(0, 0.05), diag=1, upper=0."""
    w = torch.randn(C, C, K, K, generator=gen) * std
    for c in range(C):
        w[c, c, -1, -1] = 1.0
        w[c, c + 1:, -1, -1] = 0.0
    return w


def effective(w, diag):
    """Weight the exact solver uses: solve_mc.py:105-109 (diag tap: unit diagonal, lower part only)."""
    we = w.clone()
    C = w.shape[0]
    for c in range(C):
        if not diag:
            we[c, c, -1, -1] = 1.0
        we[c, c + 1:, -1, -1] = 0.0
    return we


def get_mask(C, K, order, diag):
    """inf/layers/inv_conv.py:233-248."""
    m = torch.ones(C, C, K, K, dtype=torch.float64)
    for c in range(C):
        if not diag:
            m[c, c, -1, -1] = 0.0
        m[c, c + 1:, -1, -1] = 0.0
    return flip(m, order)


def op_apply(z, w_stored_eff, order):
    """A z for a layer of the given order with its *stored* (order-flipped) effective weight:
    F.conv2d(F.pad(z, pad[order]), W)  -- inf/layers/conv.py:103-108."""
    kh, kw = w_stored_eff.shape[2:]
    return F.conv2d(F.pad(z, PADS[order](kh, kw)), w_stored_eff)


def case(name, B, C, H, W, K, wkind, order="TL", diag=0, seed=0, with_solve=True):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=gen)
    g = torch.randn(B, C, H, W, generator=gen)
    if wkind == "ref_init":
        w_tl = ref_init_weight(C, K, gen)
    elif wkind == "randn05":
        w_tl = finc_init_weight(C, K, gen)
    elif wkind == "gendiag":
        w_tl = finc_init_weight(C, K, gen, std=0.1)
        for c in range(C):
            w_tl[c, c, -1, -1] = (1.0 + 0.3 * torch.randn((), generator=gen)) * (-1.0 if c % 3 == 2 else 1.0)
    else:
        raise ValueError(wkind)
    # the layer stores the weight flipped for its order (inv_conv.py:172-179)
    w = flip(w_tl, order).contiguous()
    out = dict(x=x.numpy(), g=g.numpy(), w=w.numpy(), order=order, diag=diag,
               shape=np.array([B, C, H, W, K]))

    # ---- z = A^-1 x from the reference's exact solvers (flip-in / flip-out, conv.py:192-219)
    xin = flip(x, order).contiguous()
    wt = flip(w, order).contiguous()  # back to TL for the solver
    we_tl = effective(wt, diag)
    z64 = inverse_op_cython.inverse_conv(xin.double().numpy(), we_tl.double().numpy())
    out["z_cython_f64"] = flip(torch.from_numpy(z64), order).contiguous().numpy()
    if not diag:
        # solve_parallel_mc.pyx:95-98 sweeps only 2W-1 (or 2W) diagonals: it is exact only when
        # that covers all H+W-1 of them (a reference limitation for H > W; noted in DESIGN.md)
        n_steps = 2 * W if (H % 2 == 0 and W % 2 == 1) else 2 * W - 1
        if n_steps >= H + W - 1:
            zp = solve_parallel_mc.solve_parallel(xin.double().numpy(), wt.double().numpy(), (K, K))
            out["z_parallel_f64"] = flip(torch.from_numpy(np.asarray(zp)), order).contiguous().numpy()
        if with_solve:
            zs = ref_solve(xin.clone(), wt.clone(), (K, K))
            out["z_solve_f32"] = flip(zs, order).contiguous().numpy()

    # ---- forward / reconstruction: xhat = A z (fp64)
    z = torch.from_numpy(out["z_cython_f64"])
    we = flip(we_tl, order).contiguous().double()
    out["xhat_f64"] = op_apply(z, we, order).numpy()

    # ---- log|det A| : slogdet of the reference's dense Toeplitz matrix (selfnorm.py:240-246)
    n = C * H * W
    if n <= 4096:
        T_idxs, f_idxs = ref_toeplitz.get_toeplitz_idxs(we_tl.shape, (C, H, W), (1, 1), (K - 1, K - 1))
        Hp = H + K - 1
        Wp = W + K - 1
        T = torch.sparse_coo_tensor(T_idxs, ref_toeplitz.get_filter_vals(we_tl.double(), f_idxs),
                                    (C * Hp * Wp, n)).to_dense()
        T = T.view(C, Hp, Wp, n)[:, :H, :W, :].reshape(n, n)  # crop = TL padding
        # cross-check the Toeplitz operator against conv on this input (toeplitz.py:66-112 style)
        xt = (T @ xin.double().reshape(B, n).T).T.reshape(B, C, H, W)
        assert torch.allclose(xt, op_apply(xin.double(), we_tl.double(), "TL"), atol=1e-10)
        out["logdet_slogdet"] = float(torch.slogdet(T)[1])
    diagv = torch.stack([we_tl[c, c, -1, -1] for c in range(C)]).double()
    out["logdet_formula"] = float(H * W * torch.log(diagv.abs()).sum())  # emerging_module.py:26-32

    # ---- gradients: autograd through the dense solve (reference compute_expensive recipe)
    if n <= 1024:
        wp = w.double().clone().requires_grad_(True)
        xp = x.double().clone().requires_grad_(True)
        m = get_mask(C, K, order, diag)
        # effective stored weight, differentiable: masked entries replaced by constants
        const = flip(effective(torch.zeros(C, C, K, K), diag), order).double()  # 1 on unit diagonal
        we_d = wp * m + const * (1 - m) if not diag else wp * m
        eye = torch.eye(n, dtype=torch.float64).reshape(n, C, H, W)
        A = op_apply(eye, we_d, order).reshape(n, n).T  # column j = A e_j
        zz = torch.linalg.solve(A, xp.reshape(B, n).T).T.reshape(B, C, H, W)
        assert torch.allclose(zz, z, atol=1e-8), float((zz - z).abs().max())
        (zz * g.double()).sum().backward()
        out["dx_f64"] = xp.grad.numpy()
        out["dw_f64"] = wp.grad.numpy()
        out["mask"] = m.numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith("z_")})


def selfnorm_case(name, B, C, H, W, K, pad, seed, bias=True):
    """SelfNormConvFunc.backward + add_recon_grad, restated from inf/layers/selfnorm.py:52-90,187-229."""
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=gen, dtype=torch.float64)
    Wt = torch.nn.init.dirac_(torch.empty(C, C, K, K)).double() + 0.05 * torch.randn(C, C, K, K, generator=gen, dtype=torch.float64)
    R = torch.flip(Wt, (2, 3)).permute(1, 0, 2, 3).clone() + 0.02 * torch.randn(C, C, K, K, generator=gen, dtype=torch.float64)
    bw = 0.1 * torch.randn(C, generator=gen, dtype=torch.float64) if bias else None
    p = (pad, pad)
    z = F.conv2d(x, Wt, bw, 1, p)
    gz = torch.randn(z.shape, generator=gen, dtype=torch.float64)

    def flipk(k):
        return torch.flip(k, (2, 3)).permute(1, 0, 2, 3).clone()

    bwf = lambda go, inp, shape: torch.nn.grad.conv2d_weight(inp, shape, go, 1, p)  # noqa: E731
    multiple = bwf(torch.ones_like(z), torch.ones_like(x), Wt.shape) / B
    dzxt = bwf(gz, x, Wt.shape)
    wg_fwd = (dzxt - flipk(R) * multiple) / 2.0
    ig = torch.nn.grad.conv2d_input(x.shape, Wt, gz, 1, p)
    Wx = z - bw.view(1, -1, 1, 1) if bw is not None else z
    neg = bwf(-ig, Wx, R.shape)
    wg_inv = (neg + flipk(Wt) * flipk(multiple)) / 2.0
    bg = gz.flatten(2).sum(-1).sum(0) if bw is not None else None

    # add_recon_grad (selfnorm.py:187-229), recon_loss_weight = 1, sym_recon_grad both ways
    out = dict(x=x.numpy(), w=Wt.numpy(), r=R.numpy(), gz=gz.numpy(), pad=pad, z=z.numpy(),
               dx=ig.numpy(), dw_fwd=wg_fwd.numpy(), dw_inv=wg_inv.numpy(), multiple=multiple.numpy())
    if bw is not None:
        out.update(bias=bw.numpy(), dbias=bg.numpy())
    for sym in (False, True):
        Wp = Wt.clone().requires_grad_(True)
        Rp = R.clone().requires_grad_(True)
        zz = F.conv2d(x, Wp, None, 1, p)
        xh = F.conv2d(zz, Rp, None, 1, p)
        rl = (x - xh).pow(2).flatten(1).sum(-1)
        if sym:
            zsym = zz.detach()
            xsym = F.conv2d(zz, Rp, None, 1, p)
            zh = F.conv2d(xsym, Wp, None, 1, p)
            rl = (rl + (zsym - zh).pow(2).flatten(1).sum(-1)) / 2.0
        loss = rl.mean()
        loss.backward()
        tag = "sym" if sym else "asym"
        out["recon_loss_" + tag] = float(loss)
        out["recon_dw_" + tag] = Wp.grad.numpy()
        out["recon_dr_" + tag] = Rp.grad.numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


def main():
    torch.set_num_threads(4)
    # SURVEY 8c shape list
    case("inv_b2c1_28x28_k3_refinit", 2, 1, 28, 28, 3, "ref_init", seed=1)          # config 1 (if_cnn_mnist)
    case("inv_b1c4_5x5_k3_refinit", 1, 4, 5, 5, 3, "ref_init", seed=2)              # tests/inf/test_layers.py:183
    case("inv_b1c4_5x5_k3_randn05", 1, 4, 5, 5, 3, "randn05", seed=3)
    case("inv_b1c4_5x5_k2_randn05", 1, 4, 5, 5, 2, "randn05", seed=4)               # test_layers.py:152 (2,2)
    case("inv_b3c5_6x4_k3_randn05", 3, 5, 6, 4, 3, "randn05", seed=5)               # H != W, odd C
    case("inv_b2c8_8x8_k3_refinit", 2, 8, 8, 8, 3, "ref_init", seed=6)
    case("inv_b2c8_8x8_k3_randn05", 2, 8, 8, 8, 3, "randn05", seed=7)
    case("inv_b2c16_7x7_k3_randn05", 2, 16, 7, 7, 3, "randn05", seed=8)
    case("inv_b1c64_8x8_k3_refinit", 1, 64, 8, 8, 3, "ref_init", seed=9)
    case("inv_b1c64_8x8_k3_randn05", 1, 64, 8, 8, 3, "randn05", seed=10, with_solve=False)
    for order in ("TR", "BL", "BR"):
        case("inv_b2c4_6x5_k3_randn05_" + order, 2, 4, 6, 5, 3, "randn05", order=order, seed=11)
        case("inv_b1c4_5x5_k3_refinit_" + order, 1, 4, 5, 5, 3, "ref_init", order=order, seed=12)
    # general (non-unit) diagonal: emerging semantics, inverse_op_cython.pyx:64, emerging_module.py:26-32
    case("inv_b2c6_6x6_k3_gendiag", 2, 6, 6, 6, 3, "gendiag", diag=1, seed=13)
    case("inv_b2c4_5x7_k2_gendiag_BR", 2, 4, 5, 7, 2, "gendiag", order="BR", diag=1, seed=14)
    selfnorm_case("selfnorm_b3c4_8x8_k3_p1", 3, 4, 8, 8, 3, 1, seed=20)
    selfnorm_case("selfnorm_b2c8_6x6_k3_p1_nobias", 2, 8, 6, 6, 3, 1, seed=21, bias=False)
    selfnorm_case("selfnorm_b2c6_5x5_k1_p0", 2, 6, 5, 5, 1, 0, seed=22)


if __name__ == "__main__":
    main()
