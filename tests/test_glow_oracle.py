"""CPU: the oracle's restatements of ActNorm / Squeeze / Coupling (oracle/oracle.py) against the golden vectors the
reference's own layers produced (tests/golden/make_golden_glow.py)."""
import numpy as np
import pytest

from conftest import golden_files, load_golden, rel_err

TOL = 2e-6  # the fixtures are fp32 results of fp32 torch code; the oracle computes in fp64


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    return o


@pytest.mark.parametrize("path", golden_files("actnorm_"), ids=lambda p: p.split("/")[-1][:-4])
def test_actnorm(oracle, path):
    g = load_golden(path)
    if int(g["init_from_data"]):
        mean, lstd = oracle.actnorm_stats(g["x"])
        assert rel_err(mean, g["translation"]) < TOL and rel_err(lstd, g["log_scale"]) < TOL
    y, ld = oracle.actnorm_forward(g["x"], g["translation"], g["log_scale"])
    assert rel_err(y, g["y"]) < TOL and rel_err(ld, g["logdet"]) < TOL
    assert rel_err(oracle.actnorm_reverse(g["y"], g["translation"], g["log_scale"]), g["x_rev"]) < TOL
    gx, gt, gls = oracle.actnorm_backward(g["gy"], g["gld"], g["x"], g["translation"], g["log_scale"])
    assert rel_err(gx, g["gx"]) < TOL and rel_err(gt, g["g_translation"]) < 1e-5 and rel_err(gls, g["g_log_scale"]) < 1e-5


@pytest.mark.parametrize("path", golden_files("squeeze_"), ids=lambda p: p.split("/")[-1][:-4])
def test_squeeze(oracle, path):
    g = load_golden(path)
    assert np.array_equal(oracle.space_to_depth(g["x"]), g["y"])
    assert np.array_equal(oracle.depth_to_space(g["y"]), g["x_back"]) and np.array_equal(g["x_back"], g["x"])


@pytest.mark.parametrize("path", golden_files("coupling_"), ids=lambda p: p.split("/")[-1][:-4])
def test_coupling(oracle, path):
    g = load_golden(path)
    y, ld = oracle.coupling_forward(g["x"], g["h"])
    assert rel_err(y, g["y"]) < TOL and rel_err(ld, g["logdet"]) < 1e-5
    assert rel_err(oracle.coupling_reverse(g["y"], g["h"]), g["x_rev"]) < 1e-5
    gx, gh = oracle.coupling_backward(g["gy"], g["gld"], g["x"], g["h"])
    assert rel_err(gh, g["gh"]) < 1e-5
    # the direct part of dL/dx: the second half of the channels receives nothing through the net
    ch = g["x"].shape[1] // 2
    assert rel_err(gx[:, ch:], g["gx_total"][:, ch:]) < 1e-5


@pytest.mark.parametrize("path", golden_files("slr_"), ids=lambda p: p.split("/")[-1][:-4])
def test_slr(oracle, path):
    g = load_golden(path)
    a = float(g["alpha"])
    y, ld = oracle.slr_forward(g["x"], a)
    assert rel_err(y, g["y"]) < TOL and rel_err(ld, g["logdet"]) < 1e-5
    assert rel_err(oracle.slr_backward(g["gy"], g["gld"], g["x"], a), g["gx"]) < 1e-5
    assert rel_err(oracle.slr_reverse(g["y"], a), g["x_rev"]) < 1e-5


@pytest.mark.parametrize("path", golden_files("spline_"), ids=lambda p: p.split("/")[-1][:-4])
def test_spline(oracle, path):
    g = load_golden(path)
    tb = float(g["tail_bound"])
    cw, ch, dv = oracle.spline_tables(g["p_unnormalized_widths"], g["p_unnormalized_heights"],
                                      g["p_unnormalized_derivatives"], tb)
    y, lad = oracle.rqspline(g["x"], cw, ch, dv, tb)
    assert rel_err(y, g["y"]) < 1e-5 and rel_err(lad.reshape(len(y), -1).sum(-1), g["logdet"]) < 1e-4
    xr, _ = oracle.rqspline(g["y"], cw, ch, dv, tb, inverse=True)
    assert rel_err(xr, g["x_rev"]) < 1e-5 and rel_err(xr, g["x"]) < 1e-5
    # the input gradient by central differences of the oracle's own forward (fp64)
    eps = 1e-6
    gl = g["gld"].astype(np.float64).reshape(-1, 1, 1, 1)
    yp, lp = oracle.rqspline(g["x"].astype(np.float64) + eps, cw, ch, dv, tb)
    ym, lm = oracle.rqspline(g["x"].astype(np.float64) - eps, cw, ch, dv, tb)
    gx = g["gy"] * (yp - ym) / (2 * eps) + gl * (lp - lm) / (2 * eps)
    knot = np.min(np.abs(g["x"][..., None] - cw), axis=-1) < 1e-4  # (the difference quotient straddles a knot there)
    assert rel_err(np.where(knot, 0, gx), np.where(knot, 0, g["gx"])) < 1e-4


@pytest.mark.parametrize("path", golden_files("splinepe_"), ids=lambda p: p.split("/")[-1][:-4])
def test_spline_individual_weights(oracle, path):
    """SplineActivation(individual_weights=True) of the reference (one set of knots per element) against the oracle's
    element-by-element restatement: outputs, log-det, reverse"""
    g = load_golden(path)
    tb = float(g["tail_bound"])
    uw, uh, ud = g["p_unnormalized_widths"], g["p_unnormalized_heights"], g["p_unnormalized_derivatives"]
    y, lad = oracle.rqspline_individual(g["x"], uw, uh, ud, tb)
    assert rel_err(y, g["y"]) < 1e-5 and rel_err(lad.reshape(len(y), -1).sum(-1), g["logdet"]) < 1e-4
    xr, _ = oracle.rqspline_individual(g["y"], uw, uh, ud, tb, inverse=True)
    assert rel_err(xr, g["x_rev"]) < 1e-5 and rel_err(xr, g["x"]) < 1e-5
