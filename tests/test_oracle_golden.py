"""The oracle is only as good as its pinning: check every restatement under oracle/ against the golden
vectors produced by running the reference itself (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden
from oracle import cport, unproject_np
from oracle.reference_loop_torch import unprojection_cpu_loop

MODES = ("softmax", "sum", "mean", "max")
CASES = golden_cases("unproj")


def test_all_fixture_files_present():
    assert len(CASES) >= 8


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", MODES)
def test_c_oracle_forward_matches_reference(case, mode):
    d = load_golden("unproj", case)
    if "out_" + mode not in d:
        pytest.skip("mode not stored for this (large) case")
    out = cport.forward(d["features"], d["proj"], d["coords"], mode)
    ref = d["out_" + mode]
    if mode == "softmax":      # exp / division order differ from ATen's vectorised softmax by an ulp or two
        np.testing.assert_allclose(out, ref, rtol=0, atol=2e-6 * max(1.0, np.abs(ref).max()))
    else:                      # projection, taps and sums are restated bit-for-bit
        np.testing.assert_array_equal(out, ref)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", MODES)
def test_c_oracle_backward_matches_reference(case, mode):
    d = load_golden("unproj", case)
    if "gfeat_" + mode not in d:
        pytest.skip("gradient not stored for this (large) case")
    g = cport.backward(d["grad_out"], d["features"], d["proj"], d["coords"], mode)
    ref = d["gfeat_" + mode]
    np.testing.assert_allclose(g, ref, rtol=0, atol=1e-5 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("case", [c for c in CASES if "config0" not in c])
@pytest.mark.parametrize("mode", MODES)
def test_numpy_oracle_matches_reference(case, mode):
    """Second, independent restatement (array ops, no fma): agrees to fp32 re-ordering noise (~4e-5 at |x|~4)."""
    d = load_golden("unproj", case)
    ref = d["out_" + mode]
    tol = 5e-5 * max(1.0, np.abs(ref).max())
    np.testing.assert_allclose(unproject_np.forward(d["features"], d["proj"], d["coords"], mode), ref, rtol=0, atol=tol)
    gref = d["gfeat_" + mode]
    g = unproject_np.backward(d["grad_out"], d["features"], d["proj"], d["coords"], mode)
    np.testing.assert_allclose(g, gref, rtol=0, atol=5e-5 * max(1.0, np.abs(gref).max()))


@pytest.mark.parametrize("case", CASES)
def test_torch_loop_port_matches_reference(case):
    """The timed CPU baseline (per-(b,v) F.grid_sample loop) reproduces the reference output."""
    d = load_golden("unproj", case)
    for mode in MODES:
        if "out_" + mode not in d:
            continue
        out = unprojection_cpu_loop(torch.from_numpy(d["features"]), torch.from_numpy(d["proj"]),
                                    torch.from_numpy(d["coords"]), mode).numpy()
        np.testing.assert_allclose(out, d["out_" + mode], rtol=0, atol=1e-6 * max(1.0, np.abs(d["out_" + mode]).max()))


def test_oracle_rejects_unknown_method():
    d = load_golden("unproj", "tiny_b2v2c4")
    with pytest.raises(ValueError):
        unproject_np.forward(d["features"], d["proj"], d["coords"], "median")
    with pytest.raises(ValueError):
        unprojection_cpu_loop(torch.from_numpy(d["features"]), torch.from_numpy(d["proj"]), torch.from_numpy(d["coords"]), "median")


def test_adversarial_fixture_really_is_adversarial():
    """Guards the fixture itself: it must contain behind-camera voxels, an exact z == 0 and out-of-frame taps."""
    d = load_golden("unproj", "adversarial_v4c6")
    pts = d["coords"][0].reshape(-1, 3)
    hom = np.concatenate([pts, np.ones((len(pts), 1), np.float32)], 1)
    z = np.stack([(hom @ d["proj"][0, v].T)[:, 2] for v in range(4)])
    assert (z <= 0).mean() > 0.1 and (z == 0).any()
    _, w, ok = unproject_np.tap_table(d["proj"][0, 0], pts, 16, 16)
    assert (~ok).any() and ok.any()
