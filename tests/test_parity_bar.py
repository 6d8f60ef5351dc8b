"""The gradient bar itself (tests/parity_bar.py) on the CPU: it must fail for what it exists to catch, whatever the gradient's magnitude,
and its float64 tie-break must only ever excuse what the fp32 oracle itself cannot hold."""
import numpy as np
import pytest

from conftest import make_inputs
from parity_bar import grad_close, rel_errors

LEAVES = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")


@pytest.mark.parametrize("scale", [1e-9, 1e-4, 1.0, 1e5])
def test_the_bar_is_scale_free(scale):
    rng = np.random.default_rng(0)
    ref = (rng.normal(size=(4, 50, 3)) * scale).astype(np.float32)
    assert grad_close(ref.copy(), ref) == "ok"
    assert grad_close(ref * np.float32(1 + 2e-5), ref) == "ok"
    for bad in (np.zeros_like(ref), ref * np.float32(1.001), -ref, np.where(np.arange(4)[:, None, None] == 2, 0, ref).astype(np.float32)):
        with pytest.raises(AssertionError):
            grad_close(bad, ref, what="mutant at scale %g" % scale)
    nan = ref.copy(); nan[1, 2, 0] = np.nan
    with pytest.raises(AssertionError):
        grad_close(nan, ref)
    e, l2 = rel_errors(ref * np.float32(1.001), ref)
    assert 5e-4 < e < 2e-3 and 5e-4 < l2 < 2e-3


def test_an_all_zero_reference_needs_an_all_zero_result():
    z = np.zeros((3, 9), np.float32)
    assert grad_close(z.copy(), z) == "ok"
    with pytest.raises(AssertionError):
        grad_close(z + np.float32(1e-30), z)


def test_the_float64_tie_break_excuses_conditioning_and_nothing_else(oracle):
    """The fp32 oracle against its own float64 form on BASELINE config 1 under an O(1) upstream gradient: a 'got' that is the fp32 oracle
    itself perturbed within its distance from float64 passes as 'cond' when the bar is set tighter than that distance; a zero gradient, or
    one that is twice as far again, does not."""
    inp, gt, proj = make_inputs("sphere", 2, 32, 32, seed=5)
    rng = np.random.default_rng(1)
    w = rng.normal(size=(2, 32, 32, 4)).astype(np.float32)
    g32 = oracle.render_backward(inp, 32, 32, True, proj, w, None)
    g64 = oracle.render_backward(inp, 32, 32, True, proj, w.astype(np.float64), None, dtype=np.float64)
    for k in LEAVES:
        e_o32 = rel_errors(g32[k], g64[k])[0]
        assert float(np.abs(g32[k]).max()) > 0
        assert e_o32 < 5e-3, (k, e_o32)                               # the two instantiations are the same algorithm
        assert grad_close(g32[k], g32[k], ref64=g64[k]) == "ok"
        if e_o32 == 0.0:
            continue
        tight = e_o32 / 4                                             # a bar the fp32 oracle itself misses against float64
        got = g64[k].astype(np.float32)                               # the best an fp32 result can be: the float64 gradient, rounded
        if rel_errors(got, g32[k])[0] > tight:
            assert grad_close(got, g32[k], rtol=tight, ref64=g64[k]) == "cond"
        with pytest.raises(AssertionError):
            grad_close(np.zeros_like(g32[k]), g32[k], rtol=tight, ref64=lambda: g64[k], what=k)
