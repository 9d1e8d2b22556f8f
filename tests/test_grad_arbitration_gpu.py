"""Which side is right where the HIP backward and the fp32 CPU oracle disagree?  `tools/grad_arbiter.py` walks the
tiles of the worst surfels a third time in fp64 (torch autograd over the oracle's own sorted lists,
`oracle/autograd_ref.py`) -- at the benchmark size that is 5.5 minutes of CPU and lives in
`profiles/r03_grad_arbitration.json`; this test re-runs the same arbitration on every GPU test run at a size the fp64
walk finishes in seconds (128 x 128, P = 27 648, the 1 worst surfel per gradient tensor, at most 4 tiles), so that the
claim "where they disagree, the HIP gradients are as close to the truth as the oracle's" is checked on every run, not just filed
(renderer_2dgs.py:139-165 is the call whose backward this is)."""
import importlib.util
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def _arbiter():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "grad_arbiter.py")
    spec = importlib.util.spec_from_file_location("grad_arbiter", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("regime", ["init", "trained"])
def test_hip_backward_is_as_close_to_fp64_as_the_fp32_oracle(hip_lib, regime):
    out = _arbiter().arbitrate(n_worst=1, regime=regime, res=128, grid=24, max_tiles=4)
    print(json.dumps(out))
    assert out["tiles"] >= 1 and out["list_entries_walked_in_fp64"] > 100, out
    # the fp32 oracle's own forward agrees with the fp64 walk on those tiles (same lists, same quirk)
    assert out["forward_max_abs_diff_fp32_oracle_vs_fp64_on_tiles"] < 5e-4, out
    # One (pixel, entry) pair at the alpha = 1/255 cut may be decided differently by v_exp_f32 / v_rcp_f32 than by libm (either
    # side of a discontinuity of the published algorithm; the fp64 walk sides with libm): it moves a gradient by up to
    # (1/255) T |dL/dC| -- about 1e-2 of the tensor's maximum at this size, and exactly what the "worst surfel" selection
    # finds first (measured at "init": ONE pixel of the two tiles differs from the fp64 forward by 1.0e-3, every other by
    # < 2e-4, and the gradients of the surfel behind it by shs 3.3e-3, opacities 4e-4 of max).  Beyond that allowance the HIP path may not be
    # farther from fp64 than 3x the fp32 oracle is; on the ill-conditioned tensors (scales, rotations: the fp32 oracle itself
    # is 1e-2 ... 2e-1 of max from fp64) it is within a few per cent of the oracle's distance either way.
    assert out["forward_pixels_beyond_2e-4_hip_vs_fp64"] <= 2 and out["forward_max_abs_diff_hip_vs_fp64_on_tiles"] < 5e-3, out
    for k, row in out["per_tensor"].items():
        hip, orc = row["hip"]["max_err_rel_to_max"], row["fp32_oracle"]["max_err_rel_to_max"]
        assert hip <= max(3.0 * orc, 1e-2), (regime, k, row)
        assert row["hip"]["rel_l2"] <= max(3.0 * row["fp32_oracle"]["rel_l2"], 2e-3), (regime, k, row)
