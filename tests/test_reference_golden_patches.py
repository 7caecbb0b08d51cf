"""Pins against the reference's REAL output: crops of its golden image
tests/rendered/18_expected.png (committed as data in tests/golden/ref_z18_patches.json, made by
tests/golden/make_ref_patches.py) must be reproduced pixel-exactly
  - by the CPU oracle (this is what pins the oracle's stroke / cap / blend / u8 / fill rules), and
  - by the HIP path through the C ABI (GPU vs the reference's own pixels, no oracle in between)."""
import json
import os

import numpy as np
import pytest

from osm_renderer_amd import abi
from osm_renderer_amd.display_list import TileBuilder

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_z18_patches.json")))
CAP = {"none": abi.CAP_NONE, "butt": abi.CAP_BUTT, "round": abi.CAP_ROUND, "square": abi.CAP_SQUARE}


def _display_list(patch):
    tb = TileBuilder(zoom=18, scale=1, canvas=tuple(patch["canvas"]))
    for op in patch["ops"]:
        if op["kind"] == "stroke":
            tb.stroke(op["points"], op["width"], tuple(op["color"]), op["opacity"], cap=CAP[op["cap"]])
        else:
            tb.fill(op["ring"], tuple(op["color"]), op["opacity"])
    return tb.build()


def _check_stub(rgb):
    p = FIX["stub"]
    x0, x1, y0, y1 = p["window_x0_x1_y0_y1"]
    mask = np.array([[c == "1" for c in row] for row in p["mask_rows"]])
    want = np.array(p["expected_rgb"], dtype=np.uint8)
    got = rgb[y0 : y1 + 1, x0 : x1 + 1]
    diff = (got != want).any(-1) & mask
    assert mask.sum() == 852 and diff.sum() == 0, f"{int(diff.sum())} of {int(mask.sum())} stub pixels differ from the reference golden"
    # the patch is not trivial: hundreds of covered pixels, > 100 anti-aliased ones, 20+ distinct colours
    cov = (want != np.array(p["canvas"], dtype=np.uint8)).any(-1) & mask
    assert cov.sum() > 400 and len(np.unique(want[mask].reshape(-1, 3), axis=0)) >= 20


def _check_wood(rgb):
    p = FIX["wood"]
    x0, x1, y0, y1 = p["window_x0_x1_y0_y1"]
    want = np.array([[c == "1" for c in row] for row in p["expected_fill_mask_rows"]])
    got = (rgb[y0 : y1 + 1, x0 : x1 + 1] == np.array(p["fill_rgb"], dtype=np.uint8)).all(-1)
    assert want.sum() == 3411 and (got != want).sum() == 0
    outside = rgb[y0 : y1 + 1, x0 : x1 + 1][~want]
    assert (outside == np.array(p["canvas"], dtype=np.uint8)).all()


def test_oracle_reproduces_reference_stroke_patch(oracle):
    _check_stub(oracle.render_job(_display_list(FIX["stub"]), 0)[..., :3])


def test_oracle_reproduces_reference_fill_patch(oracle):
    _check_wood(oracle.render_job(_display_list(FIX["wood"]), 0)[..., :3])


def test_stroke_patch_is_selective(oracle):
    """Moving an endpoint by one pixel or reversing the segment no longer matches the golden."""
    p = json.loads(json.dumps(FIX["stub"]))
    for d in ((1, 0), (0, 1), (-1, 0), (0, -1)):
        q = json.loads(json.dumps(p))
        for op in q["ops"]:
            op["points"][1] = [op["points"][1][0] + d[0], op["points"][1][1] + d[1]]
        with pytest.raises(AssertionError):
            _check_stub(oracle.render_job(_display_list(q), 0)[..., :3])
    q = json.loads(json.dumps(p))
    for op in q["ops"]:
        op["points"] = op["points"][::-1]
    with pytest.raises(AssertionError):
        _check_stub(oracle.render_job(_display_list(q), 0)[..., :3])


@pytest.mark.gpu
def test_gpu_reproduces_reference_patches(gpu_ctx):
    for name, check in (("stub", _check_stub), ("wood", _check_wood)):
        out = gpu_ctx.render_batch_host(_display_list(FIX[name]))
        check(out[0, :, :, :3])
