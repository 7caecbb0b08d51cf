"""Pins against the reference's REAL output: crops of its golden image
tests/rendered/18_expected.png (committed as data in tests/golden/ref_golden_patches.json, made by
tests/golden/make_ref_patches.py) must be reproduced pixel-exactly
  - by the CPU oracle (this is what pins the oracle's stroke / cap / blend / u8 / fill rules), and
  - by the HIP path through the C ABI (GPU vs the reference's own pixels, no oracle in between)."""
import json
import os

import numpy as np
import pytest

from osm_renderer_amd import abi
from osm_renderer_amd.display_list import TileBuilder

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_golden_patches.json")))
CAP = {"none": abi.CAP_NONE, "butt": abi.CAP_BUTT, "round": abi.CAP_ROUND, "square": abi.CAP_SQUARE}


def _display_list(patch):
    tb = TileBuilder(zoom=18, scale=1, canvas=tuple(patch["canvas"]))  # zoom is irrelevant for integer points
    for op in patch["ops"]:
        if op["kind"] == "stroke":
            tb.stroke(op["points"], op["width"], tuple(op["color"]), op["opacity"], dashes=op.get("dashes"),
                      cap=CAP[op["cap"]], use_caps_for_dashes=op.get("use_caps_for_dashes", False))
        else:
            tb.fill(op["rings"] if "rings" in op else op["ring"], tuple(op["color"]), op["opacity"])
    return tb.build()


def _check_masked(name, rgb, n_mask, min_cov, min_colours):
    p = FIX[name]
    x0, x1, y0, y1 = p["window_x0_x1_y0_y1"]
    mask = np.array([[c == "1" for c in row] for row in p["mask_rows"]])
    want = np.array(p["expected_rgb"], dtype=np.uint8)
    got = rgb[y0 : y1 + 1, x0 : x1 + 1]
    diff = (got != want).any(-1) & mask
    assert mask.sum() == n_mask and diff.sum() == 0, f"{name}: {int(diff.sum())} of {int(mask.sum())} pixels differ from the reference golden"
    # the patch is not trivial: many covered pixels, many distinct (anti-aliased / blended) colours
    cov = (want != np.array(p["canvas"], dtype=np.uint8)).any(-1) & mask
    assert cov.sum() > min_cov and len(np.unique(want[mask].reshape(-1, 3), axis=0)) >= min_colours


def _check_stub(rgb):
    _check_masked("stub", rgb, 852, 400, 20)


def _check_dashed(rgb):
    # three generations, the last one a half-transparent round-capped dash pattern: 161 distinct colours
    _check_masked("dashed", rgb, 1239, 700, 150)


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


def test_oracle_reproduces_reference_dashed_patch(oracle):
    """dash pattern + round caps for dashes (use_caps_for_dashes) + traveled distance + 0.5-opacity blend."""
    _check_dashed(oracle.render_job(_display_list(FIX["dashed"]), 0)[..., :3])
    q = json.loads(json.dumps(FIX["dashed"]))
    for op in q["ops"]:
        op["points"] = op["points"][::-1]  # the dash phase starts at the other end: must NOT match
    with pytest.raises(AssertionError):
        _check_dashed(oracle.render_job(_display_list(q), 0)[..., :3])


def _check_building(rgb):
    # translucent fill (0.9) + 0.2-px outline of a closed 8-vertex ring: every pixel of the window
    _check_masked("building", rgb, 990, 500, 60)


def test_oracle_reproduces_reference_building_patch(oracle):
    """fill-opacity < 1 blend, thin (mul < 1) stroke on a closed multi-edge ring, fill/stroke pass order."""
    _check_building(oracle.render_job(_display_list(FIX["building"]), 0)[..., :3])
    q = json.loads(json.dumps(FIX["building"]))
    q["ops"][1]["points"] = q["ops"][1]["points"][::-1]  # the stroke walk is direction sensitive
    with pytest.raises(AssertionError):
        _check_building(oracle.render_job(_display_list(q), 0)[..., :3])


def _check_courtyard(rgb):
    _check_masked("courtyard", rgb, 3193, 2000, 2)


def test_oracle_reproduces_reference_multipolygon_patch(oracle):
    """multi-ring fill (building with a courtyard): one edge table over all rings, even-odd pairing in x_min order,
    fat extents on the hole side, poisoned apex row (fill.rs:16-60 with point_pairs.rs multipolygon rings)."""
    _check_courtyard(oracle.render_job(_display_list(FIX["courtyard"]), 0)[..., :3])
    hole = FIX["courtyard"]["ops"][0]["rings"][1]
    for i in range(len(hole) - 1):
        for d in ((1, 0), (0, 1), (-1, 0), (0, -1)):
            q = json.loads(json.dumps(FIX["courtyard"]))
            r = q["ops"][0]["rings"][1]
            r[i] = [r[i][0] + d[0], r[i][1] + d[1]]
            if i == 0:
                r[-1] = list(r[0])
            with pytest.raises(AssertionError):
                _check_courtyard(oracle.render_job(_display_list(q), 0)[..., :3])
    # the same hole drawn as a second fill op (its own edge table) is NOT what the reference does
    q = json.loads(json.dumps(FIX["courtyard"]))
    q["ops"] = [dict(q["ops"][0], rings=[q["ops"][0]["rings"][0]]), dict(q["ops"][0], rings=[hole])]
    with pytest.raises(AssertionError):
        _check_courtyard(oracle.render_job(_display_list(q), 0)[..., :3])


def _check_subway(rgb):
    _check_masked("subway", rgb, 2416, 300, 60)


def test_oracle_reproduces_reference_capless_dashes_patch(oracle):
    """dashes 5,3 with NO linecap on a width-2 line (railway=subway tunnel): feathered dash ends without caps,
    traveled phase carried across a vertex; the phase is pinned to < 0.0005 px by the dash-end grey levels."""
    _check_subway(oracle.render_job(_display_list(FIX["subway"]), 0)[..., :3])
    for mod in ("lead", "cap", "nocapsfordashes_is_same"):
        q = json.loads(json.dumps(FIX["subway"]))
        if mod == "lead":
            q["ops"][0]["points"][0][0] += 1  # phase moves by ~0.64 px
        elif mod == "cap":
            q["ops"][0]["cap"] = "round"
        else:
            q["ops"][0]["use_caps_for_dashes"] = False  # cap None: the flag must not matter
            _check_subway(oracle.render_job(_display_list(q), 0)[..., :3])
            continue
        with pytest.raises(AssertionError):
            _check_subway(oracle.render_job(_display_list(q), 0)[..., :3])


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
    for name, check in (("stub", _check_stub), ("dashed", _check_dashed), ("building", _check_building),
                        ("courtyard", _check_courtyard), ("subway", _check_subway), ("wood", _check_wood)):
        out = gpu_ctx.render_batch_host(_display_list(FIX[name]))
        check(out[0, :, :, :3])
