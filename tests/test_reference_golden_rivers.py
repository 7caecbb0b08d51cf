"""Pins from the goldens BELOW z17: three crops of tests/rendered/14_expected.png and 15_expected.png — one
real waterway=river at two zooms (widths 5 and 6, Round caps), data in tests/golden/ref_river_patches.json, made by
tests/golden/make_ref_river_patches.py — reproduced pixel-exactly by the CPU oracle and, through the C ABI, by the HIP path
(GPU vs the reference's own pixels, no oracle in between).  "river14_bends" is the first pin of draw_lines' join rule: no
joins, consecutive segments overlap and the larger alpha wins inside the generation (line.rs:24-31, tile_pixels.rs:114-118),
four bends inside the window."""
import json
import os

import numpy as np
import pytest

from osm_renderer_amd import abi
from osm_renderer_amd.display_list import TileBuilder

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_river_patches.json")))
NAMES = ("river15_end", "river14_end", "river14_bends")
EXPECT = {"river15_end": (1743, 300, 50), "river14_end": (582, 150, 15), "river14_bends": (4026, 400, 50)}  # mask px, covered >, colours >=


def _display_list(patch):
    tb = TileBuilder(zoom=15, scale=1, canvas=tuple(patch["canvas"]))  # zoom is irrelevant for integer points
    for op in patch["ops"]:
        tb.stroke(op["points"], op["width"], tuple(op["color"]), op["opacity"], cap={"round": abi.CAP_ROUND, "none": abi.CAP_NONE}[op["cap"]])
    return tb.build()


def _differing(name, rgb, patch=None):
    p = patch or FIX[name]
    x0, x1, y0, y1 = p["window_x0_x1_y0_y1"]
    mask = np.array([[c == "1" for c in row] for row in p["mask_rows"]])
    want = np.array(p["expected_rgb"], dtype=np.uint8)
    n_mask, min_cov, min_colours = EXPECT[name]
    assert mask.sum() == n_mask
    cov = (want != np.array(p["canvas"], dtype=np.uint8)).any(-1) & mask
    assert cov.sum() > min_cov and len(np.unique(want[mask].reshape(-1, 3), axis=0)) >= min_colours  # not a trivial patch
    return int(((rgb[y0 : y1 + 1, x0 : x1 + 1] != want).any(-1) & mask).sum())


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_reference_river_patch(oracle, name):
    assert _differing(name, oracle.render_job(_display_list(FIX[name]), 0)[..., :3]) == 0


@pytest.mark.parametrize("name", NAMES)
def test_river_patches_are_selective(oracle, name):
    """Every +-1 move of a vertex that matters inside the window, another width and another cap no longer match the golden."""
    base = FIX[name]
    n_pts = len(base["ops"][0]["points"])
    # the two stand-in end vertices of the bends crop lie at / beyond the window's edge: their neighbours are what the window sees
    movable = range(n_pts) if name != "river14_bends" else range(1, n_pts - 1)
    for i in movable:
        for d in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            q = json.loads(json.dumps(base))
            q["ops"][0]["points"][i][0] += d[0]
            q["ops"][0]["points"][i][1] += d[1]
            assert _differing(name, oracle.render_job(_display_list(q), 0)[..., :3], q) >= 10, (name, i, d)
    for mod in ("width", "cap"):
        q = json.loads(json.dumps(base))
        if mod == "width":
            q["ops"][0]["width"] += 0.5
        else:
            q["ops"][0]["cap"] = "none"
        n = _differing(name, oracle.render_job(_display_list(q), 0)[..., :3], q)
        # (without its Round cap a width-5 end loses 6 pixels; no cap is visible in the bends window)
        assert n >= (10 if mod == "width" else 0 if name == "river14_bends" else 5), (name, mod, n)
    # the walk is direction sensitive (line.rs:65-158): the reversed way is a different pixel set
    q = json.loads(json.dumps(base))
    q["ops"][0]["points"] = q["ops"][0]["points"][::-1]
    rev = _differing(name, oracle.render_job(_display_list(q), 0)[..., :3], q)
    assert rev >= (0 if name == "river14_end" else 1), name  # (the 27-px z14 end happens to be the same set both ways)
    if name == "river14_bends":  # the vertex that lies within a pixel of its neighbours' chord is needed too
        q = json.loads(json.dumps(base))
        del q["ops"][0]["points"][3]
        assert _differing(name, oracle.render_job(_display_list(q), 0)[..., :3], q) >= 10


@pytest.mark.gpu
def test_gpu_reproduces_reference_river_patches(gpu_ctx):
    for name in NAMES:
        out = gpu_ctx.render_batch_host(_display_list(FIX[name]))
        assert _differing(name, out[0, :, :, :3]) == 0, name
