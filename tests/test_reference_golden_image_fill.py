"""Pin of Filler::Image (fill.rs:36-40, icon.rs:60-62) against the reference's REAL output: the only `fill-image`
area in its golden images, a cemetery strip of tests/rendered/18_expected.png that runs through three mosaic tiles
(fixture tests/golden/ref_image_fill_patch.json, made by tests/golden/make_ref_image_fill_patch.py).  A pixel the
fill covers must show icon[(y mod h) * w + (x mod w)] with x, y relative to ITS tile (the pattern restarts at every
tile origin), every other pixel must not; checked for the CPU oracle and for the HIP path through the C ABI."""
import json
import os

import numpy as np
import pytest

from osm_renderer_amd.display_list import TileBuilder

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_image_fill_patch.json")))
ICON = np.array(FIX["icon_rgba"], dtype=np.uint8)
CANVAS = (1, 2, 3)  # a colour the icon does not contain


def _display_list(tile, image_id, shift=(0, 0)):
    tb = TileBuilder(zoom=18, scale=1, canvas=CANVAS)
    ring = [[x + shift[0], y + shift[1]] for x, y in tile["ring_tile_coords"]]
    tb.fill_image(ring + ring[:1], image_id, opacity=0.37)  # the opacity of an image fill is ignored (fill.rs:36-40)
    return tb.build()


def _check(tile, rgb, expect_match=True):
    x0, x1, y0, y1 = tile["window_x0_x1_y0_y1"]
    want = np.array([[c == "1" for c in row] for row in tile["expected_pattern_mask_rows"]])
    h, w, _ = ICON.shape
    yy, xx = np.mgrid[y0 : y1 + 1, x0 : x1 + 1]
    pat = ICON[yy % h, xx % w][..., :3]
    win = rgb[y0 : y1 + 1, x0 : x1 + 1]
    is_pat = (win == pat).all(-1)
    is_canvas = (win == np.array(CANVAS, dtype=np.uint8)).all(-1)
    ok = bool((is_pat == want).all() and (is_canvas == ~want).all())
    if expect_match:
        assert want.sum() > 50
        assert ok, f"tile {tile['tile_col_row']}: {int((is_pat != want).sum())} of {want.size} window pixels differ from the reference golden"
        if "raw_block_rgb" in tile:  # raw golden pixels, not derived from the icon
            bx, by = tile["raw_block_x_y"]
            np.testing.assert_array_equal(rgb[by : by + 10, bx : bx + 10], np.array(tile["raw_block_rgb"], dtype=np.uint8))
    return ok


def test_fixture_is_not_trivial():
    assert len(FIX["tiles"]) == 3 and (ICON[..., 3] == 255).all()
    assert sum(sum(r.count("1") for r in t["expected_pattern_mask_rows"]) for t in FIX["tiles"]) > 2000
    assert len(np.unique(ICON.reshape(-1, 4), axis=0)) >= 15  # a real texture, not a flat colour


def test_oracle_reproduces_the_reference_image_fill(oracle):
    for tile in FIX["tiles"]:
        rgb = oracle.render_job(_display_list(tile, 0), 0, images=[ICON])[..., :3]
        _check(tile, rgb)
    # sensitivity: a one-pixel shift of the ring (coverage) or of the phase must NOT reproduce the golden
    big = max(FIX["tiles"], key=lambda t: sum(r.count("1") for r in t["expected_pattern_mask_rows"]))
    for shift in ((1, 0), (0, 1), (-1, 0)):
        rgb = oracle.render_job(_display_list(big, 0, shift), 0, images=[ICON])[..., :3]
        assert not _check(big, rgb, expect_match=False)
    # a pattern anchored at mosaic (not tile) coordinates would be shifted in tiles whose origin is not a multiple of
    # the icon size — here every origin is (256 = 16 * 16), so that variant is indistinguishable; what the three
    # tiles do pin is the restart: each was rendered as its own tile and matches with its own (x mod w, y mod h)


@pytest.mark.gpu
def test_gpu_reproduces_the_reference_image_fill(gpu_ctx):
    iid = gpu_ctx.register_image(ICON)
    for tile in FIX["tiles"]:
        rgb = gpu_ctx.render_batch_host(_display_list(tile, iid))[0][..., :3]
        _check(tile, rgb)
