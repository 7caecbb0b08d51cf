"""Label pass against the reference's REAL output: the metro-station label "Арбатская" of
tests/rendered/17_expected.png (fixture tests/golden/ref_label_patches.json, made by
tests/golden/make_ref_label_patches.py from the reference's font, icon and stylesheet with NOTHING fitted but
the node's integer position) must be reproduced pixel-exactly
  - by the CPU oracle's label pass (pins font/rasterizer.rs, set_label_pixel, the label blend, draw_icon), and
  - by the HIP label kernels through the C ABI."""
import json
import os

import numpy as np
import pytest

from osm_renderer_amd import labels
from osm_renderer_amd.display_list import TileBuilder

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_label_patches.json")))


def _inputs(p, seg_shift=(0.0, 0.0), center_shift=(0, 0)):
    dl = TileBuilder(zoom=17, scale=1, canvas=tuple(p["canvas"])).build()
    segs = np.array(p["segs"], dtype=np.float64).reshape(-1, 4)
    segs = segs + np.array([seg_shift[0], seg_shift[1], seg_shift[0], seg_shift[1]])
    tl = labels.TileLabels()
    c = p["icon_center"]
    tl.label(icon=(0, c[0] + center_shift[0], c[1] + center_shift[1]), text=(tuple(p["text_color"]), segs))
    return dl, tl.build(), np.array(p["icon_rgba"], dtype=np.uint8)


def _check(p, rgb, n_mask=1135, min_cov=250, min_colours=80):
    x0, x1, y0, y1 = p["window_x0_x1_y0_y1"]
    mask = np.array([[c == "1" for c in row] for row in p["mask_rows"]])
    want = np.array(p["expected_rgb"], dtype=np.uint8)
    got = rgb[y0 : y1 + 1, x0 : x1 + 1]
    diff = (got != want).any(-1) & mask
    assert mask.sum() == n_mask and diff.sum() == 0, f"{int(diff.sum())} of {int(mask.sum())} pixels differ from the reference golden"
    covered = (want != np.array(p["canvas"], dtype=np.uint8)).any(-1) & mask
    # 9 anti-aliased glyphs + the icon: many covered pixels, many distinct blended colours
    assert covered.sum() > min_cov and len(np.unique(want[mask].reshape(-1, 3), axis=0)) > min_colours


def _check_z14(p, rgb):
    _check(p, rgb, n_mask=188, min_cov=60, min_colours=25)


def test_oracle_reproduces_reference_station_label(oracle):
    p = FIX["station"]
    dl, ll, icon = _inputs(p)
    out, st = oracle.render_job(dl, 0, images=[icon], labels=ll, want_status=True)
    assert st.tolist() == [1]
    _check(p, out[..., :3])
    # selective: a tenth of a pixel of text offset, or the icon one pixel off, no longer matches
    for kw in ({"seg_shift": (0.1, 0.0)}, {"seg_shift": (0.0, -0.1)}, {"center_shift": (1, 0)}):
        dl, ll, icon = _inputs(p, **kw)
        with pytest.raises(AssertionError):
            _check(p, oracle.render_job(dl, 0, images=[icon], labels=ll)[..., :3])


def test_oracle_reproduces_a_label_hanging_in_from_the_tile_above(oracle):
    """z14, font-size 9: the node lies 7 px ABOVE this tile (labels live in the 3x3-tile area, tile_pixels.rs:67-72);
    only the glyphs' lowest rows reach into it — and match the golden of the tile below the station's own tile."""
    p = FIX["station_z14_from_the_tile_above"]
    dl, ll, icon = _inputs(p)
    out, st = oracle.render_job(dl, 0, images=[icon], labels=ll, want_status=True)
    assert st.tolist() == [1]
    _check_z14(p, out[..., :3])
    for kw in ({"seg_shift": (0.0, 0.2)}, {"seg_shift": (0.3, 0.0)}):
        dl, ll, icon = _inputs(p, **kw)
        with pytest.raises(AssertionError):
            _check_z14(p, oracle.render_job(dl, 0, images=[icon], labels=ll)[..., :3])


@pytest.mark.gpu
def test_gpu_reproduces_reference_station_label(gpu_ctx):
    p = FIX["station"]
    dl, ll, icon = _inputs(p)
    ll.labels["image_id"] = gpu_ctx.register_image(icon)
    scene = gpu_ctx.upload(dl, ll)
    out = gpu_ctx.render(scene).cpu().numpy()
    assert scene.label_status().tolist() == [1]
    _check(p, out[0, :, :, :3])
    scene.free()
    p = FIX["station_z14_from_the_tile_above"]
    dl, ll, icon = _inputs(p)
    ll.labels["image_id"] = gpu_ctx.register_image(icon)
    scene = gpu_ctx.upload(dl, ll)
    _check_z14(p, gpu_ctx.render(scene).cpu().numpy()[0, :, :, :3])
    scene.free()
