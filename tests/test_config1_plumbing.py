"""BASELINE.json configs[0] (the plumbing case): ONE z=15 tile -> PNG.  The reference's own fixture (tests/osm +
osmosnimki-minimal.mapcss on the Rust CPU path) cannot run here (no rustc, the .osm is missing); SURVEY.md 8(d)
substitutes Tile{15,19807,10243} — the first tile of test_zoom_15, tests/test_rendering.rs:152-155 — with the synthetic
display list, rendered by the oracle and written through the C ABI's rgb_triples_to_png counterpart (osmt_encode_png,
png_writer.rs:4-21).  The reference's tests compare DECODED pixels (tests/test_rendering.rs:15-23,46-51): so do these."""
import io
import os
import subprocess
import sys

import numpy as np
import pytest

from osm_renderer_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILE = (19807, 10243)


def _decode(png_bytes):
    from PIL import Image

    im = Image.open(io.BytesIO(png_bytes))
    assert im.mode == "RGB" and im.size == (256, 256)
    return np.asarray(im)


def test_config1_tool_oracle_to_png(tmp_path, oracle):
    """tools/render_tile_png.py --backend oracle (no GPU): file is an RGB8 256x256 PNG whose pixels are the oracle's."""
    out = tmp_path / "tile.png"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "render_tile_png.py"), str(out), "--backend", "oracle"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    dl = synth.make_tiles([TILE], zoom=15, scale=1)
    want = oracle.render_job(dl, 0)
    got = _decode(out.read_bytes())
    np.testing.assert_array_equal(got, want[..., :3])
    assert tuple(want[0, 0, :3]) != (0, 0, 0) or want[..., :3].any()  # not an empty canvas: the tile has 90 ops
    assert len(np.unique(got.reshape(-1, 3), axis=0)) > 50


@pytest.mark.gpu
def test_config1_tile_through_the_gpu_png_entry(gpu_ctx, oracle):
    """The same tile through osmt_render_batch_png (Drawer::draw_tile, drawer.rs:40-58): batch of ONE, PNG written by
    the GPU, decoded pixels = oracle; and through osmt_render_batch_rgb (the triples of TileRenderedPixels)."""
    dl = synth.make_tiles([TILE], zoom=15, scale=1)
    want = oracle.render_job(dl, 0)
    files = gpu_ctx.render_batch_png(dl)
    assert len(files) == 1
    np.testing.assert_array_equal(_decode(files[0]), want[..., :3])
    rgb = gpu_ctx.render_batch_rgb(dl)
    np.testing.assert_array_equal(rgb.reshape(256, 256, 3), want[..., :3])
