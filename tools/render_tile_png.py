#!/usr/bin/env python
"""BASELINE.json configs[0] (plumbing): one z=15 tile -> PNG.

    python tools/render_tile_png.py out.png [--backend oracle|gpu] [--x 19807 --y 10243 --scale 1]

The reference's own config (tests/osm fixture + Rust CPU path) is not runnable here (no rustc, the
.osm is missing), so the tile is the synthetic display list of SURVEY.md 8(d) for Tile{15,19807,10243}
(the first tile of test_zoom_15, tests/test_rendering.rs:152-155) with the osmosnimki-minimal canvas
colour, rendered by the CPU oracle (default, no GPU needed) or by the HIP path, and encoded with
osmt_encode_png (the rgb_triples_to_png counterpart)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osm_renderer_amd import synth
from osm_renderer_amd.renderer import encode_png

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--backend", choices=["oracle", "gpu"], default="oracle")
ap.add_argument("--x", type=int, default=19807)
ap.add_argument("--y", type=int, default=10243)
ap.add_argument("--scale", type=int, default=1)
a = ap.parse_args()
dl = synth.make_tiles([(a.x, a.y)], zoom=15, scale=a.scale)
if a.backend == "oracle":
    from oracle import oracle_py
    rgba = oracle_py.render_job(dl, 0)
else:
    from osm_renderer_amd.renderer import Context
    rgba = Context(0).render_batch_host(dl)[0]
with open(a.out, "wb") as f:
    f.write(encode_png(rgba))
print(f"wrote {a.out}: {rgba.shape[1]}x{rgba.shape[0]} RGB PNG, backend={a.backend}")
