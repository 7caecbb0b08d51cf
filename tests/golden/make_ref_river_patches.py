#!/usr/bin/env python
"""Builds tests/golden/ref_river_patches.json: three crops of the reference's z14 / z15 golden images
(tests/rendered/14_expected.png, 15_expected.png: the expected output of tests/test_rendering.rs:147-155) with display-list
inputs that re-synthesise them — the first pins mined from the goldens below z17.

All three are the same real feature, waterway=river, seen at two zooms: opaque, linecap round,
  z14: color #b5d0d0 width 5   (tests/mapcss/mapnik.mapcss:542-546)
  z15: color #b5d0d0 width 6   (tests/mapcss/mapnik.mapcss:556-560)
over the canvas #f1eee8 (mapnik.mapcss:8-9), one Stroke-pass op each.  The integer vertices were recovered by search against
the CPU oracle (brute force over both end points / coordinate descent + vertex insertion for the bends, both way directions;
the searches are the ones of tests/golden/fit_search.py pointed at these windows) until ZERO pixels of the window differed:

  "river15_end"   z15, mosaic tile (col 0, row 1): the free end of a tributary, (141,123) -> (169,170): a width-6 stroke and
                  its Round end cap (a cap stub of line.rs:33-57).  43 x 49 window; the park polygons that enter it on the
                  right are masked out (pixels whose golden colour is not on the canvas -> #b5d0d0 mixing line): 1743 compared
                  pixels, 0 differ.  The reversed way differs in 2 pixels, every other end-point pair of the 7^4 searched in >= 54.
  "river14_end"   z14, mosaic tile (col 0, row 1): the same free end one zoom lower, (198,61) -> (212,85), width 5 (z14 pixel =
                  (z15 mosaic pixel + 256) / 2: the tile origins differ by one z15 tile).  33 x 24 window (the main course enters the golden two pixels further down), same kind of mask.
  "river14_bends" z14, mosaic tile (col 0, row 1): 66 rows of the river's main course with FOUR vertices inside the window —
                  (150,193) (153,181) (155,175) (158,169) (174,136) (193,111), drawn in this order (south to north).  draw_lines
                  has no joins: consecutive segments overlap and set_pixel keeps the larger alpha inside the generation
                  (line.rs:24-31, tile_pixels.rs:114-118) — this crop pins exactly that, at four bends of 3 ... 24 degrees, and
                  the direction of the walk: the whole 61 x 66 window (4026 px, 51 colours) matches; drawn north to south ONE
                  pixel differs; without the vertex (158,169) — it lies within a pixel of the line (155,175)-(174,136) — 14 do.

None of the three shortens the list of rules that NO golden can pin (Square / Butt caps: no such line is visible in any golden;
use_caps_for_dashes = false: the goldens were rendered with the Josm style type; label collisions: need the .osm) — see
oracle/osm_oracle.cpp's header.  What else was tried at z14-z16 and why it is not here: the dead-end highway=secondary stub of the
z14 golden (casing 8.5 + width 8 with the dashes 4,2 that the cascade leaves on the main layer) fits to 22 px of 347 only — its
upper end is not the way's end; the railway=subway tunnel piece of the z15 golden (dashes 5,3, no cap, a free END) fits to 47 px
with a straight lead-in: the dash phase depends on the hidden part of the way under highway=primary, which has bends; the z16
river lies over landuse polygons whose vertices would all have to be fitted too.

Run in the build container only (reads /root/reference); the JSON it writes is the fixture."""
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
CANVAS = [0xF1, 0xEE, 0xE8]
RIVER = [0xB5, 0xD0, 0xD0]

PATCHES = {
    # name: (golden, tile col, tile row, points, width, window x0 x1 y0 y1 (tile-relative, inclusive), palette mask?)
    "river15_end": ("15_expected.png", 0, 1, [(141, 123), (169, 170)], 6.0, (136, 178, 130, 178), True),
    "river14_end": ("14_expected.png", 0, 1, [(198, 61), (212, 85)], 5.0, (192, 224, 63, 86), True),
    "river14_bends": ("14_expected.png", 0, 1, [(150, 193), (153, 181), (155, 175), (158, 169), (174, 136), (193, 111)], 5.0, (140, 200, 124, 189), False),
}


def main():
    out = {"_provenance": __doc__}
    for name, (png, col, row, pts, width, (x0, x1, y0, y1), palette) in PATCHES.items():
        im = np.array(Image.open(os.path.join("/root/reference/tests/rendered", png)).convert("RGB"))
        tile = im[row * 256 : (row + 1) * 256, col * 256 : (col + 1) * 256]
        win = tile[y0 : y1 + 1, x0 : x1 + 1].astype(int)
        mask = np.ones(win.shape[:2], bool)
        if palette:  # only pixels on the canvas -> river mixing line: other features enter the window
            bg, fg = np.array(CANVAS), np.array(RIVER)
            a = (win[..., 0] - bg[0]) / (fg[0] - bg[0])
            mask = np.abs(win - (bg + a[..., None] * (fg - bg))).max(-1) < 2.0
        out[name] = {
            "source": f"tests/rendered/{png}, mosaic tile (col {col}, row {row}), tile-relative pixel coordinates",
            "window_x0_x1_y0_y1": [x0, x1, y0, y1],
            "canvas": CANVAS,
            "ops": [{"kind": "stroke", "points": [list(p) for p in pts], "width": width, "color": RIVER, "opacity": 1.0, "cap": "round"}],
            "mask_rows": ["".join("1" if v else "0" for v in r) for r in mask],
            "expected_rgb": win.tolist(),
        }
        print(name, "mask px", int(mask.sum()), "colours", len(np.unique(win[mask].reshape(-1, 3), axis=0)),
              "covered", int(((win != np.array(CANVAS)).any(-1) & mask).sum()))
    with open(os.path.join(HERE, "ref_river_patches.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
