#!/usr/bin/env python
"""Builds tests/golden/ref_z18_patches.json: two small crops of the reference's OWN golden
image tests/rendered/18_expected.png (the expected output of its integration test
tests/test_rendering.rs:166-170, z18 mosaic, 256-px tiles with a red grid) together with
display-list inputs that re-synthesise them.

Why: the reference (Rust) cannot be built here and the .osm input of its rendering tests is
missing, so the golden PNGs are the only surviving OUTPUT of the real implementation.  The
inputs below were recovered by search (tests/golden/fit_search.py): the style parameters come
from tests/mapcss/mapnik.mapcss, the integer vertices were fitted until the CPU oracle
reproduced the crop with ZERO differing pixels.  A restatement with a different stroke walk,
feathering, cap shape, blend formula, u8 truncation or polygon boundary rule does not reach
zero on these crops (all +-1 neighbours of the fitted stroke endpoints differ in >= 40 pixels;
the reversed segment differs in 1), so they pin the oracle to the reference's real arithmetic.

  patch "stub":  dead-end highway=service way of mosaic tile (col 0, row 1):
                 ::roads-casing  color #999999 width 7 linecap round   (mapnik.mapcss:2138-2144)
                 main            color white   width 6 linecap round   (mapnik.mapcss:3546-3551)
                 over landuse=residential fill #dddddd, opaque          (mapnik.mapcss:83-85)
  patch "wood":  natural=wood / landuse=wood polygon, fill #aed1a0 opaque (mapnik.mapcss:243-246)
                 over the same #dddddd.

Run in the build container only (reads /root/reference); the JSON it writes is the fixture.
"""
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/tests/rendered/18_expected.png"
TILE_COL, TILE_ROW = 0, 1

STUB = dict(J=(180, 120), E=(252, 162), window=(200, 254, 130, 170))  # x0, x1, y0, y1 inclusive, tile coords
WOOD = dict(
    ring=[(201, 158), (190, 181), (171, 171), (155, 163), (148, 156), (141, 151), (129, 142), (124, 138), (120, 131),
          (120, 117), (126, 111), (130, 108), (134, 104), (138, 101), (153, 105), (159, 109), (201, 158)],
    window=(115, 204, 95, 189),
)


def main():
    im = np.array(Image.open(SRC).convert("RGB"))
    tile = im[TILE_ROW * 256 : (TILE_ROW + 1) * 256, TILE_COL * 256 : (TILE_COL + 1) * 256]

    x0, x1, y0, y1 = STUB["window"]
    ys, xs = np.mgrid[y0 : y1 + 1, x0 : x1 + 1]
    P = np.stack([xs, ys], -1).astype(float)
    J, E = np.array(STUB["J"], float), np.array(STUB["E"], float)
    d = E - J
    t = np.clip(((P - J) @ d) / (d @ d), 0, 1)
    dist = np.linalg.norm(P - (J + t[..., None] * d), axis=-1)
    mask = (dist <= 7.5) & (xs >= 205)  # the stub's own pixels; excludes the junction and neighbouring features
    stub = {
        "source": "tests/rendered/18_expected.png, mosaic tile (col 0, row 1), tile-relative pixel coordinates",
        "window_x0_x1_y0_y1": list(STUB["window"]),
        "canvas": [221, 221, 221],
        "ops": [
            {"kind": "stroke", "points": [list(STUB["J"]), list(STUB["E"])], "width": 7.0, "color": [0x99, 0x99, 0x99],
             "opacity": 1.0, "cap": "round"},
            {"kind": "stroke", "points": [list(STUB["J"]), list(STUB["E"])], "width": 6.0, "color": [255, 255, 255],
             "opacity": 1.0, "cap": "round"},
        ],
        "mask_rows": ["".join("1" if v else "0" for v in row) for row in mask],
        "expected_rgb": tile[y0 : y1 + 1, x0 : x1 + 1].tolist(),
    }

    x0, x1, y0, y1 = WOOD["window"]
    green = (tile[y0 : y1 + 1, x0 : x1 + 1] == np.array([174, 209, 160])).all(-1)
    wood = {
        "source": stub["source"],
        "window_x0_x1_y0_y1": list(WOOD["window"]),
        "canvas": [221, 221, 221],
        "ops": [{"kind": "fill", "ring": [list(p) for p in WOOD["ring"]], "color": [174, 209, 160], "opacity": 1.0}],
        "fill_rgb": [174, 209, 160],
        "expected_fill_mask_rows": ["".join("1" if v else "0" for v in row) for row in green],
    }
    out = {"_provenance": __doc__, "stub": stub, "wood": wood}
    with open(os.path.join(HERE, "ref_z18_patches.json"), "w") as f:
        json.dump(out, f)
    print("stub mask px", int(mask.sum()), "wood px", int(green.sum()))


if __name__ == "__main__":
    main()
