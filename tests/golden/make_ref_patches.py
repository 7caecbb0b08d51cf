#!/usr/bin/env python
"""Builds tests/golden/ref_golden_patches.json: two small crops of the reference's OWN golden
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
  patch "dashed": a highway=service way with access=private of mosaic tile (col 0, row 0): the same casing +
                 white strokes, then the ::access overlay color #efa9a9 width 6 opacity 0.5 dashes 6,8
                 linecap round (mapnik.mapcss:4171-4180) with use_caps_for_dashes = true (Josm style type,
                 src/mapcss/styler.rs:95), over landuse=retail fill #efc8c8 (mapnik.mapcss:232).  The way runs
                 from the junction (230,181) to the free end (130,119); the reversed direction differs in 111 px.
  patch "building": z17 golden (tests/rendered/17_expected.png), mosaic tile (col 1, row 3): a building
                 polygon, fill #bca9a9 fill-opacity 0.9 (mapnik.mapcss:2349-2353) in the Fill pass, then its
                 outline color #330066 width 0.2 (mapnik.mapcss:2355-2359; no linecap) in the Stroke pass, over
                 the canvas #f1eee8.  8 fitted vertices; the whole 30x33 window matches (66 colours); the other
                 ring direction differs in 8 px, every +-1 vertex move in >= 29.
  patch "courtyard": z17 golden, mosaic tile (col 4, row 0): the courtyard of a type=multipolygon building
                 (fill #bca9a9 fill-opacity 0.9, Fill pass only: drawer.rs:82-105 draws multipolygons in no
                 other pass, hence no outline) over the canvas #f1eee8.  ONE fill op with TWO rings: the fitted
                 9-vertex inner ring, and a stand-in rectangle for the outer ring, whose real outline lies
                 outside the crop (its only effect inside the window is one crossing left and one right of the
                 hole on every row, which the rectangle supplies).  Pins the multi-ring rule of fill.rs:16-60:
                 all rings share one edge table, crossings pair up in x_min order across rings, so the hole's
                 left edge ends a span with its x_max and its right edge starts one with its x_min, and the
                 hole's apex row is poisoned.  57x57 window minus 56 px of an icon/label drawn later; 3193
                 compared pixels, 0 differ; every +-1 move of a hole vertex differs.
  patch "subway": z17 golden, mosaic tile (col 0, row 2): a railway=subway tunnel, color #999999 width 2
                 dashes 5,3 and NO linecap (mapnik.mapcss:3377-3381), over the canvas.  The visible piece is one
                 straight segment on the line through (-256,243),(256,221) (slope and offset regressed from the
                 anti-aliased values, then the integer family with slope -11/256 picked); the way's earlier
                 part lies in other tiles, so ONE lead-in segment (-887,-504)->(-256,243) outside the tile
                 stands in for it: its only visible effect is the dash phase at the tile edge (traveled mod 8 =
                 1.83946 at (-256,243); 0.0005 off already differs).  Pins dashes WITHOUT caps (cap None with
                 use_caps_for_dashes), the feathered dash ends, traveled accumulation across a join and the
                 thin-line (half-width 1) across feather: whole 151x16 window, 0 of 2416 px differ.
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
DASHED = dict(J=(130, 119), E=(230, 181), window=(118, 226, 108, 190))  # drawn E -> J
BUILDING = dict(  # drawn in this vertex order (closed)
    ring=[(197, 77), (195, 84), (202, 86), (198, 102), (208, 104), (210, 95), (226, 99), (230, 85), (197, 77)],
    window=(194, 223, 74, 106),
)
WOOD = dict(
    ring=[(201, 158), (190, 181), (171, 171), (155, 163), (148, 156), (141, 151), (129, 142), (124, 138), (120, 131),
          (120, 117), (126, 111), (130, 108), (134, 104), (138, 101), (153, 105), (159, 109), (201, 158)],
    window=(115, 204, 95, 189),
)


SUBWAY = dict(points=[(-887, -504), (-256, 243), (256, 221)], window=(0, 150, 222, 237))
COURTYARD = dict(
    outer_stand_in=[(-50, 100), (400, 100), (400, 400), (-50, 400), (-50, 100)],
    hole=[(79, 193), (95, 208), (97, 210), (72, 235), (68, 240), (68, 242), (61, 236), (65, 232), (53, 220), (79, 193)],
    window=(50, 106, 190, 246),
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

    tile00 = im[0:256, 0:256]
    x0, x1, y0, y1 = DASHED["window"]
    ys, xs = np.mgrid[y0 : y1 + 1, x0 : x1 + 1]
    P = np.stack([xs, ys], -1).astype(float)
    J, E = np.array(DASHED["J"], float), np.array(DASHED["E"], float)
    d = E - J
    t = np.clip(((P - J) @ d) / (d @ d), 0, 1)
    dist = np.linalg.norm(P - (J + t[..., None] * d), axis=-1)
    # whole width near the free end; further on (building outlines touch the casing) only the opaque core
    dmask = ((dist <= 7.5) & (xs <= 185)) | ((dist <= 2.2) & (xs <= 218))
    pts = [list(DASHED["E"]), list(DASHED["J"])]
    dashed = {
        "source": "tests/rendered/18_expected.png, mosaic tile (col 0, row 0), tile-relative pixel coordinates",
        "window_x0_x1_y0_y1": list(DASHED["window"]),
        "canvas": [0xEF, 0xC8, 0xC8],
        "ops": [
            {"kind": "stroke", "points": pts, "width": 7.0, "color": [0x99, 0x99, 0x99], "opacity": 1.0, "cap": "round",
             "use_caps_for_dashes": True},
            {"kind": "stroke", "points": pts, "width": 6.0, "color": [255, 255, 255], "opacity": 1.0, "cap": "round",
             "use_caps_for_dashes": True},
            {"kind": "stroke", "points": pts, "width": 6.0, "color": [0xEF, 0xA9, 0xA9], "opacity": 0.5, "cap": "round",
             "dashes": [6.0, 8.0], "use_caps_for_dashes": True},
        ],
        "mask_rows": ["".join("1" if v else "0" for v in row) for row in dmask],
        "expected_rgb": tile00[y0 : y1 + 1, x0 : x1 + 1].tolist(),
    }

    im17 = np.array(Image.open("/root/reference/tests/rendered/17_expected.png").convert("RGB"))
    tile17 = im17[3 * 256 : 4 * 256, 1 * 256 : 2 * 256]
    x0, x1, y0, y1 = BUILDING["window"]
    ring = [list(p) for p in BUILDING["ring"]]
    building = {
        "source": "tests/rendered/17_expected.png, mosaic tile (col 1, row 3), tile-relative pixel coordinates",
        "window_x0_x1_y0_y1": list(BUILDING["window"]),
        "canvas": [0xF1, 0xEE, 0xE8],
        "ops": [
            {"kind": "fill", "ring": ring, "color": [0xBC, 0xA9, 0xA9], "opacity": 0.9},
            {"kind": "stroke", "points": ring, "width": 0.2, "color": [0x33, 0x00, 0x66], "opacity": 1.0, "cap": "none"},
        ],
        "mask_rows": ["1" * (x1 - x0 + 1) for _ in range(y0, y1 + 1)],
        "expected_rgb": tile17[y0 : y1 + 1, x0 : x1 + 1].tolist(),
    }

    tile17c = im17[0:256, 4 * 256 : 5 * 256]
    x0, x1, y0, y1 = COURTYARD["window"]
    win = tile17c[y0 : y1 + 1, x0 : x1 + 1]
    known = (win == np.array([193, 175, 175])).all(-1) | (win == np.array([0xF1, 0xEE, 0xE8])).all(-1)
    known[:2, -3:] = False  # the real outer ring's own corner enters the window here
    courtyard = {
        "source": "tests/rendered/17_expected.png, mosaic tile (col 4, row 0), tile-relative pixel coordinates",
        "window_x0_x1_y0_y1": list(COURTYARD["window"]),
        "canvas": [0xF1, 0xEE, 0xE8],
        "ops": [{"kind": "fill", "rings": [[list(p) for p in COURTYARD["outer_stand_in"]], [list(p) for p in COURTYARD["hole"]]],
                 "color": [0xBC, 0xA9, 0xA9], "opacity": 0.9}],
        "mask_rows": ["".join("1" if v else "0" for v in row) for row in known],
        "expected_rgb": win.tolist(),
    }

    tile17s = im17[2 * 256 : 3 * 256, 0:256]
    x0, x1, y0, y1 = SUBWAY["window"]
    subway = {
        "source": "tests/rendered/17_expected.png, mosaic tile (col 0, row 2), tile-relative pixel coordinates",
        "window_x0_x1_y0_y1": list(SUBWAY["window"]),
        "canvas": [0xF1, 0xEE, 0xE8],
        "ops": [{"kind": "stroke", "points": [list(p) for p in SUBWAY["points"]], "width": 2.0, "color": [0x99, 0x99, 0x99],
                 "opacity": 1.0, "cap": "none", "dashes": [5.0, 3.0], "use_caps_for_dashes": True}],
        "mask_rows": ["1" * (x1 - x0 + 1) for _ in range(y0, y1 + 1)],
        "expected_rgb": tile17s[y0 : y1 + 1, x0 : x1 + 1].tolist(),
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
    out = {"_provenance": __doc__, "stub": stub, "dashed": dashed, "building": building, "courtyard": courtyard, "subway": subway, "wood": wood}
    with open(os.path.join(HERE, "ref_golden_patches.json"), "w") as f:
        json.dump(out, f)
    print("stub mask px", int(mask.sum()), "dashed mask px", int(dmask.sum()), "wood px", int(green.sum()),
          "courtyard px", int(known.sum()))


if __name__ == "__main__":
    main()
