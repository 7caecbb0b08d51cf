#!/usr/bin/env python
"""Builds tests/golden/ref_image_fill_patch.json: the one `fill-image` area that occurs in the reference's golden
images — a `landuse=cemetery` strip of tests/rendered/18_expected.png filled with symbols/grave_yard_generic.png
(mapnik.mapcss: `area|z14-[landuse=cemetery]...[religion!=christian][religion!=jewish] { fill-image: ... }`).

How it was found: all 19 `fill-image` icons of tests/mapcss/mapnik.mapcss were matched, PHASE-ALIGNED PER TILE
(pixel (x, y) of a 256-px mosaic tile against icon[(y mod h) * w + (x mod w)], the rule of fill.rs:36-40 /
icon.rs:60-62), against all five goldens; only this icon matches anywhere (63 % of a 32x32 window; every other icon
< 2 %).  The strip runs diagonally through FOUR mosaic tiles — (7,2), (8,2), (7,3), (8,3) — and the pattern restarts
at every tile origin, which is what pins "tile-relative coordinates".

The ring was fitted like the other patches (tests/golden/fit_search.py): four integer vertices in mosaic coordinates,
exhaustive joint search of both end points of each edge (+-5), objective = pixels whose "is pattern" state differs from
the golden; the two long edges are reproduced with ZERO differing pixels over their whole length (rows 756..887); the
20 remaining differences sit at the two short ends (y <= 754, y >= 889), where the real polygon has more vertices than
the stand-in quadrilateral, and are outside the compared windows.  The fixture stores, per tile, the window, the
expected "is pattern" mask of the golden and a block of raw golden pixels from inside the strip.

Run in the build container only (reads /root/reference); the JSON it writes is the fixture.
"""
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tests"
RING = [(2024, 747), (2194, 893), (2186, 901), (2016, 754)]  # mosaic coordinates of 18_expected.png, drawn in this order (closed)
ROWS = (756, 887)  # rows of the strip away from its two ends
TILES = [(7, 2), (8, 2), (7, 3), (8, 3)]


def main():
    icon = np.array(Image.open(f"{REF}/mapcss/symbols/grave_yard_generic.png").convert("RGBA"))
    im = np.array(Image.open(f"{REF}/rendered/18_expected.png").convert("RGB"))
    h, w, _ = icon.shape
    tiles = []
    total = 0
    for col, row in TILES:
        ox, oy = col * 256, row * 256
        t = im[oy : oy + 256, ox : ox + 256]
        yy, xx = np.mgrid[0:256, 0:256]
        pat = icon[yy % h, xx % w][..., :3]
        gold = (t == pat).all(-1)
        # window: the strip's rows inside this tile, minus the mosaic's red grid (row 0 and column 255 of every tile,
        # tests/test_rendering.rs:108-114)
        y0, y1 = max(ROWS[0] - oy, 1), min(ROWS[1] - oy, 255)
        if y0 > y1:
            continue
        # columns: the strip lies between its two long edges; bound them from the ring itself (+- 6 px)
        (ax, ay), (bx_, by_), (cx, cy), (dx, dy) = RING
        xs_at = lambda y: sorted([ax + (bx_ - ax) * (y - ay) / (by_ - ay), dx + (cx - dx) * (y - dy) / (cy - dy)])
        lo = min(xs_at(oy + y0)[0], xs_at(oy + y1)[0]) - ox
        hi = max(xs_at(oy + y0)[1], xs_at(oy + y1)[1]) - ox
        x0, x1 = max(int(lo) - 6, 0), min(int(hi) + 6, 254)
        if x0 > x1 or not gold[y0 : y1 + 1, x0 : x1 + 1].any():
            continue
        win = gold[y0 : y1 + 1, x0 : x1 + 1]
        # a block of raw golden pixels well inside the strip (direct reference-pixel comparison)
        best = None
        for by in range(y0, y1 - 10):
            for bx in range(x0, x1 - 10):
                if gold[by : by + 10, bx : bx + 10].all():
                    best = (bx, by)
                    break
            if best:
                break
        entry = {
            "tile_col_row": [col, row],
            "window_x0_x1_y0_y1": [x0, x1, y0, y1],
            "ring_tile_coords": [[x - ox, y - oy] for x, y in RING],
            "expected_pattern_mask_rows": ["".join("1" if v else "0" for v in r) for r in win],
        }
        if best:
            bx, by = best
            entry["raw_block_x_y"] = [bx, by]
            entry["raw_block_rgb"] = t[by : by + 10, bx : bx + 10].tolist()
        tiles.append(entry)
        total += int(win.sum())
    out = {
        "_provenance": __doc__,
        "source": "tests/rendered/18_expected.png + tests/mapcss/symbols/grave_yard_generic.png",
        "icon_rgba": icon.tolist(),
        "tiles": tiles,
    }
    with open(os.path.join(HERE, "ref_image_fill_patch.json"), "w") as f:
        json.dump(out, f)
    print("tiles", len(tiles), "pattern pixels compared", total, [t["window_x0_x1_y0_y1"] for t in tiles])


if __name__ == "__main__":
    main()
