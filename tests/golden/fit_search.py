#!/usr/bin/env python
"""The searches that recovered the inputs of tests/golden/ref_golden_patches.json (kept for
reproducibility; build container only — reads /root/reference and uses the CPU oracle).

  stub : brute force over both endpoints (13x13 x 9x11 candidates, both directions) of the
         two-stroke service-road stub; objective = differing pixels in the crop.  Best:
         J=(180,120) -> E=(252,162): 0 differing pixels inside the stub mask; every other
         candidate >= 1 (reversed direction) / >= 40 (moved endpoints).
  wood : convex hull of the exact-colour pixels -> Douglas-Peucker -> coordinate descent on
         vertices (+-2), then "insert one vertex near a differing pixel, wiggling both
         neighbours +-1" until the symmetric difference of fill masks is empty (16 vertices).

  dashed   : same brute force for the three-stroke access=private service way (7x7 x 12x12 endpoint
             candidates, both directions): E=(230,181) -> J=(130,119) gives 0 differing pixels in the
             1239-px mask, the reversed direction 111, every other candidate >= 215.
  building : start from eye-balled corners; exhaustive search of the two vertices that lie outside the
             comparison window (16x10 x 10x8 positions, both ring directions), then coordinate descent
             (+-2) on all eight: 0 differing pixels in the 30x33 window; reversed ring 8; any +-1 move >= 29.
"""
import itertools
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, "/root/repo")
from oracle import oracle_py as O  # noqa: E402
from osm_renderer_amd import abi  # noqa: E402

im = np.array(Image.open("/root/reference/tests/rendered/18_expected.png").convert("RGB"))
tile = im[256:512, 0:256]


def stub_score(J, E, fwd, window=(200, 254, 130, 170)):
    x0, x1, y0, y1 = window
    p = O.Pixels(1)
    p.reset((221, 221, 221))
    pts = [J, E] if fwd else [E, J]
    p.draw_lines(O.ring_to_pairs(pts), 7.0, (0x99, 0x99, 0x99), 1.0, cap=abi.CAP_ROUND)
    p.bump_generation()
    p.draw_lines(O.ring_to_pairs(pts), 6.0, (255, 255, 255), 1.0, cap=abi.CAP_ROUND)
    p.bump_generation()
    p.blend_unfinished_pixels()
    r = p.to_rgb()[y0 : y1 + 1, x0 : x1 + 1]
    return int((r != tile[y0 : y1 + 1, x0 : x1 + 1]).any(-1).sum())


def search_stub():
    best = None
    for jx, jy, ex, ey, f in itertools.product(range(172, 185), range(113, 126), range(245, 254), range(155, 166), (True, False)):
        s = stub_score((jx, jy), (ex, ey), f)
        if best is None or s < best[0]:
            best = (s, (jx, jy), (ex, ey), f)
            print(best)
    return best


_P = None


def wood_score(V, window=(115, 204, 95, 189)):
    global _P
    x0, x1, y0, y1 = window
    if _P is None:
        _P = O.Pixels(1)
        _P.reset((221, 221, 221))
    gold = np.zeros((256, 256), bool)
    gold[y0 : y1 + 1, x0 : x1 + 1] = (tile[y0 : y1 + 1, x0 : x1 + 1] == np.array([174, 209, 160])).all(-1)
    ring = [tuple(v) for v in V] + [tuple(V[0])]
    _P.fill_contour(O.ring_to_pairs(ring), (174, 209, 160), 1.0)
    a = _P.pending_alpha(0) > 0
    _P.blend_unfinished_pixels()
    return int((a != gold).sum())


if __name__ == "__main__":
    print("stub (fitted):", stub_score((180, 120), (252, 162), True), "differing pixels in the whole crop "
          "(all of them belong to neighbouring features outside the stub mask)")
    V = [(201, 158), (190, 181), (171, 171), (155, 163), (148, 156), (141, 151), (129, 142), (124, 138), (120, 131),
         (120, 117), (126, 111), (130, 108), (134, 104), (138, 101), (153, 105), (159, 109)]
    print("wood (fitted):", wood_score(V), "differing pixels")
    if "--search" in sys.argv:
        search_stub()
