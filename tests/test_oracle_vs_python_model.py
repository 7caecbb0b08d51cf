"""The oracle's area path against an independent Python restatement (tests/_py_area_model.py) on random ops that
exercise what no golden image covers: Square and Butt caps, use_caps_for_dashes = false, odd-length / tiny / huge dash
lists, zero-width lines, degenerate edges, translucent multi-ring fills — pixel-exact after u8 truncation."""
import numpy as np

from osm_renderer_amd import abi
from osm_renderer_amd.display_list import TileBuilder
from tests import _py_area_model as M

CAPS = {"none": abi.CAP_NONE, "butt": abi.CAP_BUTT, "round": abi.CAP_ROUND, "square": abi.CAP_SQUARE}


def _pairs(pts):
    return [(tuple(a), tuple(b)) for a, b in zip(pts[:-1], pts[1:])]


def test_random_ops_match_the_python_model(oracle):
    rng = np.random.default_rng(41)
    n_checked = 0
    for trial in range(30):
        canvas = tuple(int(v) for v in rng.integers(0, 256, size=3)) if trial % 3 else None
        tb = TileBuilder(canvas=canvas)
        px = M.Pixels(canvas)
        for _ in range(int(rng.integers(4, 10))):
            color = tuple(int(v) for v in rng.integers(0, 256, size=3))
            opacity = float(rng.choice([1.0, 0.5, 0.25, 1.0 / 3.0]))
            if rng.random() < 0.35:
                rings = []
                for _ in range(int(rng.integers(1, 3))):
                    c = rng.integers(-10, 90, size=2)
                    pts = [tuple(int(v) for v in c + rng.integers(-25, 26, size=2)) for _ in range(int(rng.integers(3, 7)))]
                    rings.append(pts + [pts[0]])
                tb.fill(rings, color, opacity)
                M.fill_contour([p for r in rings for p in _pairs(r)], color, opacity, px)
            else:
                p = rng.integers(-15, 95, size=2)
                pts = [tuple(int(v) for v in p)]
                for _ in range(int(rng.integers(1, 5))):
                    p = p + rng.integers(-30, 31, size=2)
                    pts.append(tuple(int(v) for v in p))
                if rng.random() < 0.15:
                    pts.insert(1, pts[0])  # a degenerate first edge: `first` is cleared without drawing a cap
                width = float(rng.choice([0.0, 0.3, 1.0, 1.5, 2.0, 3.0, 5.5, 9.0]))
                cap = str(rng.choice(["none", "butt", "round", "square"]))
                dashes = None
                if rng.random() < 0.6:
                    dashes = [float(rng.choice([0.4, 1.0, 2.0, 3.0, 5.0, 7.5, 12.0])) for _ in range(int(rng.integers(1, 6)))]
                ucd = bool(rng.integers(0, 2))
                tb.stroke(pts, width, color, opacity, dashes=dashes, cap=CAPS[cap], use_caps_for_dashes=ucd)
                M.draw_lines(_pairs(pts), width, color, opacity, dashes, None if cap == "none" else cap, ucd, px)
            px.gen += 1
        px.finish()
        got = oracle.render_job(tb.build(), 0)[..., :3]
        want = px.rgb()
        bad = np.argwhere((got != want).any(-1))
        assert len(bad) == 0, f"trial {trial}: {len(bad)} pixels differ, first at (y, x) = {bad[0].tolist()}: oracle {got[tuple(bad[0])].tolist()} model {want[tuple(bad[0])].tolist()}"
        n_checked += 1
    assert n_checked == 30
