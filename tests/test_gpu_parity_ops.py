"""Differential GPU-vs-oracle tests per operation, through the C ABI: the K1..K8 vectors and
the edge cases of fill.rs / line.rs / opacity_calculator.rs / tile_pixels.rs.  Integer
coverage and the f64 canvas must be bit-exact."""
import json
import os

import numpy as np
import pytest

from osm_renderer_amd import abi, display_list
from osm_renderer_amd.display_list import TileBuilder
from tests._parity import assert_parity

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
CAPS = [abi.CAP_NONE, abi.CAP_BUTT, abi.CAP_ROUND, abi.CAP_SQUARE]


def _f64(gpu_ctx, dl):
    scene = gpu_ctx.upload(dl)
    out = gpu_ctx.render_f64(scene).cpu().numpy()
    scene.free()
    return out


def _alpha_of_single_op(gpu_ctx, dl):
    """With a black canvas (0,0,0,1) and a white op, canvas.r == the op's alpha exactly
    (a*1.0 + (1-a)*0.0)."""
    return _f64(gpu_ctx, dl)[0, :, :, 0]


def test_k1_k2_fill_coverage(gpu_ctx):
    for key in ("K1_fill_square", "K2_fill_triangle"):
        tb = TileBuilder(canvas=None)
        tb.fill(KAT[key]["ring"], (255, 255, 255), 1.0)
        a = _alpha_of_single_op(gpu_ctx, tb.build())
        ys, xs = np.nonzero(a)
        got = set(zip(xs.tolist(), ys.tolist()))
        if key == "K1_fill_square":
            k = KAT[key]
            want = {(x, y) for x in range(k["x"][0], k["x"][1] + 1) for y in range(k["y"][0], k["y"][1] + 1)}
        else:
            want = {(x, int(y)) for y, (lo, hi) in KAT[key]["row_spans"].items() for x in range(lo, hi + 1)}
        assert got == want


def test_k4_k5_stroke_alpha(gpu_ctx):
    import math

    k = KAT["K4_stroke_h"]
    tb = TileBuilder(canvas=None)
    tb.stroke([k["p1"], k["p2"]], k["width"], (255, 255, 255))
    a = _alpha_of_single_op(gpu_ctx, tb.build())
    ys, xs = np.nonzero(a)
    assert set(ys.tolist()) == {9, 10, 11} and xs.min() == 10 and xs.max() == 14 and set(np.unique(a[a > 0])) == {1.0}
    k = KAT["K5_stroke_diag"]
    tb = TileBuilder(canvas=None)
    tb.stroke([k["p1"], k["p2"]], k["width"], (255, 255, 255))
    a = _alpha_of_single_op(gpu_ctx, tb.build())
    ys, xs = np.nonzero(a)
    assert set(zip(xs.tolist(), ys.tolist())) == {(x, int(y)) for y, xs_ in k["pixels"].items() for x in xs_}
    for y, xs_ in k["pixels"].items():
        for x in xs_:
            assert abs(a[int(y), x] - min(1.0, 1.5 - abs(3 * x - 7 * int(y) + 8) / math.sqrt(58))) < 1e-12


def test_k6_blend_chain(gpu_ctx, oracle):
    k = KAT["K6_blend"]
    tb = TileBuilder(canvas=tuple(k["canvas"]))
    sq = [(0, -1), (3, -1), (3, 3), (0, 3), (0, -1)]
    tb.fill(sq, tuple(k["steps"][0]["color"]), k["steps"][0]["alpha"])
    tb.fill(sq, tuple(k["steps"][1]["color"]), k["steps"][1]["alpha"])
    got = assert_parity(gpu_ctx, oracle, tb.build(), msg="K6")
    assert got[0, 0, 0, :3].tolist() == k["steps"][1]["rgb"]


def test_empty_and_degenerate(gpu_ctx, oracle):
    tb = TileBuilder()
    assert_parity(gpu_ctx, oracle, tb.build(), msg="no ops")  # canvas only
    tb = TileBuilder(canvas=None)
    tb.nop()
    tb.fill([(5, 5)], (1, 2, 3))  # single point: no edges
    tb.fill([(5, 5), (5, 5)], (1, 2, 3))  # zero-length edge
    tb.fill([(5, 5), (50, 5), (5, 5)], (200, 2, 3))  # horizontal only: every row poisoned
    tb.fill([(5, 5), (5, 60), (5, 5)], (3, 200, 3))  # zero-area vertical sliver
    tb.stroke([(7, 7)], 3.0, (9, 9, 9))
    tb.stroke([(7, 7), (7, 7)], 3.0, (9, 9, 9), cap=abi.CAP_ROUND)
    tb.stroke([(7, 7), (7, 7), (20, 9)], 3.0, (99, 9, 9), cap=abi.CAP_ROUND)  # degenerate first edge clears `first`
    tb.stroke([(30, 30), (60, 30)], 0.0, (9, 99, 9))  # zero width
    tb.fill([(100, 100), (140, 100), (140, 140), (100, 140), (100, 100)], (9, 9, 99), 0.0)  # zero opacity
    assert_parity(gpu_ctx, oracle, tb.build(), msg="degenerate")


def test_fills_random_polygons(gpu_ctx, oracle):
    rnd = np.random.default_rng(1234)
    tiles = []
    for t in range(6):
        tb = TileBuilder(canvas=(rnd.integers(0, 256), rnd.integers(0, 256), rnd.integers(0, 256)))
        for _ in range(25):
            n = int(rnd.integers(3, 12))
            c = rnd.integers(-40, 296, size=2)
            pts = (c + rnd.integers(-70, 71, size=(n, 2))).tolist()  # self-intersecting, any winding
            if rnd.random() < 0.8:
                pts.append(pts[0])  # closed (open contours are legal too: edges are just pairs)
            tb.fill(pts, tuple(rnd.integers(0, 256, size=3)), float(rnd.choice([1.0, 0.7, 0.33, 0.05])))
        tiles.append(tb.build())
    assert_parity(gpu_ctx, oracle, display_list.concat(tiles), msg="random fills")


def test_fill_far_outside_and_huge_edges(gpu_ctx, oracle):
    tb = TileBuilder()
    tb.fill([(-100000, -70000), (90000, -3000), (40000, 150000), (-100000, -70000)], (10, 200, 30), 0.6)
    tb.fill([(-5, -5), (300, -5), (300, 300), (-5, 300), (-5, -5)], (200, 20, 30), 0.4)  # covers the tile
    tb.fill([(1000, 1000), (2000, 1000), (1500, 3000), (1000, 1000)], (1, 2, 3))  # fully outside
    tb.fill([(-2000000, 10), (2000000, 200), (-2000000, 250), (-2000000, 10)], (0, 0, 255), 0.5)  # 4M px long
    tb.fill([(128, -3000000), (140, 3000000), (100, 3000000), (128, -3000000)], (255, 0, 255), 0.5)
    assert_parity(gpu_ctx, oracle, tb.build(), msg="huge fills")


def test_multipolygon_rings_share_one_edge_index(gpu_ctx, oracle):
    # outer ring + hole + island in the hole: even-odd over all rings together (fill.rs:19)
    outer = [(20, 20), (220, 25), (230, 210), (15, 200), (20, 20)]
    hole = [(60, 60), (180, 70), (170, 160), (70, 150), (60, 60)]
    island = [(100, 90), (140, 95), (135, 130), (100, 90)]
    tb = TileBuilder()
    tb.fill([outer, hole, island], (30, 90, 200), 0.8)
    tb.fill([hole, outer], (200, 90, 30), 0.5)  # ring order changes the stable-sort tie-breaks
    tb.fill([outer, outer], (9, 200, 9), 0.5)  # duplicated ring: pairs cancel differently
    assert_parity(gpu_ctx, oracle, tb.build(), msg="multipolygon")


def test_fill_many_crossings_uses_the_streaming_path(gpu_ctx, oracle):
    # a comb with > 32 crossings per row in one 32-px sub-tile column band (ROWCAP overflow)
    pts = []
    for i in range(60):
        x = 2 + 4 * i
        pts += [(x, 10), (x + 1, 200), (x + 2, 10)]
    pts += [(250, 5), (2, 5), (2, 10)]
    tb = TileBuilder(scale=1)
    tb.fill(pts, (10, 10, 200), 0.7)
    dense = []
    for i in range(80):  # 160 crossings inside x in [32, 64)
        x = 32 + (i * 31) % 32
        dense += [(x, 20 + (i % 7)), (x + (i % 3) - 1, 230 - (i % 11))]
    dense.append(dense[0])
    tb.fill(dense, (200, 10, 10), 0.6)
    assert_parity(gpu_ctx, oracle, tb.build(), msg="comb")


def test_image_fill(gpu_ctx, oracle):
    rnd = np.random.default_rng(5)
    icon = rnd.integers(0, 256, size=(7, 5, 4), dtype=np.uint8)
    icon[0, 0, 3] = 0
    icon[1, 1, 3] = 255
    icon2 = rnd.integers(0, 256, size=(16, 16, 4), dtype=np.uint8)
    ids = [gpu_ctx.register_image(icon), gpu_ctx.register_image(icon2)]
    assert ids[1] == ids[0] + 1
    images = [None] * ids[0] + [icon, icon2]
    images = [im if im is not None else np.zeros((1, 1, 4), np.uint8) for im in images]
    tb = TileBuilder(scale=2)
    tb.fill([(10, 10), (400, 40), (300, 480), (30, 300), (10, 10)], (0, 0, 0), 1.0)
    tb.fill_image([(50, 50), (450, 90), (350, 400), (60, 350), (50, 50)], ids[0], opacity=0.3)  # opacity ignored
    tb.fill_image([(-20, 200), (600, 220), (300, 520), (-20, 200)], ids[1])
    tb.fill_image([(0, 0), (10, 0), (10, 10), (0, 0)], 9999)  # unknown icon: draws nothing
    assert_parity(gpu_ctx, oracle, tb.build(), images=images, msg="image fill")


@pytest.mark.parametrize("scale", [1, 2])
def test_strokes_all_caps_dashes_widths(gpu_ctx, oracle, scale):
    rnd = np.random.default_rng(77 + scale)
    dash_sets = [None, [3, 3], [10, 8], [6, 6], [1, 2, 3], [0.5, 0.5], [12, 3, 2, 3], [4]]
    widths = [0.0, 0.1, 0.5, 1.0, 1.5, 2.0, 3.0, 4.5, 7.0, 15.0]
    tiles = []
    W = 256 * scale
    for t in range(10):
        tb = TileBuilder(scale=scale, canvas=(252, 248, 228))
        for _ in range(14):
            n = int(rnd.integers(2, 7))
            p0 = rnd.integers(-20, W + 20, size=2)
            pts = (p0 + np.cumsum(rnd.integers(-60 * scale, 60 * scale + 1, size=(n, 2)), axis=0)).tolist()
            d = dash_sets[int(rnd.integers(0, len(dash_sets)))]
            tb.stroke(
                pts,
                float(widths[int(rnd.integers(0, len(widths)))]) * scale,
                tuple(rnd.integers(0, 256, size=3)),
                float(rnd.choice([1.0, 0.6, 0.3])),
                dashes=None if d is None else [v * scale for v in d],
                cap=CAPS[int(rnd.integers(0, 4))],
                use_caps_for_dashes=bool(rnd.integers(0, 2)),
            )
        tiles.append(tb.build())
    assert_parity(gpu_ctx, oracle, display_list.concat(tiles), msg=f"strokes scale {scale}")


def test_negative_and_huge_widths(gpu_ctx, oracle):
    """The calculator works with sqrt(h*h - cap_dist^2), i.e. with |half_width| (opacity_calculator.rs:36): a negative
    width draws like its absolute value, and the stroke cull must size its reach from |h|.  Very wide lines cover
    whole sub-tiles far from their centre line; widths beyond the supported range are refused, not mis-drawn."""
    from osm_renderer_amd.lib import OsmtError

    tb = TileBuilder(canvas=(252, 248, 228))
    tb.stroke([(30, 40), (200, 90), (120, 220)], -9.0, (200, 30, 30), 0.8)
    tb.stroke([(10, 200), (240, 180)], -3.0, (30, 30, 200), 1.0, dashes=[6, 4], cap=abi.CAP_ROUND, use_caps_for_dashes=True)
    tb.stroke([(128, -40), (140, 300)], 90.0, (20, 160, 60), 0.5, cap=abi.CAP_SQUARE)
    tb.stroke([(-500, 128), (700, 131)], 300.0, (250, 250, 0), 0.3)
    assert_parity(gpu_ctx, oracle, tb.build(), msg="negative / huge widths")
    bad = TileBuilder()
    bad.stroke([(0, 0), (10, 10)], 1.0e6, (0, 0, 0), 1.0)
    with pytest.raises(OsmtError) as e:
        gpu_ctx.upload(bad.build())
    assert e.value.code == abi.UNSUPPORTED and "width" in str(e.value)


def test_stroke_every_direction(gpu_ctx, oracle):
    # direction sensitivity (SURVEY.md 7 "hard parts"): all octants, both orientations, steep & shallow
    tb = TileBuilder()
    c = (128, 128)
    k = 0
    for dx in range(-6, 7):
        for dy in (-6, -3, -1, 0, 1, 3, 6):
            if dx == 0 and dy == 0:
                continue
            e = (c[0] + 17 * dx, c[1] + 17 * dy)
            tb.stroke([c, e] if k % 2 else [e, c], 1.0 + (k % 5), ((37 * k) % 256, (91 * k) % 256, (53 * k) % 256), 0.5,
                      cap=CAPS[k % 4])
            k += 1
    assert_parity(gpu_ctx, oracle, tb.build(), msg="directions")


def test_stroke_long_segments_crossing_the_tile(gpu_ctx, oracle):
    tb = TileBuilder(scale=2)
    tb.stroke([(-30000, -20000), (30500, 20400)], 9.0, (200, 0, 0), 0.7, dashes=[30, 10], cap=abi.CAP_ROUND,
              use_caps_for_dashes=True)
    tb.stroke([(100, -400000), (130, 400000)], 5.0, (0, 0, 200), 0.7)
    tb.stroke([(-400000, 300), (400000, 310)], 3.0, (0, 200, 0), 0.7, dashes=[7, 7])
    tb.stroke([(1000, 1000), (5000, 5000)], 20.0, (1, 1, 1))  # outside
    tb.stroke([(515, 100), (530, 400)], 12.0, (9, 9, 9), cap=abi.CAP_SQUARE)  # only its feather reaches the tile
    assert_parity(gpu_ctx, oracle, tb.build(), msg="long segments")


def test_polyline_traveled_distance_and_caps(gpu_ctx, oracle):
    # dash phase carries across edges (line.rs:31); caps only on the first / last edge (line.rs:33-57)
    pts = [(20, 200), (60, 30), (60, 30), (110, 190), (160, 40), (230, 220)]
    tb = TileBuilder()
    for i, cap in enumerate(CAPS):
        off = [(x, y + 6 * i - 9) for x, y in pts]
        tb.stroke(off, 4.0, (20 + 50 * i, 10, 200 - 40 * i), 0.9, dashes=[9, 5], cap=cap, use_caps_for_dashes=(i % 2 == 0))
    assert_parity(gpu_ctx, oracle, tb.build(), msg="traveled")


def test_many_ops_per_tile_more_than_one_chunk(gpu_ctx, oracle):
    rnd = np.random.default_rng(3)
    tb = TileBuilder()
    for i in range(700):  # > 256 ops: several cull chunks; interleaved kinds keep the order honest
        p = rnd.integers(0, 256, size=2)
        if i % 3 == 0:
            q = p + rnd.integers(-40, 41, size=2)
            tb.stroke([p.tolist(), q.tolist()], float(rnd.choice([1, 2, 4])), tuple(rnd.integers(0, 256, size=3)), 0.6)
        elif i % 3 == 1:
            r = rnd.integers(4, 30)
            tb.fill([(p[0] - r, p[1] - r), (p[0] + r, p[1] - r // 2), (p[0], p[1] + r), (p[0] - r, p[1] - r)],
                    tuple(rnd.integers(0, 256, size=3)), 0.5)
        else:
            tb.nop()
    assert_parity(gpu_ctx, oracle, tb.build(), msg="many ops")


def test_no_canvas_and_scales(gpu_ctx, oracle):
    for scale in (1, 2, 3, 4):
        tb = TileBuilder(scale=scale, canvas=None)
        s = scale
        tb.fill([(10 * s, 10 * s), (200 * s, 30 * s), (120 * s, 240 * s), (10 * s, 10 * s)], (250, 120, 20), 0.75)
        tb.stroke([(0, 0), (255 * s, 255 * s)], 3.0 * s, (255, 255, 255), 0.5, dashes=[5 * s, 5 * s])
        assert_parity(gpu_ctx, oracle, tb.build(), msg=f"scale {scale}")


def test_dense_config5_tile(gpu_ctx, oracle):
    from osm_renderer_amd import synth

    dl = synth.make_tiles(synth.config_tiles(1, x0=79000, y0=40000), zoom=17, n_poly=1500, n_line=1200,
                          radius=(2.0, 12.0), step=12.0)
    assert_parity(gpu_ctx, oracle, dl, msg="dense")


def test_long_ways_and_multipolygons_use_the_block_path(gpu_ctx, oracle):
    """Ops with more than 64 edges: several record rounds per stroke, 64-edge block culling for
    strokes and fills, cap stubs in the last round, rings crossing block boundaries."""
    t = np.linspace(0, 9 * np.pi, 400)
    spiral = np.stack([128 + (6 + 4.1 * t) * np.cos(t), 128 + (6 + 4.1 * t) * np.sin(t)], 1).round().astype(int).tolist()
    wave = [(int(x), int(130 + 60 * np.sin(x / 17.0))) for x in range(-300, 600, 3)]
    tb = TileBuilder()
    tb.stroke(spiral, 3.0, (200, 30, 30), 0.8, dashes=[7, 4], cap=abi.CAP_ROUND, use_caps_for_dashes=True)
    tb.stroke(wave, 6.0, (30, 30, 200), 0.6, cap=abi.CAP_SQUARE)
    ring_a = [(int(128 + 110 * np.cos(a)), int(128 + 100 * np.sin(a))) for a in np.linspace(0, 2 * np.pi, 150)]
    ring_b = [(int(128 + 60 * np.cos(a)), int(120 + 55 * np.sin(-a))) for a in np.linspace(0, 2 * np.pi, 90)]
    ring_c = [(int(300 + 90 * np.cos(a)), int(40 + 80 * np.sin(a))) for a in np.linspace(0, 2 * np.pi, 70)]
    for r in (ring_a, ring_b, ring_c):
        r[-1] = r[0]
    tb.fill([ring_a, ring_b, ring_c], (20, 160, 60), 0.7)
    tb.stroke(ring_a, 1.0, (0, 0, 0), 1.0)
    dl1 = tb.build()
    # the same at @2x with everything shifted far to the left: most blocks are culled
    tb2 = TileBuilder(scale=2)
    far = [(x * 2 - 700, y * 2) for x, y in wave]
    tb2.stroke(far, 9.0, (10, 10, 10), 0.9, dashes=[20, 6])
    tb2.fill([[(x * 2 - 300, y * 2 + 40) for x, y in ring_a]], (250, 200, 10), 0.5)
    assert_parity(gpu_ctx, oracle, dl1, msg="long ops")
    assert_parity(gpu_ctx, oracle, tb2.build(), msg="long ops @2x")


def test_two_stroke_ops_share_one_ring(gpu_ctx, oracle):
    """A casing and its stroke drawn from the SAME ring (what a host that does not duplicate a way's points per pass
    emits; drawer.rs:186-215 draws both passes from one entity): the per-edge tables of the pre-pass are keyed by
    (op, edge), not by point, so the two ops' sub-tile windows (which depend on each op's width) do not collide."""
    tb = TileBuilder(canvas=(250, 250, 240))
    way = [(20, 30), (90, 60), (150, 40), (200, 120), (120, 200), (40, 170)]
    tb.fill([(10, 10), (240, 20), (230, 240), (20, 230), (10, 10)], (200, 220, 200), 0.8)
    tb.stroke(way, 11.0, (90, 60, 30), 1.0, cap=abi.CAP_ROUND)                       # casing
    tb.stroke(way, 6.0, (250, 240, 120), 0.9, dashes=[9.0, 5.0], cap=abi.CAP_BUTT)   # stroke, other width, dashed
    tb.stroke(way, 1.0, (0, 0, 0), 0.5)                                              # centre line
    dl = tb.build()
    ring_of_casing = int(dl.ops["ring_off"][1])
    dl.ops["ring_off"][2] = ring_of_casing  # share: ops 2 and 3 now reference the casing's ring
    dl.ops["ring_off"][3] = ring_of_casing
    assert_parity(gpu_ctx, oracle, dl, msg="shared ring")
    # and a multi-ring stroke op that shares its rings with a single-ring one (traveled differs per op)
    tb = TileBuilder(canvas=None)
    tb.stroke(way[:3], 5.0, (255, 0, 0), 1.0, dashes=[6.0, 4.0], cap=abi.CAP_ROUND, use_caps_for_dashes=True)
    tb.stroke(way[3:], 5.0, (0, 255, 0), 1.0, dashes=[6.0, 4.0])
    tb.stroke(way[3:], 3.0, (0, 0, 255), 0.7, dashes=[6.0, 4.0])
    dl = tb.build()
    dl.ops["n_rings"][2] = 2  # op 2 = rings 0 and 1 in one draw_lines call: traveled runs on across the rings
    dl.ops["ring_off"][2] = 0
    assert_parity(gpu_ctx, oracle, dl, msg="shared rings, multi-ring op")


@pytest.mark.parametrize("n_ops", [7, 8, 9, 31, 32, 33, 63, 64, 65, 97])
def test_list_lengths_around_every_chunk_and_stage_boundary(gpu_ctx, oracle, n_ops):
    """k_raster stages a sub-tile's list OPCHUNK = 32 entries at a time and keeps the coverage words / calculator constants
    of the first STAGECAP = 8 fills / strokes of a chunk in LDS (the others read the arena: scalar loads for a fill's
    words); the stroke slots of consecutive entries share a filter pass while they fit SEGCAP = 32 lanes.  Lists of
    exactly, one fewer and one more than each of these lengths, all landing in ONE sub-tile, kinds interleaved in three
    patterns (fills first, strokes first, alternating), image fills among the unstaged ones."""
    rnd = np.random.default_rng(100 + n_ops)
    icon = rnd.integers(0, 256, size=(7, 5, 4)).astype(np.uint8)
    icon[:3, :, 3] = 255
    img_id = gpu_ctx.register_image(icon)
    images = [np.zeros((1, 1, 4), dtype=np.uint8)] * img_id + [icon]
    tiles = []
    for pattern in range(3):
        tb = TileBuilder(x=pattern, canvas=(250, 248, 240))
        for i in range(n_ops):
            is_fill = (i < n_ops // 2) if pattern == 0 else (i >= n_ops // 2) if pattern == 1 else (i % 2 == 0)
            col = tuple(int(v) for v in rnd.integers(0, 256, size=3))
            # everything inside the sub-tile at (64..95, 32..47) and a little around it
            cx, cy = int(rnd.integers(60, 100)), int(rnd.integers(28, 52))
            if is_fill:
                r = int(rnd.integers(3, 12))
                ring = [(cx - r, cy - r), (cx + r, cy - r // 2), (cx + r // 2, cy + r), (cx - r, cy + r // 3), (cx - r, cy - r)]
                if i % 7 == 3:
                    tb.fill_image([ring], img_id)
                else:
                    tb.fill([ring], col, float(rnd.choice([1.0, 0.5, 0.25])))
            else:
                pts = [(cx, cy), (cx + int(rnd.integers(-14, 15)), cy + int(rnd.integers(-10, 11))), (cx + int(rnd.integers(-20, 21)), cy + 3)]
                tb.stroke(pts, float(rnd.choice([0.5, 1.0, 2.0, 3.5])), col, float(rnd.choice([1.0, 0.6])),
                          dashes=[3.0, 2.0] if i % 5 == 0 else None, cap=[abi.CAP_NONE, abi.CAP_ROUND, abi.CAP_SQUARE, abi.CAP_BUTT][i % 4])
        tiles.append(tb.build())
    from osm_renderer_amd.display_list import concat

    assert_parity(gpu_ctx, oracle, concat(tiles), images=images, msg=f"{n_ops} ops in one sub-tile")


def test_filter_groups_cut_by_slots_and_by_kept_records(gpu_ctx, oracle):
    """k_raster filters the stroke slots of a GROUP of list entries in one pass (256 slots, 32 kept records).  Every way
    a group can end: (a) many small ops whose records all sit in ONE sub-tile (the 33rd kept record ends the group in front
    of its entry), (b) one op with more than 32 records in one sub-tile (filtered and walked 32 slots at a time), (c) ops
    with more than 256 slots (long ways) between small ones, (d) fills and no-ops between the strokes of a group, dashed
    and capped ops among them, more ops than a chunk holds."""
    rnd = np.random.default_rng(11)

    def squiggle(x0, y0, n, reach):
        p = np.array([x0, y0]) + np.cumsum(rnd.integers(-reach, reach + 1, size=(n + 1, 2)), axis=0)
        return p.tolist()

    tb = TileBuilder(canvas=(240, 236, 225))
    # (a) 14 ops x 7 tiny edges inside the sub-tile (32..63, 16..31); widths / opacities / colours differ, two are dashed
    for i in range(14):
        pts = [(40 + int(rnd.integers(0, 16)), 20 + int(rnd.integers(0, 8))) for _ in range(8)]
        tb.stroke(pts, float(rnd.choice([0.7, 1.0, 2.0, 3.0])), tuple(rnd.integers(0, 256, size=3)), float(rnd.choice([1.0, 0.6, 0.3])),
                  dashes=[3.0, 2.0] if i in (3, 9) else None, cap=CAPS[i % 4])
        if i % 4 == 1:
            tb.fill([(36, 18), (60, 19), (50, 30), (36, 18)], tuple(rnd.integers(0, 256, size=3)), 0.4)
        if i % 5 == 2:
            tb.nop()
    # (b) one op with 45 tiny edges in one sub-tile, then one with 70 edges spread over two
    tb.stroke([(100 + int(rnd.integers(0, 20)), 70 + int(rnd.integers(0, 9))) for _ in range(46)], 1.5, (10, 80, 200), 0.7)
    tb.stroke([(130 + int(rnd.integers(0, 50)), 100 + int(rnd.integers(0, 12))) for _ in range(71)], 1.0, (200, 80, 10), 0.9, cap=abi.CAP_ROUND)
    # (c) long ways with hundreds of slots, small ops in front of, between and behind them
    tb.stroke(squiggle(20, 200, 5, 6), 2.0, (0, 0, 0), 1.0)
    tb.stroke([(int(x), int(128 + 100 * np.sin(x / 23.0))) for x in range(-40, 300, 9)], 2.5, (120, 20, 160), 0.8, dashes=[9.0, 5.0])
    tb.stroke(squiggle(200, 40, 6, 5), 1.0, (20, 120, 20), 0.5, cap=abi.CAP_SQUARE)
    tb.stroke([(int(128 + (10 + 3.5 * t) * np.cos(t)), int(128 + (10 + 3.5 * t) * np.sin(t))) for t in np.linspace(0, 8 * np.pi, 120)], 4.0, (250, 200, 0), 0.6)
    tb.stroke(squiggle(128, 128, 7, 4), 3.0, (0, 90, 90), 0.75)
    # (d) many more small strokes over the same few sub-tiles: several chunks per list, groups of every size
    for i in range(60):
        x0, y0 = 32 * int(rnd.integers(1, 4)) + int(rnd.integers(0, 32)), 16 * int(rnd.integers(1, 4)) + int(rnd.integers(0, 16))
        tb.stroke(squiggle(x0, y0, int(rnd.integers(1, 7)), 5), float(rnd.choice([0.5, 1.0, 1.5, 4.0])), tuple(rnd.integers(0, 256, size=3)),
                  float(rnd.choice([1.0, 0.5])), cap=CAPS[int(rnd.integers(0, 4))])
        if i % 7 == 0:
            tb.fill([(x0 - 9, y0 - 7), (x0 + 12, y0 - 3), (x0 + 2, y0 + 11), (x0 - 9, y0 - 7)], tuple(rnd.integers(0, 256, size=3)), 0.5)
    assert_parity(gpu_ctx, oracle, tb.build(), msg="filter groups")


def test_small_batches_build_their_lists_in_the_raster_kernel(gpu_ctx, oracle):
    """Batches of at most 64 tiles: tiles with at most 128 ops get no lists from k_sublist, their sub-tile waves read the op
    bits themselves (k_raster<FOLD>).  One batch with tiles on both sides of the limit — 0, 1, 64, 65, 127, 128, 129 and 300
    ops, fills, strokes and no-ops mixed so that list order matters — and the same tiles in a batch of 70 (lists for all)."""
    from osm_renderer_amd import display_list

    rnd = np.random.default_rng(23)

    def tile(n_ops):
        tb = TileBuilder(canvas=(250, 245, 230))
        for i in range(n_ops):
            p = rnd.integers(10, 246, size=2)
            k = i % 4
            if k == 0:
                q = p + rnd.integers(-60, 61, size=2)
                tb.stroke([p.tolist(), q.tolist()], float(rnd.choice([1.0, 2.5, 5.0])), tuple(rnd.integers(0, 256, size=3)), float(rnd.choice([1.0, 0.5])))
            elif k == 1:
                r = int(rnd.integers(5, 40))
                tb.fill([(p[0] - r, p[1] - r), (p[0] + r, p[1] - r // 3), (p[0] + r // 2, p[1] + r), (p[0] - r, p[1] - r)],
                        tuple(rnd.integers(0, 256, size=3)), float(rnd.choice([1.0, 0.6])))
            elif k == 2:
                tb.nop()
            else:
                pts = (p + np.cumsum(rnd.integers(-25, 26, size=(4, 2)), axis=0)).tolist()
                tb.stroke(pts, 3.0, tuple(rnd.integers(0, 256, size=3)), 0.8, dashes=[6.0, 3.0], cap=CAPS[i % 4])
        return tb.build()

    tiles = [tile(n) for n in (0, 1, 64, 65, 127, 128, 129, 300)]
    small = display_list.concat(tiles)
    got_small = assert_parity(gpu_ctx, oracle, small, msg="small batch, lists folded")
    big = display_list.concat(tiles + [tile(3) for _ in range(62)])  # 70 tiles: every tile has lists
    got_big = assert_parity(gpu_ctx, oracle, big, f64_jobs=[5, 6], msg="the same tiles with lists")
    assert np.array_equal(got_small, got_big[: len(tiles)])


@pytest.mark.parametrize("hits", [1, 63, 64, 65, 127, 128, 129, 200])
def test_sublist_sift_ring_boundaries(gpu_ctx, oracle, hits):
    """k_sublist (round 6) sifts a tile's ops per sub-tile ROW into a 128-entry ring in LDS — the ops that draw into the row, in op
    order — and builds the row's eight lists from 64 sifted ops at a time.  A tile whose row 2 is hit by exactly `hits` ops (one
    fewer, exactly and one more than a batch and than the ring), scattered among ops that draw elsewhere or nowhere, in a batch of
    more than 64 tiles (smaller batches build their lists in k_raster<FOLD>); a second tile piles the same ops into one column."""
    from osm_renderer_amd.display_list import concat

    rnd = np.random.default_rng(900 + hits)
    tiles = []
    for variant in range(2):
        tb = TileBuilder(x=variant, canvas=(245, 240, 230))
        n_total = hits * 3 + 40
        chosen = set(rnd.choice(n_total, size=hits, replace=False).tolist())
        for i in range(n_total):
            col = tuple(int(v) for v in rnd.integers(0, 256, size=3))
            if i in chosen:  # inside sub-tile row 2 (y in 32..47): a short thin stroke or a small box
                x = int(rnd.integers(4, 250)) if variant == 0 else int(rnd.integers(70, 90))
                y = int(rnd.integers(35, 45))
                if i % 3:
                    tb.stroke([(x, y), (x + int(rnd.integers(2, 6)), y + int(rnd.integers(-1, 2)))], 1.0, col, 0.7)
                else:
                    tb.fill([[(x, y - 1), (x + 3, y - 1), (x + 3, y + 2), (x, y + 2), (x, y - 1)]], col, 0.5)
            elif i % 4 == 0:
                tb.nop()
            else:  # far from row 2 (rows 5 and below), or outside the tile altogether
                x, y = int(rnd.integers(4, 250)), int(rnd.integers(90, 250)) if i % 4 != 3 else int(rnd.integers(300, 400))
                tb.stroke([(x, y), (x + 4, y + 1)], 1.0, col, 0.9)
        tiles.append(tb.build())
    empty = TileBuilder(x=9, canvas=(1, 2, 3)).build()
    dl = concat(tiles + [empty] * 64)  # 66 tiles: every tile's lists come from k_sublist
    assert_parity(gpu_ctx, oracle, dl, f64_jobs=[0, 1], msg=f"{hits} ops in one sub-tile row")
