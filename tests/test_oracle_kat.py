"""Pins of the CPU oracle: the reference's projection doctests and the K1..K8 known-answer
vectors (tests/golden/kat.json; provenance in the file)."""
import json
import math
import os

import numpy as np
import pytest

from osm_renderer_amd import abi

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def test_projection_doctests(oracle):
    # src/tile.rs:83-86 assert_floor_eq(coords_to_xy(..), ..)
    for k in KAT["projection_coords_to_xy_floor"]:
        x, y = oracle.coords_to_xy(k["lat"], k["lon"], k["zoom"])
        assert (int(x), int(y)) == (k["x"], k["y"])
    # src/tile.rs:26-28 coords_to_max_zoom_tile
    for k in KAT["projection_max_zoom_tile"]:
        assert oracle.coords_to_max_zoom_tile(k["lat"], k["lon"]) == (k["x"], k["y"])


def _covered(alpha):
    ys, xs = np.nonzero(alpha)
    return set(zip(xs.tolist(), ys.tolist()))


def test_k1_fill_square(oracle):
    k = KAT["K1_fill_square"]
    p = oracle.Pixels(1)
    p.reset()
    p.fill_contour(oracle.ring_to_pairs(k["ring"]), (0, 0, 255), 1.0)
    want = {(x, y) for x in range(k["x"][0], k["x"][1] + 1) for y in range(k["y"][0], k["y"][1] + 1)}
    assert _covered(p.pending_alpha(0)) == want and len(want) == k["count"]


def test_k2_fill_triangle(oracle):
    k = KAT["K2_fill_triangle"]
    p = oracle.Pixels(1)
    p.reset()
    p.fill_contour(oracle.ring_to_pairs(k["ring"]), (9, 9, 9), 0.4)
    want = {(x, int(y)) for y, (a, b) in k["row_spans"].items() for x in range(a, b + 1)}
    assert _covered(p.pending_alpha(0)) == want
    assert set(np.unique(p.pending_alpha(0))) == {0.0, 0.4}


def test_k3_zingl_walk(oracle):
    for k in KAT["K3_zingl_walk"]:
        assert oracle.fill_edge_walk(k["p1"], k["p2"]).tolist() == k["pixels"]


def test_k4_stroke_horizontal(oracle):
    k = KAT["K4_stroke_h"]
    p = oracle.Pixels(1)
    p.reset()
    p.draw_lines(oracle.ring_to_pairs([k["p1"], k["p2"]]), k["width"], (0, 0, 0))
    a = p.pending_alpha(0)
    want = {(x, y) for x in range(k["x"][0], k["x"][1] + 1) for y in range(k["y"][0], k["y"][1] + 1)}
    assert _covered(a) == want
    assert set(np.unique(a[a > 0])) == {k["alpha"]}


def test_k5_stroke_diagonal(oracle):
    k = KAT["K5_stroke_diag"]
    p = oracle.Pixels(1)
    p.reset()
    p.draw_lines(oracle.ring_to_pairs([k["p1"], k["p2"]]), k["width"], (0, 0, 0))
    a = p.pending_alpha(0)
    want = {(x, int(y)) for y, xs in k["pixels"].items() for x in xs}
    assert _covered(a) == want
    for y, xs in k["pixels"].items():
        for x, a3 in zip(xs, k["alpha_3dp"][y]):
            exact = min(1.0, 1.5 - abs(3 * x - 7 * int(y) + 8) / math.sqrt(58))
            assert abs(a[int(y), x] - a3) < 6e-4
            assert abs(a[int(y), x] - exact) < 1e-12
    p.reset()
    p.draw_lines(oracle.ring_to_pairs([k["p2"], k["p1"]]), k["width"], (0, 0, 0))
    assert (_covered(p.pending_alpha(0)) != want) == k["reversed_differs"]


def test_k6_blend(oracle):
    k = KAT["K6_blend"]
    p = oracle.Pixels(1)
    p.reset(tuple(k["canvas"]))
    for st in k["steps"]:
        a = st["alpha"]
        p.set_pixel(0, 0, [a * (c / 255.0) for c in st["color"]] + [a])
        p.bump_generation()
        p.blend_unfinished_pixels()
        assert p.to_rgb()[0, 0].tolist() == st["rgb"]


def test_k7_across_profile(oracle):
    for k in KAT["K7_across_profile"]:
        op, inl = oracle.opacity_calculate(k["width"] / 2.0, None, abi.CAP_NONE, 0.0, k["cd"], 0.0)
        assert (op, inl) == (k["alpha"], k["in"])


def test_k8_u8_roundtrip(oracle):
    # 255 * ((v/255)/1.0) truncates back to v for every v (tile_pixels.rs:171-175)
    p = oracle.Pixels(1)
    for v in range(256):
        p.reset((v, 255 - v, (v * 7) % 256))
        assert p.to_rgb()[0, 0].tolist() == [v, 255 - v, (v * 7) % 256]


def test_canvas_alpha_stays_one(oracle):
    """the canvas alpha is exactly 1.0 after any blend sequence."""
    from osm_renderer_amd import synth

    dl = synth.config2(2)
    _, f64 = oracle.render_job(dl, 1, want_f64=True)
    assert np.all(f64[..., 3] == 1.0)


def test_default_canvas_is_opaque_black(oracle):
    p = oracle.Pixels(1)
    p.reset(None)
    assert p.to_rgb().max() == 0 and np.all(p.pixels_f64()[..., 3] == 1.0)


def test_same_generation_keeps_max_alpha_then_over(oracle):
    # tile_pixels.rs:107-129
    p = oracle.Pixels(1)
    p.reset((10, 20, 30))
    p.set_pixel(5, 5, [0.1, 0.1, 0.1, 0.2])
    p.set_pixel(5, 5, [0.3, 0.3, 0.3, 0.6])
    p.set_pixel(5, 5, [0.2, 0.2, 0.2, 0.6])  # equal alpha does not replace (strict >)
    assert p.pending_alpha(0)[5, 5] == 0.6
    p.bump_generation()
    p.set_pixel(5, 5, [0.0, 0.0, 0.0, 0.5])  # flushes generation 0 first
    f = p.pixels_f64()[5, 5]
    assert f[0] == 0.3 + (1.0 - 0.6) * (10 / 255.0)
    p.set_pixel(-1, 5, [1, 1, 1, 1])  # outside bb: ignored
    p.set_pixel(256, 5, [1, 1, 1, 1])
    p.blend_unfinished_pixels()
    assert p.pixels_f64()[5, 5][0] == 0.0 + (1.0 - 0.5) * f[0]


def test_push_away_from(oracle):
    # point.rs:27-35
    assert oracle.push_away_from((10, 10), (0, 10), 3.0) == (13, 10)
    assert oracle.push_away_from((0, 0), (3, 4), 5.0) == (-3, -4)
    assert oracle.push_away_from((0, 0), (1, 1), 1.0) == (-1, -1)  # round(0.7071) = 1 on both axes


def test_persistent_pool_equals_the_one_shot_renderer(oracle):
    """oracle_py.Pool (canvases allocated once per worker, http_server.rs:69-72) reuses its TilePixels across calls:
    same pixels as the allocate-per-call path, also on the second call and with labels."""
    import numpy as np

    from osm_renderer_amd import labels, synth

    dl = synth.config2(6)
    ll = labels.make_labels(6, labels_per_tile=4, seed=3)
    want = oracle.render_batch(dl, threads=2)
    want_l, st = oracle.render_batch(dl, threads=2, labels=ll, want_status=True)
    pool = oracle.Pool(3)
    out = np.zeros_like(want)
    for _ in range(2):
        pool.render(dl, out)
        assert np.array_equal(out, want)
    got_st = np.zeros_like(st)
    pool.render(dl, out, labels=ll, status=got_st)
    assert np.array_equal(out, want_l) and np.array_equal(got_st, st)
    pool.render(dl, out)  # labels of the previous call leave nothing behind
    assert np.array_equal(out, want)
    pool.close()
