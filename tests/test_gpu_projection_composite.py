"""GPU projection (tile.rs:88-106 + point.rs:11-19) and layer compositing
(tile_pixels.rs:205-223,164-181) against the oracle, through the C ABI."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def test_projection_doctest_values(gpu_ctx):
    # the reference's own known answers: floor(coords_to_xy) == round(x - 0.5) for non-integers;
    # check through the tile-relative integer path: the tile containing the point.
    for k in KAT["projection_coords_to_xy_floor"]:
        tx, ty = k["x"] // 256, k["y"] // 256
        xy = gpu_ctx.project([[k["lat"], k["lon"]]], k["zoom"], tx, ty, 1.0)[0]
        # round(v) is floor(v) or floor(v)+1
        assert xy[0] - (k["x"] - tx * 256) in (0, 1) and xy[1] - (k["y"] - ty * 256) in (0, 1)


def test_projection_matches_oracle_bit_exact(gpu_ctx, oracle):
    """OCML tan/log are not glibc's; a difference can only flip a round() tie.  Bar: identical
    integer points on 2M seeded coordinates (mismatch count reported, must be 0)."""
    rnd = np.random.default_rng(2024)
    total = 0
    for zoom, scale in [(15, 1.0), (15, 2.0), (17, 1.0), (18, 2.0), (5, 1.0), (0, 1.0)]:
        n = 350000
        lat = rnd.uniform(-84.0, 84.0, n)
        lon = rnd.uniform(-179.9, 179.9, n)
        tx, ty = (19807 if zoom == 15 else 3), (10243 if zoom == 15 else 2)
        ll = np.stack([lat, lon], axis=1)
        got = gpu_ctx.project(ll, zoom, tx, ty, scale)
        want = oracle.project_points(ll, zoom, tx, ty, scale)
        total += int((got != want).any(axis=1).sum())
    assert total == 0, f"{total} projected points differ from the oracle"


def test_projection_near_tile(gpu_ctx, oracle):
    # points of the tile of test_rendering.rs:152-155 (z15 19807/10243) and its neighbours
    rnd = np.random.default_rng(7)
    dim = 256.0 * 2**15
    wx = (19807 + rnd.uniform(-1, 2, 200000)) * 256.0
    wy = (10243 + rnd.uniform(-1, 2, 200000)) * 256.0
    lon = wx / dim * 360.0 - 180.0
    lat = np.degrees(np.arctan(np.sinh(np.pi * (1 - 2 * wy / dim))))
    ll = np.stack([lat, lon], axis=1)
    for scale in (1.0, 2.0):
        assert np.array_equal(gpu_ctx.project(ll, 15, 19807, 10243, scale), oracle.project_points(ll, 15, 19807, 10243, scale))


@pytest.mark.parametrize("L", [0, 1, 3, 4, 8])
def test_composite_matches_oracle(gpu_ctx, oracle, L):
    rnd = np.random.default_rng(100 + L)
    n, H, W = 3, 40, 56
    alpha = rnd.choice([0.0, 1.0, 0.25, 0.5], size=(n, L, H, W)) * rnd.choice([1.0, rnd.random()], size=(n, L, H, W))
    col = rnd.integers(0, 256, size=(n, L, 1, 1, 3)) / 255.0
    planes = np.empty((n, L, H, W, 4))
    planes[..., :3] = alpha[..., None] * col
    planes[..., 3] = alpha
    canvas = [0xF1 / 255.0, 0xEE / 255.0, 0xE8 / 255.0, 1.0]
    got = gpu_ctx.composite_host(planes, canvas)
    want = oracle.composite(planes, canvas)
    np.testing.assert_array_equal(got, want)


def test_composite_odd_inputs(gpu_ctx, oracle):
    # arbitrary (even inconsistent) premultiplied inputs are blended literally: alpha > 1,
    # negative, transparent canvas (a == 0 -> 0), NaN -> 0 after the saturating cast
    rnd = np.random.default_rng(9)
    planes = rnd.uniform(-0.5, 1.5, size=(2, 5, 16, 32, 4))
    planes[0, 2, 3, 4, :] = np.nan
    planes[1, :, 5, 6, :] = 0.0
    for canvas in ([0.2, 0.4, 0.6, 1.0], [0.0, 0.0, 0.0, 0.0]):
        np.testing.assert_array_equal(gpu_ctx.composite_host(planes, canvas), oracle.composite(planes, canvas))


def test_composite_config3_shape_and_identity(gpu_ctx, oracle):
    """Full-size shape (512x512, L=8): oracle parity on 2 tiles + size-independent properties on 24."""
    import torch

    from osm_renderer_amd import synth

    planes = synth.composite_planes(24, L=8, dim=512, device=gpu_ctx.device)
    canvas = [0xFC / 255.0, 0xF8 / 255.0, 0xE4 / 255.0, 1.0]
    out = gpu_ctx.composite(planes, canvas)
    torch.cuda.synchronize()
    want = oracle.composite(planes[:2].cpu().numpy(), canvas, threads=8)
    np.testing.assert_array_equal(out[:2].cpu().numpy(), want)
    # idempotence / determinism
    assert torch.equal(out, gpu_ctx.composite(planes, canvas))
    # a pixel whose top layer is opaque shows exactly that layer's colour; all-transparent shows the canvas
    top = planes[:, -1]
    opaque = top[..., 3] == 1.0
    exp = (255.0 * top[..., :3]).to(torch.uint8)  # truncation; 255*(c/255) round-trips (K8)
    assert torch.equal(out[..., :3][opaque], exp[opaque])
    transparent = (planes[..., 3] == 0.0).all(dim=1)
    cv = torch.tensor([0xFC, 0xF8, 0xE4], dtype=torch.uint8, device=out.device)
    assert transparent.any() and torch.equal(out[..., :3][transparent], cv.expand_as(out[..., :3][transparent]))
    assert bool((out[..., 3] == 255).all())
