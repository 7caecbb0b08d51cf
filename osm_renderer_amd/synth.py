"""Deterministic synthetic geometry of BASELINE.json's configs.

Per tile, in draw order: `n_poly` FILL ops (closed 9-point rings = 8 edges) then
`n_line` STROKE ops (6-point polylines = 5 segments), i.e. the "50-poly /
200-segment" tile for the defaults.  Every tile owns a SplitMix64 stream seeded
with 0x05EED ^ (zoom << 48) ^ (x << 24) ^ y; the k-th draw of a stream is
mix(seed + (k+1)*GOLDEN), so generation is vectorised over tiles.  Vertices are
produced in scaled tile pixels and converted to (lat, lon) by the inverse Web
Mercator so the projection of the reference (src/tile.rs:88-106) is exercised.
"""
import numpy as np

from . import abi
from .display_list import JOB_DTYPE, OP_DTYPE, RING_DTYPE, DisplayList

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

CANVAS_OSMOSNIMKI = (0xFC, 0xF8, 0xE4)  # mapcss/osmosnimki-minimal.mapcss:1-4
WIDTHS = np.array([0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, 9.0])
WIDTH_CDF = np.cumsum([0.20, 0.25, 0.10, 0.10, 0.10, 0.15, 0.06, 0.04])
DASH_PATTERNS = np.array([[3.0, 3.0], [10.0, 8.0], [6.0, 6.0]])


def splitmix64_draws(seeds, n_draws):
    """[len(seeds), n_draws] uint64: draw k of stream i = mix(seed_i + (k+1)*GOLDEN)."""
    seeds = np.asarray(seeds, dtype=np.uint64).reshape(-1, 1)
    k = np.arange(1, n_draws + 1, dtype=np.uint64).reshape(1, -1)
    with np.errstate(over="ignore"):
        z = seeds + k * _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform(z):
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def tile_seed(zoom, x, y):
    return (
        np.uint64(0x05EED)
        ^ (np.asarray(zoom, dtype=np.uint64) << np.uint64(48))
        ^ (np.asarray(x, dtype=np.uint64) << np.uint64(24))
        ^ np.asarray(y, dtype=np.uint64)
    )


def pixels_to_latlon(px, py, zoom, tx, ty, scale):
    """Inverse of coords_to_xy_tile_relative * scale (tile.rs:88-106)."""
    dim = 256.0 * (2.0**zoom)
    wx = tx * 256.0 + px / scale
    wy = ty * 256.0 + py / scale
    lon = wx / dim * 360.0 - 180.0
    lat = np.degrees(np.arctan(np.sinh(np.pi * (1.0 - 2.0 * wy / dim))))
    return lat, lon


def make_tiles(
    tiles_xy,
    zoom=15,
    scale=1,
    n_poly=50,
    n_line=40,
    radius=(8.0, 48.0),
    step=48.0,
    canvas=CANVAS_OSMOSNIMKI,
    caps_for_dashes_prob=0.0,
    coord_kind=abi.COORD_LATLON_F64,
):
    """DisplayList for tiles [(x, y), ...] at `zoom`, integer `scale`."""
    tiles_xy = np.asarray(tiles_xy, dtype=np.int64).reshape(-1, 2)
    n = len(tiles_xy)
    tx = tiles_xy[:, 0].astype(np.float64).reshape(-1, 1)
    ty = tiles_xy[:, 1].astype(np.float64).reshape(-1, 1)
    s = float(scale)
    W = 256.0 * s
    POLY_D, LINE_D = 21, 21
    draws = splitmix64_draws(tile_seed(zoom, tiles_xy[:, 0], tiles_xy[:, 1]), n_poly * POLY_D + n_line * LINE_D)
    u = _uniform(draws)

    # ---- polygons: [n, n_poly, 21] -------------------------------------------------
    up = u[:, : n_poly * POLY_D].reshape(n, n_poly, POLY_D)
    zp = draws[:, : n_poly * POLY_D].reshape(n, n_poly, POLY_D)
    cx = -32.0 + up[:, :, 0] * (W + 64.0)
    cy = -32.0 + up[:, :, 1] * (W + 64.0)
    ang = 2.0 * np.pi * np.arange(8).reshape(1, 1, 8) / 8.0 + (up[:, :, 2:10] * 0.6 - 0.3)
    rad = (radius[0] + up[:, :, 10:18] * (radius[1] - radius[0])) * s
    vx = cx[:, :, None] + rad * np.cos(ang)
    vy = cy[:, :, None] + rad * np.sin(ang)
    vx = np.concatenate([vx, vx[:, :, :1]], axis=2)  # closed ring: first == last
    vy = np.concatenate([vy, vy[:, :, :1]], axis=2)
    pcol = (zp[:, :, 18:21] & np.uint64(0xFF)).astype(np.uint8)
    uo = up[:, :, 20]  # re-uses the blue draw's fraction for the opacity class
    pop = np.where(uo < 0.6, 1.0, np.where(uo < 0.8, 0.7, 0.5))

    # ---- polylines: [n, n_line, 21] --------------------------------------------------
    ul = u[:, n_poly * POLY_D :].reshape(n, n_line, LINE_D)
    zl = draws[:, n_poly * POLY_D :].reshape(n, n_line, LINE_D)
    sx = -16.0 + ul[:, :, 0] * (W + 32.0)
    sy = -16.0 + ul[:, :, 1] * (W + 32.0)
    dxs = (ul[:, :, 2:7] * 2.0 - 1.0) * step * s
    dys = (ul[:, :, 7:12] * 2.0 - 1.0) * step * s
    lx = np.concatenate([sx[:, :, None], sx[:, :, None] + np.cumsum(dxs, axis=2)], axis=2)
    ly = np.concatenate([sy[:, :, None], sy[:, :, None] + np.cumsum(dys, axis=2)], axis=2)
    widx = np.minimum(np.searchsorted(WIDTH_CDF, ul[:, :, 12], side="right"), len(WIDTHS) - 1)
    width = WIDTHS[widx] * s
    uo = ul[:, :, 13]
    lop = np.where(uo < 0.5, 1.0, np.where(uo < 0.8, 0.6, 0.3))
    dashed = ul[:, :, 14] < 0.25
    dash_idx = np.minimum((ul[:, :, 15] * 3).astype(np.int64), 2)
    uc = ul[:, :, 16]
    cap = np.where(uc < 0.6, abi.CAP_NONE, np.where(uc < 0.9, abi.CAP_ROUND, abi.CAP_SQUARE)).astype(np.uint8)
    lcol = (zl[:, :, 17:20] & np.uint64(0xFF)).astype(np.uint8)
    ucd = ul[:, :, 20] < caps_for_dashes_prob

    # ---- pools -------------------------------------------------------------------------
    npp, npl = 9, 6
    pts_per_tile = n_poly * npp + n_line * npl
    ops_per_tile = n_poly + n_line
    px = np.concatenate([vx.reshape(n, -1), lx.reshape(n, -1)], axis=1)
    py = np.concatenate([vy.reshape(n, -1), ly.reshape(n, -1)], axis=1)
    if coord_kind == abi.COORD_LATLON_F64:
        lat, lon = pixels_to_latlon(px, py, zoom, tx, ty, s)
        coords = np.stack([lat, lon], axis=2).reshape(-1, 2)
    else:
        coords = np.stack([np.round(px), np.round(py)], axis=2).reshape(-1, 2).astype(np.int32)

    jobs = np.zeros(n, JOB_DTYPE)
    jobs["x"], jobs["y"], jobs["zoom"] = tiles_xy[:, 0], tiles_xy[:, 1], zoom
    jobs["has_canvas"] = 1
    jobs["canvas_rgb"] = canvas
    jobs["n_ops"] = ops_per_tile
    jobs["op_off"] = np.arange(n) * ops_per_tile
    jobs["n_pts"] = pts_per_tile
    jobs["pt_off"] = np.arange(n) * pts_per_tile

    ops = np.zeros((n, ops_per_tile), OP_DTYPE)
    ops["n_rings"] = 1
    ops["ring_off"] = np.arange(n * ops_per_tile).reshape(n, ops_per_tile)
    ops["kind"][:, :n_poly] = abi.OP_FILL_COLOR
    ops["color"][:, :n_poly] = pcol
    ops["opacity"][:, :n_poly] = pop
    ops["kind"][:, n_poly:] = abi.OP_STROKE
    ops["color"][:, n_poly:] = lcol
    ops["opacity"][:, n_poly:] = lop
    ops["width"][:, n_poly:] = width
    ops["cap"][:, n_poly:] = cap
    ops["use_caps_for_dashes"][:, n_poly:] = ucd
    ops["has_dashes"][:, n_poly:] = dashed
    ops["n_dashes"][:, n_poly:] = np.where(dashed, 2, 0)
    dash_off = np.cumsum(np.where(dashed, 2, 0).reshape(-1)) - np.where(dashed, 2, 0).reshape(-1)
    ops["dashes_off"][:, n_poly:] = np.where(dashed, dash_off.reshape(n, n_line), 0)
    dashes = (DASH_PATTERNS[dash_idx[dashed]] * s).reshape(-1)

    rings = np.zeros((n, ops_per_tile), RING_DTYPE)
    local_first = np.concatenate([np.arange(n_poly) * npp, n_poly * npp + np.arange(n_line) * npl])
    rings["first_pt"] = jobs["pt_off"].reshape(-1, 1) + local_first.reshape(1, -1)
    rings["n_pts"][:, :n_poly] = npp
    rings["n_pts"][:, n_poly:] = npl

    return DisplayList(jobs, ops.reshape(-1), rings.reshape(-1), coords, dashes, coord_kind, scale)


def config_tiles(n_tiles, x0=19000, y0=10000, per_row=100):
    """Config 4's tile numbering: x = 19000 + (i mod 100), y = 10000 + i // 100."""
    i = np.arange(n_tiles)
    return np.stack([x0 + (i % per_row), y0 + i // per_row], axis=1)


def config2(n_tiles=1024, scale=1):
    """BASELINE.json configs[1]: z=15, 256x256, 50 polygons + 40 polylines (200 segments) per tile."""
    return make_tiles(config_tiles(n_tiles), zoom=15, scale=scale)


def config3(n_tiles=1024):
    """configs[2]: the same geometry at @2x (512x512)."""
    return make_tiles(config_tiles(n_tiles), zoom=15, scale=2)


def config5(n_tiles=8, scale=1):
    """configs[4]: dense city — 5000 polygons + 4000 polylines (20000 segments) per tile, z=17."""
    return make_tiles(config_tiles(n_tiles, x0=79000, y0=40000), zoom=17, scale=scale, n_poly=5000, n_line=4000,
                      radius=(2.0, 12.0), step=12.0)


def composite_planes(n_tiles, L=8, dim=512, seed=0xC0FFEE, device=None):
    """Config 3's composite-pass input: planes[n][L][dim][dim][4] premultiplied f64 =
    from_color(colour_l, alpha_l(x, y)); alpha field = 30% zeros, 40% ones, 30% U(0,1)."""
    import torch

    g = torch.Generator(device=device if device is not None else "cpu")
    g.manual_seed(seed)
    kw = dict(generator=g, device=device if device is not None else "cpu")
    col = torch.randint(0, 256, (n_tiles, L, 1, 1, 3), **kw).to(torch.float64) / 255.0
    sel = torch.rand((n_tiles, L, dim, dim), dtype=torch.float64, **kw)
    ua = torch.rand((n_tiles, L, dim, dim), dtype=torch.float64, **kw)
    alpha = torch.where(sel < 0.3, torch.zeros_like(ua), torch.where(sel < 0.7, torch.ones_like(ua), ua))
    planes = torch.empty((n_tiles, L, dim, dim, 4), dtype=torch.float64, device=alpha.device)
    planes[..., :3] = alpha.unsqueeze(-1) * col  # o * (c/255)   (tile_pixels.rs:12-19)
    planes[..., 3] = alpha
    return planes
