"""Label pass, CPU side: the oracle's restatement of font/rasterizer.rs, tile_pixels.rs:131-162 and
labeler.rs:91-106 against hand-derived known answers (derived from the reference SOURCE: exact areas of
simple polygons, the collision rules of set_label_pixel), plus the host-side label lists."""
import numpy as np
import pytest

from osm_renderer_amd import labels
from osm_renderer_amd.display_list import TileBuilder


def _square(x0, y0, x1, y1):
    """left edge walked towards +y = the orientation that accumulates positive coverage (sign, rasterizer.rs:34)"""
    return [(x0, y0, x0, y1), (x0, y1, x1, y1), (x1, y1, x1, y0), (x1, y0, x0, y0)]


def test_rasterizer_exact_areas(oracle):
    xy, t = oracle.rasterizer_pixels(_square(2, 3, 5, 6))
    assert xy.tolist() == [[x, y] for y in (3, 4, 5) for x in (2, 3, 4)] and (t == 1.0).all()
    # the opposite orientation accumulates negative coverage: nothing is > 0 (rasterizer.rs:140)
    xy, t = oracle.rasterizer_pixels([(c, d, a, b) for (a, b, c, d) in _square(2, 3, 5, 6)][::-1])
    assert len(xy) == 0
    # half-open pixels: a square from 0.5 to 2.5 covers 1/4, 1/2, 1 of the pixels it touches
    xy, t = oracle.rasterizer_pixels(_square(0.5, 0.5, 2.5, 2.5))
    got = {tuple(p): v for p, v in zip(xy.tolist(), t)}
    assert got == {(0, 0): 0.25, (1, 0): 0.5, (2, 0): 0.25, (0, 1): 0.5, (1, 1): 1.0, (2, 1): 0.5, (0, 2): 0.25, (1, 2): 0.5, (2, 2): 0.25}
    # right triangle (0,0) (0,2) (2,0): pixel (0,0) fully inside, its two neighbours half
    xy, t = oracle.rasterizer_pixels([(0, 0, 0, 2), (0, 2, 2, 0), (2, 0, 0, 0)])  # same orientation
    assert {tuple(p): v for p, v in zip(xy.tolist(), t)} == {(0, 0): 1.0, (1, 0): 0.5, (0, 1): 0.5}
    # a hole (inner contour with the other orientation) subtracts; coverage is clamped at 1 (min(.., 1.0))
    outer, inner = _square(0, 0, 6, 6), [(c, d, a, b) for (a, b, c, d) in _square(2, 2, 4, 4)][::-1]
    xy, t = oracle.rasterizer_pixels(outer + inner)
    got = {tuple(p) for p in xy.tolist()}
    assert (2, 2) not in got and (3, 3) not in got and (1, 1) in got and len(got) == 32
    xy, t = oracle.rasterizer_pixels(outer + outer)
    assert len(xy) == 36 and (t == 1.0).all()
    # horizontal edges are ignored (delta == 0.0), visiting order is stripes by y then x ascending
    xy, t = oracle.rasterizer_pixels([(0, 1, 9, 1)])
    assert len(xy) == 0


def test_flatten_quad_matches_the_oracle(oracle):
    rng = np.random.default_rng(3)
    for _ in range(50):
        q = rng.uniform(-40, 300, size=6)
        out = []
        labels.flatten_quad(*q, out)
        ref = oracle.flatten_quad(*q)
        assert np.array_equal(np.array(out, dtype=np.float64).reshape(-1, 4).view(np.uint64), ref.view(np.uint64))
    # a straight "curve" is one line (d01 + d12 <= 1.0001 * d02)
    assert len(oracle.flatten_quad(0, 0, 5, 5, 10, 10)) == 1


def _text(color, x0, y0, x1, y1):
    return (color, np.array(_square(x0, y0, x1, y1), dtype=np.float64))


def test_label_collisions_follow_set_label_pixel(oracle):
    """A succeeds; B overlaps A -> fails and leaves no trace; C overlaps only B -> succeeds (B's pixels are
    overwritten freely); D collides with A far outside the tile (labels_bb is 3x3 tiles) -> fails; E lies
    completely outside labels_bb -> every set_label_pixel returns true -> succeeds, draws nothing."""
    tb = TileBuilder(canvas=(255, 255, 255))
    dl = tb.build()
    tl = labels.TileLabels()
    tl.label(text=_text((255, 0, 0), 10, 10, 20, 20))      # A
    tl.label(text=_text((0, 255, 0), 15, 15, 30, 30))      # B: hits A
    tl.label(text=_text((0, 0, 255), 25, 25, 40, 40))      # C: overlaps B only
    tl.label(text=_text((9, 9, 9), -200, -200, -190, -190))  # A2 outside the tile, inside labels_bb
    tl.label(text=_text((7, 7, 7), -195, -195, -100, -100))  # D: hits A2
    tl.label(text=_text((5, 5, 5), 900, 900, 950, 950))      # E: outside labels_bb
    tl.label(text=_text((1, 2, 3), 256, 100, 300, 120))      # F: starts right at the tile edge (x = 256 is outside the tile)
    ll = tl.build()
    out, status = oracle.render_job(dl, 0, labels=ll, want_status=True)
    assert status.tolist() == [1, 0, 1, 1, 0, 1, 1]
    assert out[12, 12, :3].tolist() == [255, 0, 0] and out[17, 17, :3].tolist() == [255, 0, 0]
    assert out[22, 22, :3].tolist() == [255, 255, 255]  # B never blended
    assert out[27, 27, :3].tolist() == [0, 0, 255]
    assert (out[100:120, 250:256, :3] == 255).all()


def test_icon_then_text_and_generation_rules(oracle):
    icon = np.zeros((4, 6, 4), dtype=np.uint8)  # 6 wide, 4 high
    icon[..., 0] = 200
    icon[..., 3] = 255
    icon[0, 0, 3] = 0  # a transparent corner still OWNS its pixel (set_label_pixel is called for it)
    tb = TileBuilder(canvas=(10, 20, 30))
    dl = tb.build()
    tl = labels.TileLabels()
    # icon centred at (50.5, 40): start = (50.5 - 3) as i32 = 47, (40 - 2) = 38; text overwrites part of it
    tl.label(icon=(0, 50.5, 40.0), text=_text((0, 255, 0), 49, 39, 51, 41))
    tl.label(text=_text((255, 255, 255), 47, 38, 48, 39))  # hits only the transparent icon corner -> fails
    tl.label(icon=(0, 52.0, 41.0))                          # icon overlapping the first one -> fails
    tl.label(icon=(7, 80.0, 80.0), text=_text((1, 1, 1), 80, 80, 82, 82))  # unknown image id: no icon, text drawn
    tl.label()                                               # neither icon nor text: succeeds, draws nothing
    ll = tl.build()
    out, status = oracle.render_job(dl, 0, images=[icon], labels=ll, want_status=True)
    assert status.tolist() == [1, 0, 0, 1, 1]
    assert out[38, 47, :3].tolist() == [10, 20, 30]      # transparent corner: canvas unchanged
    assert out[38, 48, :3].tolist() == [200, 0, 0]        # icon
    assert out[40, 50, :3].tolist() == [0, 255, 0]        # text replaced the icon pixel
    assert out[41, 52, :3].tolist() == [200, 0, 0]
    assert out[42, 52, :3].tolist() == [10, 20, 30]
    assert out[81, 81, :3].tolist() == [1, 1, 1]


def test_label_blend_is_over_the_area_canvas(oracle):
    """labels blend with `new + (1 - a) * old` over the canvas AFTER the areas (drawer.rs:104-125)."""
    tb = TileBuilder(canvas=(0, 0, 0))
    tb.fill([[(0, 0), (100, 0), (100, 100), (0, 100), (0, 0)]], (100, 100, 100), 1.0)
    dl = tb.build()
    tl = labels.TileLabels()
    tl.label(text=_text((255, 255, 255), 10.5, 10.5, 12.5, 12.5))
    out = oracle.render_job(dl, 0, labels=tl.build())
    # coverage 0.25 at the corner: 0.25*1 + 0.75*(100/255) -> u8 truncation
    assert out[10, 10, 0] == int(255.0 * (0.25 * (255 / 255.0) + (1.0 - 0.25) * (1.0 * (100 / 255.0))))
    assert out[11, 11, 0] == 255 and out[9, 9, 0] == 100


def test_label_lists_concat_and_subset_roundtrip():
    ll = labels.make_labels(5, labels_per_tile=4, n_images=2, image_sizes=[(16, 16), (12, 20)], seed=3)
    assert ll.n_jobs == 5 and len(ll.labels) == 20
    parts = [ll.subset([i]) for i in range(5)]
    back = labels.concat_labels(parts)
    assert np.array_equal(back.labels, ll.labels) and np.array_equal(back.segs, ll.segs)
    assert np.array_equal(back.job_label_off, ll.job_label_off)
    assert ll.algorithmic_bytes() == 40 * 20 + 32 * len(ll.segs)
