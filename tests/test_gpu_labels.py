"""Label pass on the GPU (k_label_segprep / k_label_cover / k_label_resolve + k_raster<LABELS>) against the
oracle, through the C ABI: RGBA8 bit-exact and label_generation_statuses identical."""
import numpy as np
import pytest

from osm_renderer_amd import labels, synth
from osm_renderer_amd.display_list import TileBuilder, concat
from tests.test_labels_oracle import _square, _text

pytestmark = pytest.mark.gpu


def _icons(rng, sizes):
    out = []
    for h, w in sizes:
        img = rng.integers(0, 256, size=(h, w, 4)).astype(np.uint8)
        img[: h // 3, :, 3] = 255
        img[0, 0, 3] = 0
        out.append(img)
    return out


def _check(gpu_ctx, oracle, dl, ll, images=(), msg=""):
    for img in images:
        pass
    scene = gpu_ctx.upload(dl, ll)
    got = gpu_ctx.render(scene).cpu().numpy()
    st = scene.label_status()
    want, wst = oracle.render_batch(dl, images=images, threads=min(8, max(1, dl.n_jobs)), labels=ll, want_status=True)
    assert np.array_equal(st, wst), f"{msg}: label statuses differ at {np.nonzero(st != wst)[0][:8].tolist()}"
    bad = np.nonzero((got != want).any(axis=-1))
    assert len(bad[0]) == 0, (
        f"{msg}: {len(bad[0])} pixels differ; first (tile,y,x)={tuple(int(b[0]) for b in bad)} "
        f"gpu={got[bad][0].tolist()} oracle={want[bad][0].tolist()}"
    )
    # the host-buffer entry point gives the same pixels
    host = gpu_ctx.render_batch_host(dl.subset([0]), ll.subset([0]))
    assert np.array_equal(host[0], want[0])
    scene.free()
    return got, st


@pytest.fixture(scope="module")
def icon_ctx(gpu_ctx):
    rng = np.random.default_rng(11)
    imgs = _icons(rng, [(16, 16), (12, 20), (5, 7)])
    ids = [gpu_ctx.register_image(i) for i in imgs]
    # image ids are positions in the context-wide registry: the oracle gets the same list
    all_imgs = [None] * (max(ids) + 1)
    for i, img in zip(ids, imgs):
        all_imgs[i] = img
    for k in range(len(all_imgs)):
        if all_imgs[k] is None:
            all_imgs[k] = np.zeros((1, 1, 4), dtype=np.uint8)
    return ids, all_imgs


def test_collision_rules_and_icons(gpu_ctx, oracle, icon_ctx):
    ids, imgs = icon_ctx
    tb = TileBuilder(canvas=(255, 250, 240))
    tb.fill([[(0, 0), (120, 0), (120, 120), (0, 120), (0, 0)]], (100, 140, 180), 0.7)
    dl = tb.build()
    tl = labels.TileLabels()
    tl.label(text=_text((255, 0, 0), 10, 10, 20, 20))
    tl.label(text=_text((0, 255, 0), 15, 15, 30, 30))
    tl.label(text=_text((0, 0, 255), 25.25, 25.5, 40.75, 40.125))
    tl.label(text=_text((9, 9, 9), -200, -200, -190, -190))
    tl.label(text=_text((7, 7, 7), -195, -195, -100, -100))
    tl.label(text=_text((5, 5, 5), 900, 900, 950, 950))
    tl.label(text=_text((1, 2, 3), 250.5, 100, 300, 120))
    tl.label(icon=(ids[0], 60.5, 60.0), text=_text((0, 255, 0), 58, 66, 70, 75))
    tl.label(icon=(ids[1], 64.0, 70.0))
    tl.label(icon=(ids[2], 200.0, 10.0), text=_text((20, 30, 40), 199, 9, 202, 12))
    tl.label(icon=(9999, 80.0, 80.0), text=_text((1, 1, 1), 80, 80, 82, 82))
    tl.label()
    tl.label(text=((3, 3, 3), np.zeros((0, 4))))
    tl.label(icon=(ids[0], -250.0, 300.0))  # straddles the labels_bb edge
    tl.label(icon=(ids[0], 255.0, 255.0))   # straddles the tile corner
    got, st = _check(gpu_ctx, oracle, dl, tl.build(), imgs, "rules")
    assert st[:7].tolist() == [1, 0, 1, 1, 0, 1, 1]


def test_rasterizer_edge_cases(gpu_ctx, oracle):
    dl = TileBuilder(canvas=(0, 0, 0)).build()
    tl = labels.TileLabels()
    w = (255, 255, 255)
    tl.label(text=(w, np.array([(10, 10, 10, 30), (10, 30, 10.5, 30), (10.5, 30, 10.5, 10), (10.5, 10, 10, 10)], dtype=float)))  # thin sliver
    tl.label(text=(w, np.array([(40, 10, 40, 12), (40, 12, 90, 12.001), (90, 12.001, 40, 10)], dtype=float)))  # near-horizontal edge
    tl.label(text=(w, np.array(_square(100, 100, 103, 103) + _square(101, 101, 104, 104), dtype=float)))  # overlapping contours: clamp
    tl.label(text=(w, np.array(_square(-5000, 140, -4990, 150) + _square(120, 140, 130, 150), dtype=float)))  # glyph far left in the same stripes
    tl.label(text=(w, np.array(_square(150, -300, 160, 900), dtype=float)))  # rows clipped to labels_bb
    tl.label(text=(w, np.array([(200.3, 200.7, 200.3, 200.9), (200.3, 200.9, 200.6, 200.9), (200.6, 200.9, 200.3, 200.7)], dtype=float)))  # inside one pixel
    tl.label(text=(w, np.array([(c, d, a, b) for (a, b, c, d) in _square(220, 220, 230, 230)][::-1], dtype=float)))  # negative orientation: no pixels
    tl.label(text=(w, np.array([(230, 50, 240, 50), (240, 50, 240, 50)], dtype=float)))  # horizontal + degenerate
    _check(gpu_ctx, oracle, dl, tl.build(), (), "edges")


@pytest.mark.parametrize("scale", [1, 2])
def test_synthetic_labels_over_area_tiles(gpu_ctx, oracle, icon_ctx, scale):
    ids, imgs = icon_ctx
    n = 6
    dl = synth.make_tiles(synth.config_tiles(n), n_poly=12, n_line=10, scale=scale)
    ll = labels.make_labels(n, labels_per_tile=20 * scale * scale, scale=scale, seed=5 + scale, n_images=3,
                            image_sizes=[i.shape[:2] for i in [imgs[k] for k in ids]])
    # make_labels numbers images 0..n_images-1: map to the registered ids
    ll.labels["image_id"] = np.array(ids, dtype=np.uint32)[ll.labels["image_id"] % 3]
    got, st = _check(gpu_ctx, oracle, dl, ll, imgs, f"synthetic scale {scale}")
    assert 0 < st.sum() < len(st)  # both outcomes occur


def test_labels_can_be_detached_and_rerendered(gpu_ctx, oracle):
    dl = synth.make_tiles(synth.config_tiles(2), n_poly=5, n_line=5)
    ll = labels.make_labels(2, labels_per_tile=8, seed=9)
    scene = gpu_ctx.upload(dl, ll)
    a = gpu_ctx.render(scene).cpu().numpy()
    b = gpu_ctx.render(scene).cpu().numpy()  # the planes are rebuilt every render
    assert np.array_equal(a, b)
    scene.set_labels(None)
    plain = gpu_ctx.render(scene).cpu().numpy()
    assert np.array_equal(plain, oracle.render_batch(dl, threads=2))
    assert np.array_equal(a, oracle.render_batch(dl, threads=2, labels=ll))
    scene.free()


def test_label_validation_errors(gpu_ctx):
    from osm_renderer_amd.lib import OsmtError

    dl = TileBuilder().build()
    tl = labels.TileLabels()
    tl.label(text=((0, 0, 0), np.array([(0, 0, float("nan"), 5)], dtype=float)))
    with pytest.raises(OsmtError):
        gpu_ctx.upload(dl, tl.build())
    tl = labels.TileLabels()
    tl.label(text=((0, 0, 0), np.array([(0, 0, 3e6, 5)], dtype=float)))
    with pytest.raises(OsmtError):
        gpu_ctx.upload(dl, tl.build())
    two = concat([dl, dl])
    with pytest.raises(AssertionError):
        gpu_ctx.upload(two, labels.TileLabels().build())


def test_reference_icons_as_fill_patterns_and_label_icons(gpu_ctx, oracle):
    """the reference's own icon files (tests/golden/ref_icons.json) as Filler::Image patterns
    (fill.rs:36-40, opacity ignored, x mod w / y mod h) and as label icons, GPU vs oracle."""
    import json
    import os

    fix = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_icons.json")))
    names = [k for k in fix if not k.startswith("_")]
    imgs = {n: np.array(fix[n]["rgba"], dtype=np.uint8) for n in names}
    ids = {n: gpu_ctx.register_image(imgs[n]) for n in names}
    all_imgs = [np.zeros((1, 1, 4), dtype=np.uint8) for _ in range(max(ids.values()) + 1)]
    for n in names:
        all_imgs[ids[n]] = imgs[n]
    tb = TileBuilder(canvas=(0xF1, 0xEE, 0xE8))
    tb.fill_image([(5, 5), (250, 20), (200, 120), (10, 100), (5, 5)], ids["forest.png"])
    tb.fill_image([(20, 110), (240, 130), (230, 250), (30, 240), (20, 110)], ids["scrub.png"], opacity=0.25)
    tb.fill_image([(100, 60), (180, 70), (170, 200), (90, 180), (100, 60)], ids["military_red_hz2.png"])  # translucent hatch over both
    tb.fill_image([(-40, -40), (60, -10), (40, 70), (-30, 50), (-40, -40)], ids["grave_yard.png"])
    tb.fill([[(150, 150), (250, 150), (250, 250), (150, 250), (150, 150)]], (20, 30, 200), 0.4)
    dl = tb.build()
    tl = labels.TileLabels()
    tl.label(icon=(ids["cafe.p.16.png"], 60.0, 60.0), text=_text((0x73, 0x4A, 0x08), 50, 70, 72, 78))
    tl.label(icon=(ids["station.png"], 200.5, 40.5))
    tl.label(icon=(ids["station.png"], 204.0, 44.0))  # overlaps the previous icon: fails
    tl.label(icon=(ids["orchard.png"], 128.0, 128.0))
    got, st = _check(gpu_ctx, oracle, dl, tl.build(), all_imgs, "reference icons")
    assert st.tolist() == [1, 1, 0, 1]
    assert len(np.unique(got[0].reshape(-1, 4), axis=0)) > 50


def test_label_status_before_the_first_render_is_all_zero(gpu_ctx, oracle):
    """osmt_scene_read_label_status between osmt_scene_set_labels and the next render: nothing has been placed yet, so
    every status is 0 and there is no error — not whatever the recycled label buffer held (the poisoned allocator of
    tests/conftest.py turned that into "label coverage window overflow (internal error 0xA5A5A5A5)" in round 5)."""
    dl = synth.config2(3)
    ll = labels.make_labels(3, labels_per_tile=7, seed=5)
    scene = gpu_ctx.upload(dl)
    gpu_ctx.render(scene).cpu()
    scene.set_labels(ll)
    st0 = scene.label_status()
    assert st0.shape == (len(ll.labels),) and not st0.any()
    got = gpu_ctx.render(scene).cpu().numpy()
    want, wst = oracle.render_batch(dl, threads=3, labels=ll, want_status=True)
    assert np.array_equal(scene.label_status(), wst) and np.array_equal(got, want)
    scene.set_labels(ll)  # attached again: the old verdicts are gone with the old buffers
    assert not scene.label_status().any()
    scene.free()
