"""Tiles WITHOUT ops inside big batches, on dirty device memory.

The reference renders a tile with no areas as plain canvas (`Drawer::draw_to_pixels` with an empty entity list,
src/draw/drawer.rs:60-131) after resetting every pixel and every pending entry (`TilePixels::reset`,
src/draw/tile_pixels.rs:89-105).  Here the per-(tile, sub-tile) list headers of a batch of more than 64 tiles come from
`k_sublist`; round 4 shipped an early return there that skipped a tile with zero ops, `k_raster` then read list headers out
of a recycled, never-zeroed buffer and faulted the GPU (Weak #1).  These tests put empty tiles first / last /
everywhere into batches of 65, 70 and 300 tiles, through every entry point that takes a batch, after a dense scene has
left its lists in the buffer cache — and the whole GPU suite runs with OSMT_POISON_ALLOC=1 (tests/conftest.py), so a
header that nobody wrote reads 0xA5A5A5A5 instead of the zero page a fresh process happens to get."""
import numpy as np
import pytest

from osm_renderer_amd import abi, display_list, labels, shard, synth
from osm_renderer_amd.display_list import TileBuilder
from osm_renderer_amd.lib import load as load_lib
from osm_renderer_amd.renderer import Context
from tests._parity import assert_parity

pytestmark = pytest.mark.gpu
CAPS = [abi.CAP_NONE, abi.CAP_BUTT, abi.CAP_ROUND, abi.CAP_SQUARE]


def _tile(rnd, n_ops, scale=1, canvas=(241, 238, 232)):
    tb = TileBuilder(scale=scale, canvas=canvas)
    W = 256 * scale
    for i in range(n_ops):
        p = rnd.integers(0, W, size=2)
        if i % 3 == 0:
            r = int(rnd.integers(4, 50)) * scale
            tb.fill([(p[0] - r, p[1] - r), (p[0] + r, p[1] - r // 2), (p[0] + r // 3, p[1] + r), (p[0] - r, p[1] - r)],
                    tuple(rnd.integers(0, 256, size=3)), float(rnd.choice([1.0, 0.5])))
        elif i % 3 == 1:
            pts = (p + np.cumsum(rnd.integers(-40, 41, size=(3, 2)) * scale, axis=0)).tolist()
            tb.stroke(pts, float(rnd.choice([1.0, 2.5, 6.0])) * scale, tuple(rnd.integers(0, 256, size=3)), float(rnd.choice([1.0, 0.7])),
                      cap=CAPS[i % 4])
        else:
            q = p + rnd.integers(-80, 81, size=2) * scale
            tb.stroke([p.tolist(), q.tolist()], 3.0 * scale, tuple(rnd.integers(0, 256, size=3)), 0.8, dashes=[7.0, 4.0], cap=CAPS[(i + 1) % 4])
    return tb.build()


def _empty_set(n, pattern):
    return {
        "first": {0},
        "last": {n - 1},
        "first_and_last": {0, n - 1},
        "all": set(range(n)),
        "every_third": set(range(0, n, 3)),
        "all_but_one": set(range(n)) - {n // 2},
    }[pattern]


def _batch(n, pattern, scale=1, seed=5, ops=(1, 12)):
    rnd = np.random.default_rng(seed + n)
    empty = _empty_set(n, pattern)
    tiles = []
    for i in range(n):
        canvas = (int(40 + 3 * i) % 256, 200, int(255 - i) % 256) if i % 2 else (241, 238, 232)
        tiles.append(_tile(rnd, 0 if i in empty else int(rnd.integers(ops[0], ops[1])), scale, canvas))
    return display_list.concat(tiles), empty


def _dirty(gpu_ctx, scale=1, n=72):
    """Leaves a dense scene's lists, headers, arenas and framebuffer in the context's buffer cache."""
    dl = synth.config2(n, scale=scale)
    scene = gpu_ctx.upload(dl)
    gpu_ctx.render(scene).cpu()
    scene.free()


def _canvas_only(got, dl, idx):
    want = np.array(list(dl.jobs["canvas_rgb"][idx]) + [255], dtype=np.uint8)
    assert (got[idx] == want).all(), f"tile {idx} has no ops and must be plain canvas {want.tolist()}"


def test_poison_mode_is_switched_on_for_the_gpu_suite():
    assert load_lib().osmt_debug_poison_enabled() == 1, "tests/conftest.py sets OSMT_POISON_ALLOC=1 before the library is loaded"


@pytest.mark.parametrize("n", [65, 70, 300])
@pytest.mark.parametrize("pattern", ["first", "last", "all", "every_third", "all_but_one"])
def test_empty_tiles_in_big_batches_through_the_scene_entry(gpu_ctx, oracle, n, pattern):
    _dirty(gpu_ctx)
    dl, empty = _batch(n, pattern)
    got = assert_parity(gpu_ctx, oracle, dl, f64_jobs=[0, n - 1], msg=f"{n} tiles, empty {pattern}")
    for i in sorted(empty)[:4]:
        _canvas_only(got, dl, i)


@pytest.mark.parametrize("n,pattern", [(17, "first_and_last"), (65, "first_and_last"), (66, "all"), (70, "every_third")])
def test_empty_tiles_at_2x(gpu_ctx, oracle, n, pattern):
    _dirty(gpu_ctx, scale=2, n=20)
    dl, empty = _batch(n, pattern, scale=2, seed=9, ops=(1, 8))
    got = assert_parity(gpu_ctx, oracle, dl, f64_jobs=[0], msg=f"@2x {n} tiles, empty {pattern}")
    for i in sorted(empty)[:2]:
        _canvas_only(got, dl, i)


@pytest.mark.parametrize("n,pattern", [(64, "every_third"), (65, "first"), (130, "all")])
def test_small_and_big_batches_agree_on_empty_tiles(gpu_ctx, oracle, n, pattern):
    """The same tiles as a batch of at most 64 (k_raster<FOLD>, no list kernel) and inside a bigger one (k_sublist)."""
    _dirty(gpu_ctx)
    dl, _ = _batch(n, pattern, seed=21)
    whole = assert_parity(gpu_ctx, oracle, dl, f64_jobs=[], msg="whole")
    for first in range(0, n, 50):
        idx = list(range(first, min(first + 50, n)))
        scene = gpu_ctx.upload(dl.subset(idx))
        part = gpu_ctx.render(scene).cpu().numpy()
        scene.free()
        assert np.array_equal(part, whole[idx]), f"tiles {first}.. differ between a small and a big batch"


def test_empty_tiles_with_labels(gpu_ctx, oracle):
    """Labels on tiles without areas (draw_labels runs whatever draw_areas drew, drawer.rs:107-125)."""
    _dirty(gpu_ctx)
    n = 70
    dl, empty = _batch(n, "every_third", seed=33)
    ll = labels.make_labels(n, labels_per_tile=5, scale=1, seed=3)
    scene = gpu_ctx.upload(dl, ll)
    got = gpu_ctx.render(scene).cpu().numpy()
    st = scene.label_status()
    scene.free()
    want, wst = oracle.render_batch(dl, threads=8, labels=ll, want_status=True)
    assert np.array_equal(st, wst)
    assert np.array_equal(got, want)
    assert any((got[i] != got[i][0, 0]).any() for i in empty), "some empty tile should carry a label"


def test_empty_tiles_through_the_host_buffer_entries(gpu_ctx, oracle):
    """osmt_render_batch (pageable), osmt_render_batch_rgb into pinned memory (the chunk pipeline: >= 256 tiles) and the PNG
    entry, 300 tiles, every third one empty and the last 40 all empty (a whole chunk tail of plain canvas)."""
    _dirty(gpu_ctx)
    n = 300
    dl, empty = _batch(n, "every_third", seed=41)
    tail, _ = _batch(40, "all", seed=42)
    dl = display_list.concat([dl, tail])
    n = dl.n_jobs
    want = oracle.render_batch(dl, threads=8)
    got = gpu_ctx.render_batch_host(dl)
    assert np.array_equal(got, want)
    pin = gpu_ctx.host_alloc((n, 256 * 256 * 3))
    try:
        pin[:] = 0x5A
        rgb = gpu_ctx.render_batch_rgb(dl, out=pin)
        assert np.array_equal(rgb.reshape(n, 256, 256, 3), want[..., :3])
    finally:
        gpu_ctx.host_free(pin)
    from tests.test_gpu_png_device import _decode

    files = gpu_ctx.render_batch_png(dl)
    assert len(files) == n
    for i in (0, 1, 3, 150, 299, n - 1):
        assert np.array_equal(_decode(files[i]), want[i, ..., :3]), f"PNG of tile {i}"


def test_empty_tiles_through_render_batch_multi(gpu_ctx, oracle):
    """Round-robin shards (tile i -> context i mod 3): with every third tile empty ONE shard is nothing but empty tiles."""
    _dirty(gpu_ctx)
    n = 210  # 70 tiles per shard: every shard is a big batch
    dl, empty = _batch(n, "every_third", seed=55)
    want = oracle.render_batch(dl, threads=8)
    ctxs = [gpu_ctx, Context(0), Context(0)]
    try:
        got, cnt = shard.render_batch_multi(ctxs, dl)
        assert cnt == n
        assert np.array_equal(got, want)
        rgb, cnt = shard.render_batch_multi(ctxs, dl, rgb=True)
        assert cnt == n and np.array_equal(rgb.reshape(n, 256, 256, 3), want[..., :3])
    finally:
        ctxs[1].close()
        ctxs[2].close()


def test_empty_tiles_through_the_worker_entry(gpu_ctx, oracle):
    """A worker request of more than 64 tiles is rendered directly, smaller ones are gathered: both with empty tiles."""
    _dirty(gpu_ctx)
    w = gpu_ctx.worker()
    try:
        for n, pattern in ((3, "first"), (1, "all"), (64, "every_third"), (70, "first_and_last")):
            dl, _ = _batch(n, pattern, seed=77)
            got = w.render(dl)
            want = oracle.render_batch(dl, threads=8)
            assert np.array_equal(got.reshape(n, 256, 256, 3), want[..., :3]), (n, pattern)
    finally:
        w.close()
