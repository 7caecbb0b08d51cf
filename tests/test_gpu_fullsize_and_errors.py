"""BASELINE-size batches through size-independent properties, and the error behaviour of the ABI."""
import ctypes as C

import numpy as np
import pytest

from osm_renderer_amd import abi, synth
from osm_renderer_amd.lib import OsmtError

pytestmark = pytest.mark.gpu


def test_config2_full_batch_properties(gpu_ctx, oracle):
    """1024 tiles (configs[1]): determinism, batch-composition independence, permutation
    equivariance, and oracle parity on a random sample of tiles."""
    import torch

    dl = synth.config2(1024)
    scene = gpu_ctx.upload(dl)
    a = gpu_ctx.render(scene)
    b = gpu_ctx.render(scene)
    assert torch.equal(a, b)
    a = a.cpu().numpy()
    rnd = np.random.default_rng(0)
    pick = sorted(rnd.choice(1024, size=24, replace=False).tolist())
    sub = dl.subset(pick)
    want = oracle.render_batch(sub, threads=8)
    np.testing.assert_array_equal(a[pick], want)  # same tiles inside the big batch == oracle
    got_sub = gpu_ctx.render_batch_host(sub)  # ... and rendered alone
    np.testing.assert_array_equal(got_sub, want)
    perm = rnd.permutation(64)
    permuted = gpu_ctx.render_batch_host(dl.subset(perm.tolist()))
    np.testing.assert_array_equal(permuted, a[perm])
    # checksum of per-tile checksums is order independent
    cs = a.reshape(1024, -1).astype(np.uint64).sum(axis=1)
    assert int(cs[perm].sum()) == int(permuted.reshape(64, -1).astype(np.uint64).sum())
    assert np.all(a[..., 3] == 255)
    scene.free()


def test_config3_full_batch_sample(gpu_ctx, oracle):
    dl = synth.config3(256)
    scene = gpu_ctx.upload(dl)
    a = gpu_ctx.render(scene).cpu().numpy()
    pick = [0, 17, 101, 255]
    np.testing.assert_array_equal(a[pick], oracle.render_batch(dl.subset(pick), threads=4))
    scene.free()


def test_error_codes(gpu_ctx):
    dl = synth.config2(1)
    bad = synth.config2(1)
    bad.ops["ring_off"][3] = 10**6
    with pytest.raises(OsmtError) as e:
        gpu_ctx.upload(bad)
    assert e.value.code == abi.INVALID_ARG and "ring range" in str(e.value)
    bad = synth.config2(1)
    bad.scale = 9
    with pytest.raises(OsmtError) as e:
        gpu_ctx.upload(bad)
    assert e.value.code == abi.INVALID_ARG
    bad = synth.config2(1)
    bad.ops["opacity"][0] = float("nan")
    with pytest.raises(OsmtError):
        gpu_ctx.upload(bad)
    bad = synth.config2(1)
    k = int(np.nonzero(bad.ops["kind"] == abi.OP_STROKE)[0][0])
    bad.ops["has_dashes"][k] = 1
    bad.ops["n_dashes"][k] = 0  # Some([]) panics in the reference
    with pytest.raises(OsmtError) as e:
        gpu_ctx.upload(bad)
    assert "empty dash list" in str(e.value)
    # out-of-range geometry is NOT an error
    far = synth.make_tiles(synth.config_tiles(1), coord_kind=abi.COORD_POINT_I32)
    far.coords[:] = far.coords + 100000
    out = gpu_ctx.render_batch_host(far)
    assert np.all(out[0, :, :, :3] == np.array(synth.CANVAS_OSMOSNIMKI, dtype=np.uint8))
    huge = synth.make_tiles(synth.config_tiles(1), coord_kind=abi.COORD_POINT_I32)
    huge.coords[0, 0] = 1 << 30
    with pytest.raises(OsmtError) as e:
        gpu_ctx.upload(huge)
    assert e.value.code == abi.UNSUPPORTED
    # stride too small
    from osm_renderer_amd.lib import load

    scene = gpu_ctx.upload(dl)
    rc = load().osmt_render_scene(gpu_ctx._h, scene._h, C.c_void_p(1), 16, None)
    assert rc == abi.INVALID_ARG
    scene.free()


def test_empty_batch(gpu_ctx):
    dl = synth.config2(1).subset([0])
    dl.jobs = dl.jobs[:0]
    with pytest.raises(OsmtError) as e:  # ops no job covers would still reach the per-op pre-pass: rejected
        gpu_ctx.render_batch_host(dl)
    assert e.value.code == abi.INVALID_ARG and "not covered by any job" in str(e.value)
    dl.ops = dl.ops[:0]
    out = gpu_ctx.render_batch_host(dl)
    assert out.shape[0] == 0


def test_pinned_output_takes_the_overlapped_chunk_pipeline(gpu_ctx, oracle):
    """osmt_render_batch[_labels] into osmt_host_alloc memory with >= 256 tiles: kernels of one 128-tile chunk
    overlap the D2H copy of the previous one; pixels equal the one-shot path and the oracle."""
    from osm_renderer_amd import labels, synth

    n = 300  # 128 + 128 + 44: a ragged last chunk
    dl = synth.make_tiles(synth.config_tiles(n), n_poly=6, n_line=6)
    pool = labels.make_labels(10, labels_per_tile=5, seed=21)
    ll = labels.concat_labels([pool.subset([i % 10]) for i in range(n)])
    pin = gpu_ctx.host_alloc((n, dl.dim, dl.dim, 4))
    try:
        pin[:] = 7
        a = gpu_ctx.render_batch_host(dl, out=pin).copy()
        b = gpu_ctx.render_batch_host(dl)  # pageable: single copy
        assert np.array_equal(a, b)
        pin[:] = 9
        c = gpu_ctx.render_batch_host(dl, labels=ll, out=pin).copy()
        d = gpu_ctx.render_batch_host(dl, labels=ll)
        assert np.array_equal(c, d) and not np.array_equal(a, c)
        idx = [0, 127, 128, 255, 256, 299]
        want = oracle.render_batch(dl.subset(idx), threads=6, labels=ll.subset(idx))
        assert np.array_equal(c[idx], want)
    finally:
        gpu_ctx.host_free(pin)


def test_rgb8_output_is_the_rgba8_output_without_alpha(gpu_ctx):
    """osmt_render_batch_rgb (the reference's RgbTriples layout, tile_pixels.rs:46,164-181): one-shot and chunk-pipelined
    paths, tight and padded strides, with and without labels, @2x"""
    from osm_renderer_amd import abi, labels, synth
    from osm_renderer_amd.lib import OsmtError

    n = 300
    dl = synth.make_tiles(synth.config_tiles(n), n_poly=6, n_line=6)
    pool = labels.make_labels(10, labels_per_tile=5, seed=22)
    ll = labels.concat_labels([pool.subset([i % 10]) for i in range(n)])
    tight = dl.dim * dl.dim * 3
    want = gpu_ctx.render_batch_host(dl)[..., :3].reshape(n, tight)
    want_l = gpu_ctx.render_batch_host(dl, labels=ll)[..., :3].reshape(n, tight)
    assert np.array_equal(gpu_ctx.render_batch_rgb(dl), want)  # pageable: one copy
    assert np.array_equal(gpu_ctx.render_batch_rgb(dl, labels=ll), want_l)
    stride = tight + 64  # padded tiles: the 2D copy
    out = np.full((n, stride), 0xAB, dtype=np.uint8)
    gpu_ctx.render_batch_rgb(dl, out=out, stride=stride)
    assert np.array_equal(out[:, :tight], want) and (out[:, tight:] == 0xAB).all()
    pin = gpu_ctx.host_alloc((n, tight))
    try:
        pin[:] = 5
        assert np.array_equal(gpu_ctx.render_batch_rgb(dl, labels=ll, out=pin), want_l)  # pinned, >= 256 tiles: chunks overlap
    finally:
        gpu_ctx.host_free(pin)
    d2 = synth.make_tiles(synth.config_tiles(3), scale=2, n_poly=10, n_line=10)
    assert np.array_equal(gpu_ctx.render_batch_rgb(d2), gpu_ctx.render_batch_host(d2)[..., :3].reshape(3, -1))
    with pytest.raises(OsmtError) as e:
        gpu_ctx.render_batch_rgb(dl, out=np.zeros((n, tight - 1), dtype=np.uint8), stride=tight - 1)
    assert e.value.code == abi.INVALID_ARG


def test_config5_dense_city_at_stated_density(gpu_ctx, oracle):
    """BASELINE configs[4] as stated: 5000 polygons + 4000 polylines (20000 segments) per z=17 tile.  Two tiles
    bit-exact (RGBA8 and f64 canvas) against the oracle, determinism and permutation equivariance on 16 tiles, and
    one tile whose dense multipolygon puts > 16 crossings on a row (the overflow path of the fill)."""
    import torch

    from osm_renderer_amd.display_list import TileBuilder, concat
    from tests._parity import assert_parity

    dl = synth.config5(16)
    assert int(dl.jobs["n_ops"][0]) == 9000 and int(dl.jobs["n_pts"][0]) == 5000 * 9 + 4000 * 6
    scene = gpu_ctx.upload(dl)
    a = gpu_ctx.render(scene)
    b = gpu_ctx.render(scene)
    assert torch.equal(a, b)
    a = a.cpu().numpy()
    scene.free()
    pick = [3, 12]
    assert_parity(gpu_ctx, oracle, dl.subset(pick), msg="config5")  # RGBA8 + f64 canvas, rendered alone
    want = oracle.render_batch(dl.subset(pick), threads=2)
    np.testing.assert_array_equal(a[pick], want)  # ... and inside the 16-tile batch
    perm = np.random.default_rng(5).permutation(16)
    permuted = gpu_ctx.render_batch_host(dl.subset(perm.tolist()))
    np.testing.assert_array_equal(permuted, a[perm])
    # a config-5 tile followed by a comb multipolygon (160 crossings per row inside one 32-px column band)
    tb = TileBuilder(zoom=17, x=79001, y=40001, scale=1, canvas=synth.CANVAS_OSMOSNIMKI)
    rnd = np.random.default_rng(17)
    for _ in range(300):
        c = rnd.integers(0, 256, size=2)
        r = rnd.integers(2, 12, size=8)
        ang = 2 * np.pi * np.arange(8) / 8
        ring = [(int(c[0] + r[k] * np.cos(ang[k])), int(c[1] + r[k] * np.sin(ang[k]))) for k in range(8)]
        tb.fill(ring + ring[:1], tuple(int(v) for v in rnd.integers(0, 256, size=3)), float(rnd.choice([1.0, 0.7, 0.5])))
    dense = []
    for i in range(80):
        x = 96 + (i * 31) % 32
        dense += [(x, 20 + (i % 7)), (x + (i % 3) - 1, 230 - (i % 11))]
    dense.append(dense[0])
    hole = [(100, 60), (120, 60), (120, 180), (100, 180), (100, 60)]
    tb.fill([dense, hole], (200, 10, 10), 0.6)
    for _ in range(300):
        p = rnd.integers(-8, 264, size=2).astype(np.int64)
        pts = [tuple(int(v) for v in p)]
        for _ in range(5):
            p = p + rnd.integers(-12, 13, size=2)
            pts.append(tuple(int(v) for v in p))
        tb.stroke(pts, float(rnd.choice([0.5, 1.0, 2.0, 4.0])), tuple(int(v) for v in rnd.integers(0, 256, size=3)), 0.6,
                  dashes=[3.0, 3.0] if rnd.random() < 0.25 else None, cap=int(rnd.choice([abi.CAP_NONE, abi.CAP_ROUND, abi.CAP_SQUARE])))
    assert_parity(gpu_ctx, oracle, tb.build(), msg="dense tile with a comb multipolygon")


def test_worker_threads_with_their_own_scenes_do_not_wait_for_each_other(gpu_ctx):
    """osmt_scene_upload / osmt_scene_set_labels / osmt_scene_read_label_status / osmt_scene_read_points /
    osmt_scene_free of one worker's scene wait for THAT scene's launches only (per-stream events, private copy
    streams), never for the device: while the main thread keeps ~0.3 s of renders queued on its stream, another
    thread's calls on its own scenes return in milliseconds.  (Kernels of two streams may still share a hardware
    queue — that is the GPU's scheduling, not a synchronisation inside the library — so the worker launches none.)"""
    import threading
    import time

    import torch

    from osm_renderer_amd import labels

    big = gpu_ctx.upload(synth.config2(1024))
    big_out = torch.empty((1024, 256, 256, 4), dtype=torch.uint8, device=gpu_ctx.device)
    s_big = torch.cuda.Stream(device=gpu_ctx.device)
    gpu_ctx.render(big, big_out, stream=s_big)
    small_dl = synth.config2(2)
    small_ll = labels.make_labels(2, labels_per_tile=6, seed=9)
    small_out = torch.empty((2, 256, 256, 4), dtype=torch.uint8, device=gpu_ctx.device)
    s = torch.cuda.Stream(device=gpu_ctx.device)
    sc = gpu_ctx.upload(small_dl)
    # one complete round trip before the clock starts: first-use costs of the runtime (hipMalloc of the label buffers,
    # kernel attributes) are paid once at start-up by a server too, and the scene now has finished launches to wait for
    sc.set_labels(small_ll)
    gpu_ctx.render(sc, small_out, stream=s)
    want_status = sc.label_status()
    torch.cuda.synchronize()
    lat = []

    def worker():
        for _ in range(5):
            t0 = time.perf_counter()
            sc.set_labels(small_ll)            # waits for sc's own last launches (finished), copies on a private stream
            st = sc.label_status()             # verdicts of the last render of THIS scene
            pts = gpu_ctx.read_points(sc)
            tmp = gpu_ctx.upload(small_dl)     # a fresh scene: upload, sizing, free
            tmp.free()
            lat.append(round(time.perf_counter() - t0, 5))
            assert len(st) == len(want_status) and pts.shape[0] == len(small_dl.coords)

    n_queue = 150  # ~1.9 ms each: ~0.3 s of queued work on s_big
    t0 = time.perf_counter()
    for _ in range(n_queue):
        gpu_ctx.render(big, big_out, stream=s_big)
    th = threading.Thread(target=worker)
    th.start()
    th.join()
    t_worker_done = time.perf_counter() - t0
    s_big.synchronize()
    t_queue_done = time.perf_counter() - t0
    sc.free()
    big.free()
    assert len(lat) == 5
    # with a device-wide synchronisation inside any of the calls the worker could not finish before the big queue did
    assert t_worker_done < 0.5 * t_queue_done, (t_worker_done, t_queue_done, lat)


def test_big_upload_checks_coordinates_on_its_helper_threads(gpu_ctx, oracle):
    """Uploads of >= 65536 coordinates run the O(n_pts) coordinate scan on helper threads beside the copies and build the
    point -> job table on the device: a bad coordinate anywhere is still refused (with the lowest offending index), before
    any kernel has seen it; a good batch renders like the oracle (k_ptjob's table = the host's)."""
    dl = synth.make_tiles(synth.config_tiles(110), n_poly=50, n_line=40)  # 75 900 points
    assert len(dl.coords) >= 65536
    good = gpu_ctx.render_batch_host(dl)
    pick = [0, 57, 109]
    np.testing.assert_array_equal(good[pick], oracle.render_batch(dl.subset(pick), threads=3))
    for idx in (len(dl.coords) - 1, 40000, 3):
        bad = synth.make_tiles(synth.config_tiles(110), n_poly=50, n_line=40)
        bad.coords[idx, 0] = 89.5  # beyond the Web-Mercator square
        bad.coords[len(bad.coords) - 2, 1] = float("nan")
        with pytest.raises(OsmtError) as e:
            gpu_ctx.upload(bad)
        assert e.value.code == abi.UNSUPPORTED
        first = min(idx, len(bad.coords) - 2)
        assert f"point {first}:" in str(e.value), str(e.value)
        with pytest.raises(OsmtError):
            gpu_ctx.render_batch_host(bad)
    again = gpu_ctx.render_batch_host(dl)  # the context is fine afterwards
    np.testing.assert_array_equal(again, good)


def test_guessed_arenas_that_turn_out_too_small_are_rendered_again(oracle):
    """Big host-buffer uploads take their pre-pass arenas from the densities recent uploads measured (+ 25 %) instead of a
    counting run and a round trip.  A batch that needs far more than the guess — same op counts, ten times the geometry —
    must come back right all the same: the kernels report the overflow in the scene's error word and the call renders
    again with exact sizing."""
    import numpy as np

    from osm_renderer_amd import synth
    from osm_renderer_amd.renderer import Context

    ctx = Context(0)  # its own density history
    try:
        small = synth.make_tiles(synth.config_tiles(96, x0=20000, y0=11000), radius=(1.0, 3.0), step=3.0)
        big = synth.make_tiles(synth.config_tiles(96, x0=20000, y0=11000))
        ctx.render_batch_host(small)  # measures: small densities
        got_small = ctx.render_batch_host(small)  # guessed, fits
        got_big = ctx.render_batch_host(big)  # guessed from the small batches: misses, rendered again
        got_big2 = ctx.render_batch_host(big)  # measured again after the miss
        scene = ctx.upload(big)  # public scenes are always sized exactly
        want_big = ctx.render(scene).cpu().numpy()
        scene.free()
        assert np.array_equal(got_big, want_big) and np.array_equal(got_big2, want_big)
        pick = [0, 17, 95]
        assert np.array_equal(want_big[pick], oracle.render_batch(big.subset(pick), threads=3))
        assert np.array_equal(got_small[pick], oracle.render_batch(small.subset(pick), threads=3))
    finally:
        ctx.close()
    ctx = Context(0)  # the PNG pipeline guesses too, and recovers from a miss the same way (in its second half)
    try:
        from tests.test_gpu_png_device import _decode

        ctx.render_batch_png(small)  # measures
        files = ctx.render_batch_png(small)  # guessed, fits
        files_big = ctx.render_batch_png(big)  # guessed from the small batches: misses, the job runs again
        assert np.array_equal(_decode(files_big[17]), want_big[17][..., :3])
        assert np.array_equal(_decode(files_big[95]), want_big[95][..., :3])
        assert np.array_equal(_decode(files[17]), got_small[17][..., :3])
    finally:
        ctx.close()


def test_split_upload_refuses_bad_batches_and_renders_good_ones(gpu_ctx, oracle):
    """Uploads of more than 4 MB of host arrays are SPLIT (round 5): a helper thread copies the caller's arrays while the calling
    thread validates the jobs and builds the index tables.  A bad op, a bad ring, a bad coordinate anywhere must still be refused
    before any kernel has seen the batch — with the helper joined and its buffer given back — and good batches render like the
    oracle, through the host-buffer entry (stream-ordered upload) and as a scene."""
    n = 420
    dl = synth.config2(n)  # ~11 MB of host arrays
    assert len(dl.coords) * 16 + len(dl.ops) * 64 > (4 << 20)
    pick = [0, 1, 211, n - 1]
    want = oracle.render_batch(dl.subset(pick), threads=4)
    np.testing.assert_array_equal(gpu_ctx.render_batch_host(dl)[pick], want)
    for kind in ("ring_off", "n_rings", "coord", "dashes", "kind"):
        bad = synth.config2(n)
        k = len(bad.ops) - 3
        if kind == "ring_off":
            bad.ops["ring_off"][k] = 10**8
            code = abi.INVALID_ARG
        elif kind == "n_rings":
            bad.ops["n_rings"][5] = 10**6
            code = abi.INVALID_ARG
        elif kind == "coord":
            bad.coords[len(bad.coords) - 7, 0] = 89.9
            code = abi.UNSUPPORTED
        elif kind == "dashes":
            strokes = np.nonzero(bad.ops["kind"] == abi.OP_STROKE)[0]
            bad.ops["has_dashes"][strokes[-1]] = 1
            bad.ops["n_dashes"][strokes[-1]] = 0
            code = abi.INVALID_ARG
        else:
            bad.ops["kind"][k] = 77
            code = abi.INVALID_ARG
        for call in (gpu_ctx.render_batch_host, gpu_ctx.upload):
            with pytest.raises(OsmtError) as e:
                call(bad)
            assert e.value.code == code, (kind, e.value)
    # nothing leaked or hung: the same good batch again, as a scene and through the RGB entry
    scene = gpu_ctx.upload(dl)
    got = gpu_ctx.render(scene).cpu().numpy()
    scene.free()
    np.testing.assert_array_equal(got[pick], want)
    np.testing.assert_array_equal(gpu_ctx.render_batch_rgb(dl).reshape(n, 256, 256, 3)[pick], want[..., :3])
