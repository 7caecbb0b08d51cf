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
