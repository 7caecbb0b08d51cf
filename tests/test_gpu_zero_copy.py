"""Small requests into PINNED caller memory: k_raster writes the pixels straight into the caller's buffer (no device framebuffer, no
copy; `osmt_render_batch_labels_once`, OSMT_ZERO_COPY_TILES = 16 tiles by default).  The reference hands every worker its own
`Vec<(u8, u8, u8)>` (`tile_pixels.rs:164-181`); what arrives in the caller's memory must be the same bytes whichever way they travel:
zero copy (pinned, <= 16 tiles, 4-byte aligned pointer and stride), the copy path (pageable memory, more tiles, an odd pointer) — with
padded strides, labels, both pixel formats and from several threads at once."""
import threading

import numpy as np
import pytest

from osm_renderer_amd import labels, synth

pytestmark = pytest.mark.gpu


def _pinned(gpu_ctx, nbytes):
    return gpu_ctx.host_alloc((nbytes,))


@pytest.mark.parametrize("n", [1, 3, 16, 17])
def test_rgb_into_pinned_memory_with_tight_and_padded_strides(gpu_ctx, oracle, n):
    dl = synth.make_tiles(synth.config_tiles(n, x0=19100), n_poly=12, n_line=10)
    want = oracle.render_batch(dl, threads=min(8, n))[..., :3].reshape(n, -1)
    tight = 256 * 256 * 3
    for stride in (tight, tight + 64):
        buf = _pinned(gpu_ctx, n * stride + 16)
        try:
            buf[:] = 0x5A
            got = gpu_ctx.render_batch_rgb(dl, out=buf[: n * stride].reshape(n, stride), stride=stride)
            assert np.array_equal(got[:, :tight], want), f"{n} tiles, stride {stride}"
            if stride > tight:
                assert (got[:, tight:] == 0x5A).all(), "the padding between tiles is the caller's"
            assert (buf[n * stride :] == 0x5A).all()
            # a pointer INTO a pinned allocation (a 4-byte aligned offset): still zero copy, the device address follows the offset
            if n * stride + 12 <= buf.size:
                inner = buf[12 : 12 + n * stride]
                buf[:] = 0x33
                got3 = gpu_ctx.render_batch_rgb(dl, out=inner.reshape(n, stride), stride=stride)
                assert np.array_equal(got3[:, :tight], want) and (buf[:12] == 0x33).all() and (buf[12 + n * stride :] == 0x33).all()
            # an odd pointer cannot take the zero-copy path (dword stores): the copy path must give the same bytes
            odd = buf[1 : 1 + n * stride]
            if stride == tight and n <= 3:
                got2 = gpu_ctx.render_batch_rgb(dl, out=odd.reshape(n, stride), stride=stride)
                assert np.array_equal(got2, want)
        finally:
            gpu_ctx.host_free(buf)
    # pageable memory: the copy path
    assert np.array_equal(gpu_ctx.render_batch_rgb(dl), want)


def test_rgba_and_labels_into_pinned_memory(gpu_ctx, oracle):
    n = 5
    dl = synth.config2(n)
    ll = labels.make_labels(n, labels_per_tile=8, seed=21)
    want = oracle.render_batch(dl, threads=n, labels=ll)
    pin = gpu_ctx.host_alloc((n, 256, 256, 4))
    try:
        pin[:] = 7
        got = gpu_ctx.render_batch_host(dl, ll, out=pin)
        assert np.array_equal(got, want)
        pin[:] = 9
        assert np.array_equal(gpu_ctx.render_batch_host(dl, out=pin), oracle.render_batch(dl, threads=n))
    finally:
        gpu_ctx.host_free(pin)
    pin3 = gpu_ctx.host_alloc((n, 256 * 256 * 3))
    try:
        got = gpu_ctx.render_batch_rgb(dl, ll, out=pin3)
        assert np.array_equal(got.reshape(n, 256, 256, 3), want[..., :3])
    finally:
        gpu_ctx.host_free(pin3)


def test_pinned_buffers_from_eight_threads(gpu_ctx, oracle):
    """Every thread its own pinned buffer and its own tile, 40 requests each, all at once: nobody's pixels end up in somebody
    else's buffer, and a buffer is complete when its call returns (no copy is in flight behind it)."""
    T = 8
    dls = [synth.make_tiles(synth.config_tiles(1, x0=19200 + 3 * t, y0=10050 + t), n_poly=20, n_line=16) for t in range(T)]
    want = [oracle.render_batch(d)[0, ..., :3].reshape(-1) for d in dls]
    bufs = [gpu_ctx.host_alloc((1, 256 * 256 * 3)) for _ in range(T)]
    bad = []

    def run(t):
        for _ in range(40):
            bufs[t][:] = 0
            got = gpu_ctx.render_batch_rgb(dls[t], out=bufs[t])
            if not np.array_equal(got[0], want[t]):
                bad.append(t)
                return

    th = [threading.Thread(target=run, args=(t,)) for t in range(T)]
    try:
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not bad, f"threads {sorted(set(bad))} read pixels that differ from the oracle"
    finally:
        for b in bufs:
            gpu_ctx.host_free(b)
