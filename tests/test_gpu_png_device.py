"""PNG files written by the GPU (k_png_encode): valid files whose DECODED pixels equal the
framebuffer (the only thing the reference's tests pin, tests/test_rendering.rs:15-23), byte-identical to the CPU
model of the encoder (tests/_png_model.py)."""
import io
import zlib

import numpy as np
import pytest

from osm_renderer_amd import labels, synth
from tests import _png_model

pytestmark = pytest.mark.gpu


def _decode(png):
    from PIL import Image

    return np.array(Image.open(io.BytesIO(png)).convert("RGB"))


def test_device_png_matches_model_and_decodes(gpu_ctx):
    import torch

    rng = np.random.default_rng(3)
    imgs = np.zeros((6, 256, 256, 4), dtype=np.uint8)
    imgs[0] = rng.integers(0, 256, size=(256, 256, 4))                      # incompressible: every literal is 8/9 bits
    imgs[1, :, :] = (241, 238, 232, 255)                                    # flat: runs of 768 zeros, 258-byte matches
    imgs[2, :, :, :3] = (np.arange(256)[None, :, None] // 3).astype(np.uint8)  # short horizontal runs, all run lengths
    imgs[3, ::2] = 255                                                      # alternating rows
    imgs[4, :, :, 0] = rng.integers(0, 2, size=(256, 256)) * 200             # run breaks at random places
    imgs[5, 100:140, 90:200, :3] = rng.integers(140, 256, size=(40, 110, 3))  # literals >= 144 (9-bit codes)
    imgs[..., 3] = 255
    slots, lens = gpu_ctx.encode_png_device(torch.from_numpy(imgs).cuda())
    slots, lens = slots.cpu().numpy(), lens.cpu().numpy()
    for i in range(len(imgs)):
        png = slots[i, : lens[i]].tobytes()
        assert np.array_equal(_decode(png), imgs[i, :, :, :3]), f"image {i}: decoded pixels differ"
        assert png == _png_model.encode(imgs[i]), f"image {i}: bytes differ from the model"
    assert lens[1] < 3000 and lens[0] > 196608  # flat tile ~2 KB; noise is slightly larger than raw RGB


@pytest.mark.parametrize("scale", [1, 2])
def test_render_batch_png_end_to_end(gpu_ctx, oracle, scale):
    n = 5
    dl = synth.make_tiles(synth.config_tiles(n), n_poly=20, n_line=15, scale=scale)
    ll = labels.make_labels(n, labels_per_tile=6, scale=scale, seed=8)
    files = gpu_ctx.render_batch_png(dl, ll)
    want = oracle.render_batch(dl, threads=n, labels=ll)
    raw = want[0][..., :3].nbytes
    for i, png in enumerate(files):
        assert np.array_equal(_decode(png), want[i][..., :3])
        assert len(png) < raw / 2
    # the zlib stream inside is standard: python's zlib inflates it to H * (3W + 1) filtered bytes
    png = files[0]
    idat_len = int.from_bytes(png[33:37], "big")
    assert len(zlib.decompress(png[41 : 41 + idat_len])) == dl.dim * (3 * dl.dim + 1)


@pytest.mark.parametrize("shape", [(70, 128), (33, 768), (5, 1024), (64, 64)])
def test_generic_widths_take_the_two_pass_kernel(gpu_ctx, shape):
    """widths other than 256 / 512 (or heights that do not split into 4 bands) run the LDS two-pass kernel"""
    import torch

    H, W = shape
    rng = np.random.default_rng(H * W)
    imgs = np.zeros((3, H, W, 4), dtype=np.uint8)
    imgs[0, :, :, :3] = rng.integers(0, 256, size=(H, W, 3))
    imgs[1, :, : W // 2, :3] = (10, 200, 30)
    imgs[2, :, :, :3] = (rng.integers(0, 4, size=(H, W, 1)) * 60).astype(np.uint8)
    imgs[..., 3] = 255
    slots, lens = gpu_ctx.encode_png_device(torch.from_numpy(imgs).cuda())
    slots, lens = slots.cpu().numpy(), lens.cpu().numpy()
    for i in range(3):
        png = slots[i, : lens[i]].tobytes()
        assert np.array_equal(_decode(png), imgs[i, :, :, :3])
        assert png == _png_model.encode(imgs[i])


def test_png_fuzz_against_model(gpu_ctx):
    """random images with random run structure (flat blocks, stripes, noise patches, gradients), 256 and 512 wide"""
    import torch

    rng = np.random.default_rng(2027)
    for W in (256, 512):
        imgs = np.zeros((8, W, W, 4), dtype=np.uint8)
        for im in imgs:
            im[..., :3] = rng.integers(0, 256, size=3)
            for _ in range(int(rng.integers(1, 30))):
                x0, y0 = rng.integers(0, W, size=2)
                w, h = rng.integers(1, W, size=2)
                kind = rng.integers(0, 4)
                sl = (slice(y0, min(W, y0 + h)), slice(x0, min(W, x0 + w)))
                if kind == 0:
                    im[sl][..., :3] = rng.integers(0, 256, size=3)
                elif kind == 1:
                    im[sl][..., :3] = rng.integers(0, 256, size=im[sl][..., :3].shape)
                elif kind == 2:
                    im[sl][..., :3] = (np.arange(im[sl].shape[1])[None, :, None] * int(rng.integers(1, 9))) % 256
                else:
                    im[sl][..., :3] = (np.arange(im[sl].shape[0])[:, None, None] // int(rng.integers(1, 5))) % 256
        imgs[..., 3] = 255
        slots, lens = gpu_ctx.encode_png_device(torch.from_numpy(imgs).cuda())
        slots, lens = slots.cpu().numpy(), lens.cpu().numpy()
        for i in range(len(imgs)):
            png = slots[i, : lens[i]].tobytes()
            assert png == _png_model.encode(imgs[i]), f"W={W} image {i}: bytes differ from the model"
            assert np.array_equal(_decode(png), imgs[i, :, :, :3])


def test_png_begin_end_pipelined_equals_the_one_piece_call(gpu_ctx, oracle):
    """osmt_render_batch_png_begin / _end: two jobs in flight from one thread (batch k + 1 begun before batch k is ended),
    different batches, one of them split into chunks; every file equals the one the one-piece call writes and decodes to
    the oracle's pixels.  A job whose second half gets a too small buffer reports the size and is released."""
    from osm_renderer_amd.lib import OsmtError

    lists = [synth.make_tiles(synth.config_tiles(n, x0=19000 + 11 * k, y0=10020 + k), n_poly=12, n_line=10) for k, n in enumerate((3, 1100, 7, 2))]
    want = [gpu_ctx.render_batch_png(dl) for dl in lists]
    bufs = [np.empty(dl.n_jobs * 96 * 1024, dtype=np.uint8) for dl in lists]
    got = [None] * len(lists)
    prev = gpu_ctx.png_begin(lists[0])
    for k in range(1, len(lists)):
        cur = gpu_ctx.png_begin(lists[k])
        got[k - 1] = gpu_ctx.png_end(prev, bufs[k - 1], as_bytes=True)
        prev = cur
    got[-1] = gpu_ctx.png_end(prev, bufs[-1], as_bytes=True)
    for k, dl in enumerate(lists):
        assert got[k] == want[k], f"batch {k}"
    ref = oracle.render_batch(lists[2], threads=4)
    for i in range(lists[2].n_jobs):
        assert np.array_equal(_decode(got[2][i]), ref[i][..., :3])
    job = gpu_ctx.png_begin(lists[0])
    with pytest.raises(OsmtError) as e:
        gpu_ctx.png_end(job, np.empty(100, dtype=np.uint8))
    assert "out_capacity" in str(e.value)
    # the context is still usable and nothing of the failed job lingers
    assert gpu_ctx.render_batch_png(lists[3]) == want[3]


def test_png_job_outlives_its_context():
    """a begun job that is ended — or just dropped — AFTER the caller destroyed the context (an exception between
    png_begin and png_end with ctx.close() in a finally block) must not hand its buffers and streams back to a context that
    has been torn down: the job holds a reference of its own, the teardown runs when the job lets go."""
    from osm_renderer_amd.renderer import Context

    dl = synth.make_tiles(synth.config_tiles(5, x0=19100, y0=10100), n_poly=10, n_line=8)
    ref_ctx = Context(0)
    want = ref_ctx.render_batch_png(dl)
    ref_ctx.close()
    for how in ("end", "drop"):
        ctx = Context(0)
        job = ctx.png_begin(dl)
        ctx.close()  # osmt_destroy with a job in flight
        if how == "end":  # (png_end only passes the job on: the closed context's handle is not used)
            got = ctx.png_end(job, np.empty(dl.n_jobs * 96 * 1024, dtype=np.uint8), as_bytes=True)
            assert got == want
        else:
            del job  # PngJob.__del__ -> osmt_render_batch_png_end(job, NULL, 0, ..): releases everything
    after = Context(0)  # the device is fine and a new context renders the same files
    assert after.render_batch_png(dl) == want
    after.close()


def test_compacted_files_in_dead_framebuffers_or_in_a_buffer_of_their_own(gpu_ctx):
    """A chunk's files are compacted into that chunk's framebuffers (dead once encoded); with more than two chunks the
    framebuffers are re-used, and the job takes a separate buffer (forced here with the diagnostic OSMT_PNG_CHUNKS, which
    is read once per process: a child).  Same files either way."""
    import hashlib
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dl = synth.make_tiles(synth.config_tiles(9, x0=19040, y0=10033), n_poly=14, n_line=12)
    want = hashlib.sha256(b"".join(gpu_ctx.render_batch_png(dl))).hexdigest()
    child = (
        "import hashlib, sys; sys.path.insert(0, %r)\n"
        "from osm_renderer_amd import synth\n"
        "from osm_renderer_amd.renderer import Context\n"
        "dl = synth.make_tiles(synth.config_tiles(9, x0=19040, y0=10033), n_poly=14, n_line=12)\n"
        "print(hashlib.sha256(b''.join(Context(0).render_batch_png(dl))).hexdigest())\n" % root
    )
    out = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, OSMT_PNG_CHUNKS="4"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == want
