"""Shared GPU-vs-oracle comparison for display lists (through the C ABI)."""
import numpy as np


def assert_parity(gpu_ctx, oracle, dl, images=(), f64_jobs=None, msg=""):
    """RGBA8 bit-exact on every tile, f64 canvas bit-exact on `f64_jobs` (default: all, <= 6)."""
    scene = gpu_ctx.upload(dl)
    got = gpu_ctx.render(scene).cpu().numpy()
    want = oracle.render_batch(dl, images=images, threads=min(8, max(1, dl.n_jobs)))
    bad = np.nonzero((got != want).any(axis=-1))
    assert len(bad[0]) == 0, (
        f"{msg}: {len(bad[0])} pixels differ; first (tile,y,x)={tuple(int(b[0]) for b in bad)} "
        f"gpu={got[bad][0].tolist()} oracle={want[bad][0].tolist()}"
    )
    jobs = range(min(dl.n_jobs, 6)) if f64_jobs is None else f64_jobs
    if len(jobs):
        f64 = gpu_ctx.render_f64(scene).cpu().numpy()
        for j in jobs:
            _, ref = oracle.render_job(dl, j, images=images, want_f64=True)
            same = f64[j].view(np.uint64) == ref.view(np.uint64)
            assert same.all(), f"{msg}: f64 canvas differs on tile {j} at {np.argwhere(~same)[0].tolist()}"
    scene.free()
    return got
