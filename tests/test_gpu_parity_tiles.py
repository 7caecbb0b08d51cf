"""GPU-vs-oracle parity on whole synthetic tiles (BASELINE.json configs 2/3), through the C ABI.

Bar: RGBA8 framebuffer bit-exact AND the un-quantised f64 canvas bit-exact."""
import numpy as np
import pytest

from osm_renderer_amd import abi, synth

pytestmark = pytest.mark.gpu


def _compare(gpu_ctx, oracle, dl, check_f64=True):
    scene = gpu_ctx.upload(dl)
    got = gpu_ctx.render(scene).cpu().numpy()
    want = oracle.render_batch(dl, threads=8)
    # projection parity first (integer points)
    if dl.coord_kind == abi.COORD_LATLON_F64:
        pts = gpu_ctx.read_points(scene)
        for j in range(dl.n_jobs):
            job = dl.jobs[j]
            ref = oracle.job_points(dl, j)
            np.testing.assert_array_equal(pts[job["pt_off"] : job["pt_off"] + job["n_pts"]], ref)
    diff = np.nonzero((got != want).any(axis=-1))
    assert len(diff[0]) == 0, f"{len(diff[0])} pixels differ, first at tile/y/x = {[int(d[0]) for d in diff]}: gpu={got[diff][0]} oracle={want[diff][0]}"
    if check_f64:
        f64 = gpu_ctx.render_f64(scene).cpu().numpy()
        for j in range(min(dl.n_jobs, 4)):
            _, ref = oracle.render_job(dl, j, want_f64=True)
            assert np.array_equal(f64[j].view(np.uint64), ref.view(np.uint64)), f"f64 canvas differs on tile {j}"
    scene.free()


def test_config2_tiles(gpu_ctx, oracle):
    _compare(gpu_ctx, oracle, synth.config2(24))


def test_config3_tiles_2x(gpu_ctx, oracle):
    _compare(gpu_ctx, oracle, synth.config3(8))


def test_config2_host_points(gpu_ctx, oracle):
    dl = synth.make_tiles(synth.config_tiles(8, x0=19100), coord_kind=abi.COORD_POINT_I32)
    _compare(gpu_ctx, oracle, dl)


def test_render_batch_host_api(gpu_ctx, oracle):
    dl = synth.config2(3)
    got = gpu_ctx.render_batch_host(dl)
    want = oracle.render_batch(dl)
    np.testing.assert_array_equal(got, want)
