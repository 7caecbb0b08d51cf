"""Synthetic workload generator and display-list plumbing, checked on CPU via the oracle."""
import numpy as np

from osm_renderer_amd import abi, display_list, synth


def test_splitmix64_reference_values():
    # published SplitMix64 outputs for seed 0
    z = synth.splitmix64_draws([0], 3)[0]
    assert [int(v) for v in z] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_config2_shape_and_bytes():
    dl = synth.config2(4)
    assert dl.n_jobs == 4 and len(dl.ops) == 4 * 90 and len(dl.coords) == 4 * 690
    assert int((dl.ops["kind"] == abi.OP_FILL_COLOR).sum()) == 200
    assert int((dl.ops["kind"] == abi.OP_STROKE).sum()) == 160
    # ~279 104 B per tile (exact value depends on the dashed-stroke count)
    per_tile = dl.algorithmic_bytes() / dl.n_jobs
    assert 16 * 690 + 64 * 90 + 4 * 256 * 256 <= per_tile <= 16 * 690 + 64 * 90 + 8 * 80 + 4 * 256 * 256
    assert dl.jobs["x"].tolist() == [19000, 19001, 19002, 19003] and set(dl.jobs["y"]) == {10000}
    dl2 = synth.config2(4)
    assert np.array_equal(dl.coords, dl2.coords) and dl.ops.tobytes() == dl2.ops.tobytes()  # deterministic


def test_tiles_are_independent_of_batch_composition(oracle):
    a = synth.config2(3)
    b = synth.make_tiles(synth.config_tiles(3)[1:2])
    assert np.array_equal(oracle.render_job(a, 1), oracle.render_job(b, 0))


def test_subset_and_concat_roundtrip(oracle):
    dl = synth.config2(3)
    parts = [dl.subset([i]) for i in range(3)]
    re = display_list.concat(parts)
    assert np.array_equal(re.coords, dl.coords) and re.ops.tobytes() == dl.ops.tobytes()
    assert np.array_equal(oracle.render_batch(re), oracle.render_batch(dl))
    rev = dl.subset([2, 0])
    want = oracle.render_batch(dl)
    got = oracle.render_batch(rev)
    assert np.array_equal(got[0], want[2]) and np.array_equal(got[1], want[0])


def test_tilebuilder_matches_direct_calls(oracle):
    tb = display_list.TileBuilder(scale=1, canvas=(241, 238, 232))
    ring = [(10, 10), (60, 14), (50, 70), (12, 40), (10, 10)]
    line = [(5, 5), (100, 80), (200, 30)]
    tb.fill(ring, (200, 10, 10), 0.5)
    tb.nop()
    tb.stroke(line, 5.0, (0, 0, 200), 0.8, dashes=[6, 3], cap=abi.CAP_ROUND)
    dl = tb.build()
    got = oracle.render_job(dl, 0)
    p = oracle.Pixels(1)
    p.reset((241, 238, 232))
    p.fill_contour(oracle.ring_to_pairs(ring), (200, 10, 10), 0.5)
    p.bump_generation()
    p.bump_generation()
    p.draw_lines(oracle.ring_to_pairs(line), 5.0, (0, 0, 200), 0.8, dashes=[6, 3], cap=abi.CAP_ROUND)
    p.bump_generation()
    p.blend_unfinished_pixels()
    assert np.array_equal(got[..., :3], p.to_rgb()) and np.all(got[..., 3] == 255)


def test_latlon_roundtrip_is_close_to_pixels(oracle):
    dl_ll = synth.make_tiles(synth.config_tiles(1), coord_kind=abi.COORD_LATLON_F64)
    dl_px = synth.make_tiles(synth.config_tiles(1), coord_kind=abi.COORD_POINT_I32)
    pts = oracle.job_points(dl_ll, 0)
    # inverse Mercator then the reference projection lands within one pixel of the generator's pixel
    assert np.abs(pts - dl_px.coords).max() <= 1


def test_config5_dense_shape():
    dl = synth.config5(1)
    assert len(dl.ops) == 9000 and len(dl.coords) == 5000 * 9 + 4000 * 6
    assert int(dl.jobs["zoom"][0]) == 17
