"""OSMT_COORD_NODE_REF (data-layout part): rings reference a shared node table like the
reference's geodata file (reader.rs:291-336); results must equal the per-point lat/lon form."""
import numpy as np
import pytest

from osm_renderer_amd import abi, synth


def _lists(n=4):
    dl = synth.make_tiles(synth.config_tiles(n), n_poly=10, n_line=8)
    # make neighbouring tiles share geometry the way get_entities_in_tile_with_neighbors duplicates it
    dl.coords[dl.jobs[1]["pt_off"] : dl.jobs[1]["pt_off"] + 40] = dl.coords[dl.jobs[0]["pt_off"] : dl.jobs[0]["pt_off"] + 40]
    return dl, dl.with_node_refs()


def test_node_ref_form_is_smaller_and_equivalent_in_the_oracle(oracle):
    dl, nr = _lists()
    assert nr.coord_kind == abi.COORD_NODE_REF and len(nr.nodes) < len(dl.coords)  # closed rings repeat their first node
    assert np.array_equal(nr.nodes[nr.coords], dl.coords)
    # bytes shrink once nodes are shared >= 1.33x (fill + casing + stroke passes of a way, neighbouring tiles): 4 B/ref + 16 B/node
    assert nr.algorithmic_bytes() - dl.algorithmic_bytes() == 16 * len(nr.nodes) + 4 * len(nr.coords) - 16 * len(dl.coords)
    assert np.array_equal(oracle.render_batch(nr, threads=4), oracle.render_batch(dl, threads=4))
    for j in range(dl.n_jobs):
        assert np.array_equal(oracle.job_points(nr, j), oracle.job_points(dl, j))


@pytest.mark.gpu
def test_gpu_node_refs_match_per_point_latlon(gpu_ctx, oracle):
    from osm_renderer_amd.lib import OsmtError

    dl, nr = _lists(6)
    sa, sb = gpu_ctx.upload(dl), gpu_ctx.upload(nr)
    a, b = gpu_ctx.render(sa).cpu().numpy(), gpu_ctx.render(sb).cpu().numpy()
    assert np.array_equal(gpu_ctx.read_points(sa), gpu_ctx.read_points(sb))
    assert np.array_equal(a, b) and np.array_equal(b, oracle.render_batch(nr, threads=6))
    assert np.array_equal(gpu_ctx.render_batch_host(nr), b)
    sa.free()
    sb.free()
    bad = dl.with_node_refs()
    bad.coords[3] = len(bad.nodes)  # dangling reference
    with pytest.raises(OsmtError):
        gpu_ctx.upload(bad)
