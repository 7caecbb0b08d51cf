"""osmt_batch_shard_create (host only): the slicing osmt_render_batch_multi and the one-process-per-GPU harness use.
Shards of a batch are valid display lists of their own, tile i of the batch is tile i // world of shard i % world, and
rendering the shards (here: with the oracle, there is no GPU) reproduces the batch tile for tile."""
import numpy as np
import pytest

from osm_renderer_amd import abi, shard, synth
from osm_renderer_amd.display_list import TileBuilder, concat
from osm_renderer_amd.lib import OsmtError
from tests.test_abi import _validate


def _mixed_batch():
    tiles = [synth.make_tiles(synth.config_tiles(7)[i : i + 1], n_poly=6, n_line=5, coord_kind=abi.COORD_POINT_I32) for i in range(7)]
    tb = TileBuilder()
    tb.fill([[(10, 10), (200, 30), (120, 220), (10, 10)], [(60, 60), (90, 60), (90, 90), (60, 60)]], (200, 10, 10), 0.6)  # two rings
    tb.nop()
    tb.stroke([(5, 5), (250, 250), (250, 5)], 5.0, (1, 2, 3), 0.5, dashes=[6, 3, 2], cap=abi.CAP_ROUND)
    tiles.insert(3, tb.build())
    return concat(tiles)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shards_tile_the_batch(oracle, world):
    dl = _mixed_batch()
    full = oracle.render_batch(dl, threads=2)
    seen = 0
    for rank in range(world):
        sh = shard.shard_display_list(dl, rank, world)
        idx = shard.shard_indices(dl.n_jobs, rank, world)
        assert sh.n_jobs == len(idx)
        assert _validate(sh)[0] == abi.OK  # its op ranges partition its own op pool
        if sh.n_jobs:
            np.testing.assert_array_equal(oracle.render_batch(sh, threads=2), full[idx])
            assert int(sh.jobs["n_ops"].sum()) == len(sh.ops) and int(sh.jobs["n_pts"].sum()) == len(sh.coords)
        seen += sh.n_jobs
    assert seen == dl.n_jobs


def test_shards_of_latlon_and_node_ref_lists(oracle):
    dl = synth.config2(5)
    nr = dl.with_node_refs()
    full = oracle.render_batch(dl, threads=2)
    for lst in (dl, nr):
        for rank in range(2):
            sh = shard.shard_display_list(lst, rank, 2)
            assert sh.coord_kind == lst.coord_kind
            np.testing.assert_array_equal(oracle.render_batch(sh), full[rank::2])


def test_shard_arguments_are_checked():
    dl = synth.config2(2)
    with pytest.raises(OsmtError) as e:
        shard.shard_display_list(dl, 2, 2)
    assert e.value.code == abi.INVALID_ARG
    bad = synth.config2(2)
    bad.jobs["n_ops"][0] -= 1  # an orphan op: the batch itself is refused
    with pytest.raises(OsmtError):
        shard.shard_display_list(bad, 0, 2)
