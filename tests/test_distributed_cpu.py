"""The N>1 path on CPU: world_size-2 gloo group; tile i -> rank i mod 2, no data-path
collective, one all-reduce of the tile count.  Each rank renders its shard
with the oracle (there is no GPU here) and the shards must tile the full batch exactly."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_tiles, q):
    import torch.distributed as dist

    from oracle import oracle_py
    from osm_renderer_amd import shard, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tiles = synth.config_tiles(n_tiles)
    idx = shard.shard_indices(n_tiles, rank, world)
    dl = synth.make_tiles(tiles[idx])
    out = oracle_py.render_batch(dl)
    total = shard.reduce_tile_count(dl.n_jobs, dist)
    q.put((rank, idx.tolist(), total, [int(o.astype(np.uint64).sum()) for o in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_covers_the_batch(oracle):
    from osm_renderer_amd import synth

    world, n_tiles = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_tiles, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = oracle.render_batch(synth.config2(n_tiles))
    seen = {}
    for rank, idx, total, sums in res:
        assert total == n_tiles  # all-reduce(sum) of the per-rank counts
        assert idx == list(range(rank, n_tiles, world))
        for i, s in zip(idx, sums):
            seen[i] = s
    assert sorted(seen) == list(range(n_tiles))
    for i in range(n_tiles):
        assert seen[i] == int(full[i].astype(np.uint64).sum())
