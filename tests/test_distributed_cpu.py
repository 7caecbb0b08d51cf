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
    # BASELINE configs[3]: the 10000-tile batch, tile i -> rank i mod world (bench.py --total-tiles 10000 and the
    # config4_strong leg use exactly these two functions); nothing is rendered here, only the partition is checked
    big = synth.config_tiles(10000)
    big_idx = shard.shard_indices(10000, rank, world)
    big_total = shard.reduce_tile_count(len(big_idx), dist)
    big_xy = big[big_idx]
    q.put((rank, idx.tolist(), total, [int(o.astype(np.uint64).sum()) for o in out],
           (big_total, len(big_idx), int(big_idx[0]), int(big_idx[-1]), big_xy[:2].tolist(), int(big_xy[:, 0].astype(np.int64).sum()))))
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
    xsum = 0
    for rank, idx, total, sums, big in res:
        big_total, n_mine, first, last, xy2, xs = big
        assert big_total == 10000 and n_mine == 5000 and first == rank and last == 9998 + rank
        assert xy2 == [[19000 + rank, 10000], [19000 + rank + 2, 10000]]  # x = 19000 + i mod 100, y = 10000 + i / 100
        xsum += xs
        assert total == n_tiles  # all-reduce(sum) of the per-rank counts
        assert idx == list(range(rank, n_tiles, world))
        for i, s in zip(idx, sums):
            seen[i] = s
    assert sorted(seen) == list(range(n_tiles))
    for i in range(n_tiles):
        assert seen[i] == int(full[i].astype(np.uint64).sum())
    assert xsum == int(synth.config_tiles(10000)[:, 0].sum())  # the two shards tile the 10000-tile batch exactly


def test_bench_watchdog_gives_up_on_a_call_that_never_returns():
    """bench.py's guard around osmt_comm_init_rank: a native call that hangs must cost a bounded wait and a
    message, a call that returns or raises comes back as it is."""
    import importlib.util
    import threading
    import time

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.call_with_timeout(lambda: 41 + 1, 5.0, "quick") == (True, 42)
    finished, res = bench.call_with_timeout(lambda: (_ for _ in ()).throw(ValueError("boom")), 5.0, "raises")
    assert finished and isinstance(res, ValueError)
    gate = threading.Event()
    t0 = time.time()
    finished, res = bench.call_with_timeout(gate.wait, 0.3, "stuck")
    assert (finished, res) == (False, None) and time.time() - t0 < 3.0
    gate.set()
