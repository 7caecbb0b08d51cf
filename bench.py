#!/usr/bin/env python
"""bench.py — tiles/sec of the MI355X tile hot path (BASELINE.json), one JSON line on rank 0.

    python bench.py --gpus N --steps K --warmup W [--total-tiles T]

A "step" is one pass of the hot path (project -> per-op pre-pass -> fused fill/stroke/blend raster -> RGBA8
framebuffer) over one batch of synthetic tiles, display lists already resident in HBM, framebuffers written to HBM.

Headline (`value`): BASELINE configs[1] — 1024 z=15 256x256 tiles (50 polygons + 200 stroke segments each) PER GPU;
with N > 1 (launched by torch.distributed.run, one rank per GPU) tile i of the global batch belongs to rank i mod N
(weak scaling), no data-path collective, one RCCL all-reduce of the tile count per step.
`--total-tiles T` switches the headline to STRONG scaling on a fixed global batch (configs[3]: T = 10000, tile i ->
rank i mod N, x = 19000 + i mod 100, y = 10000 + i / 100; N = 1 renders the same T tiles).  The same 10000-tile
strong-scaling batch is ALSO timed in every default run, at every N, as the `config4_strong` object, so a 1/2/4/8-GPU
sweep of the default command carries the north_star's ">= 6x at 8 GPUs on the 10k-tile batch" numbers.

Objects on the line (N = 1 unless said otherwise):
  roofline            k_raster, the dominant kernel: algorithmic bytes / HIP-event kernel time vs 8 TB/s (+ measured copy ceiling)
  roofline_issue      k_raster's real bound: VALU/SALU wave-instructions and VALU-busy cycles from an SQ counter pass of this run
  roofline_composite  the 8-layer @2x composite pass (the kernel the >= 40 % HBM target is defined on)
  hbm_copy_ceiling    a float4 device copy timed in this run (what "HBM speed" is on this box)
  config4_strong      (every N) the 10000-tile z=15 batch sharded round-robin
  raster_2x           configs[2] geometry: the same lists at @2x (512x512)
  config5             configs[4]: dense city, 5000 polygons + 20000 segments per tile, z=17
  sustained           >= 10 s of back-to-back steps at the very end of the run (what the driver's GPU-busy sampler can see)
  end_to_end          PCIe-inclusive: osmt_render_batch into pinned memory, osmt_render_batch_png
  png_encode, label_pass  SURVEY.md 8(f) N3 / N1 on the same tiles
  cpu_baseline        the C++ oracle (restatement of the reference's Rust CPU path, NOT the Rust binary) on the host
                      cores: persistent worker pool (one canvas per worker, allocated before the timer), thread sweep
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_SIMD = 256 * 4       # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9       # max shader clock (same guide)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota_cores():
    """CPU bandwidth the container may use (cgroup v2 cpu.max), in cores; None = unlimited / unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tiles", type=int, default=1024, help="tiles per GPU per step (weak scaling, the default headline)")
    ap.add_argument("--streams", type=int, default=2,
                    help="batches in flight per GPU: S resident copies of the batch (own lists, own workspace, own framebuffers), step i "
                         "runs on stream i mod S — the latency-bound pre-pass of one batch overlaps the raster stage of the previous "
                         "one, as a server that renders batch after batch would have it; 1 = every step waits for the one before")
    ap.add_argument("--total-tiles", type=int, default=0, help="strong scaling: a fixed global batch, tile i -> rank i mod N (configs[3]: 10000)")
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--composite-tiles", type=int, default=64)
    ap.add_argument("--n-poly", type=int, default=50, help="diagnostic: polygons per tile (default = the named config)")
    ap.add_argument("--n-line", type=int, default=40, help="diagnostic: polylines per tile (default = the named config)")
    ap.add_argument("--config4-tiles", type=int, default=10000)
    ap.add_argument("--config5-tiles", type=int, default=256)
    ap.add_argument("--sustained-seconds", type=float, default=10.0)
    ap.add_argument("--label-tiles", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-composite", action="store_true")
    ap.add_argument("--no-labels", action="store_true")
    ap.add_argument("--no-png", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (child processes)")
    ap.add_argument("--no-extra", action="store_true", help="headline + roofline only (no config4/5, 2x, sustained, end-to-end legs)")
    ap.add_argument("--keep-pmc", default="", help="directory to keep the rocpd databases of the counter passes in")
    ap.add_argument("--pmc-child", default="", help="internal: render a few steps of the named workload and exit (run under rocprofv3)")
    return ap.parse_args()


def pmc_child(name):
    """Run under rocprofv3 by tools/pmc_pass.py: 1 warm-up + 3 steps of one workload, nothing else."""
    import torch

    from osm_renderer_amd import abi, synth
    from osm_renderer_amd.renderer import Context

    ctx = Context(0)
    if name.partition(":")[0] in ("composite", "all"):
        planes = synth.composite_planes(64, L=8, dim=512, device=ctx.device)
        cout = torch.empty((64, 512, 512, 4), dtype=torch.uint8, device=ctx.device)
        for _ in range(4):
            ctx.composite(planes, [0xFC / 255.0, 0xF8 / 255.0, 0xE4 / 255.0, 1.0], out=cout)
        torch.cuda.synchronize()
        del planes, cout
        if name == "composite":
            return
    name, _, n_arg = name.partition(":")  # "config5:64" = the workload with 64 tiles
    if name == "config5":
        dl = synth.config5(int(n_arg) if n_arg else 64)
    elif name == "raster_2x":
        dl = synth.config3(int(n_arg) if n_arg else 256)
    else:
        dl = synth.config2(int(n_arg) if n_arg else 1024)
    scene = ctx.upload(dl)
    out = torch.empty((dl.n_jobs, dl.dim, dl.dim, 4), dtype=torch.uint8, device=ctx.device)
    for _ in range(4):
        ctx.render_stages(scene, abi.STAGE_PROJECT | abi.STAGE_OPINFO)
        ctx.render_stages(scene, abi.STAGE_RASTER, out)
    torch.cuda.synchronize()
    scene.free()


def latency_leg(ctx, args, batches=(1, 8, 64), workers=(1, 4, 16), seconds=0.35, entry="batch"):
    """p50 / p99 of osmt_render_batch_rgb (validation + upload + kernels + RGB8 read-back into pinned memory) for small
    batches from several worker threads on one context."""
    import ctypes as C
    import threading

    import numpy as np

    from osm_renderer_amd import synth
    from osm_renderer_amd.lib import load

    L = load()
    out = {"what": "wall clock (us) around one osmt_render_batch_rgb call: B config-2 tiles per call, W worker threads on ONE "
                   "osmt_ctx calling back to back (the reference's server shape: one tile per request per worker, "
                   "src/http_server.rs:50-83,134-181); tiles_per_s = all threads together",
           "cases": {}}
    if entry == "worker":
        out["what"] = ("the same requests through the per-request entry: W threads, each with its own osmt_worker, ONE tile per "
                       "osmt_worker_render call; concurrent requests are gathered into shared launches (group commit, "
                       "OSMT_WORKER_INFLIGHT groups on the device); Python threads — tools/worker_bench.cpp is the native twin")
    handles = []
    if entry == "worker":
        for _ in range(max(workers)):
            h = C.c_void_p()
            if L.osmt_worker_create(ctx._h, C.byref(h)) != 0:
                raise RuntimeError(L.osmt_last_error().decode())
            handles.append(h)

    def call(w, b, ptr, stride):
        if entry == "worker":
            return L.osmt_worker_render(handles[w], b, None, ptr, stride)
        return L.osmt_render_batch_rgb(ctx._h, b, None, ptr, stride)

    max_w, max_b = max(workers), max(batches)
    # the worker loops are Python threads: a thread coming back from the C call has to win the GIL again, and with the
    # default 5 ms switch interval that wait (not the library) was the p99 at 16 workers; 50 us keeps it out of the numbers
    old_switch = sys.getswitchinterval()
    sys.setswitchinterval(5e-5)
    pool = [synth.make_tiles(synth.config_tiles(max_b, x0=19000 + 7 * w, y0=10000 + 3 * w), zoom=15, scale=args.scale, n_poly=args.n_poly,
                             n_line=args.n_line) for w in range(max_w)]
    bufs = [ctx.host_alloc((max_b, pool[0].dim * pool[0].dim * 3)) for _ in range(max_w)]
    try:
        for bsz in batches:
            lists = [p.subset(range(bsz)) for p in pool]
            structs = [dl.as_batch() for dl in lists]
            stride = lists[0].dim * lists[0].dim * 3
            for nw in workers:
                lat = [[] for _ in range(nw)]
                errs = []
                start = threading.Barrier(nw + 1)

                def run(w):
                    ptr = bufs[w].ctypes.data_as(C.POINTER(C.c_uint8))
                    b = C.byref(structs[w])
                    try:
                        for _ in range(3):
                            if call(w, b, ptr, stride) != 0:
                                raise RuntimeError(L.osmt_last_error().decode())
                        start.wait()
                        t_end = time.perf_counter() + seconds
                        while True:
                            t0 = time.perf_counter()
                            rc = call(w, b, ptr, stride)
                            t1 = time.perf_counter()
                            if rc != 0:
                                raise RuntimeError(L.osmt_last_error().decode())
                            lat[w].append(t1 - t0)
                            if t1 >= t_end:
                                break
                    except Exception as e:  # noqa: BLE001
                        errs.append(repr(e))
                        try:
                            start.abort()
                        except Exception:  # noqa: BLE001
                            pass

                th = [threading.Thread(target=run, args=(w,)) for w in range(nw)]
                for t in th:
                    t.start()
                try:
                    start.wait()
                except threading.BrokenBarrierError:
                    pass
                t0 = time.perf_counter()
                for t in th:
                    t.join()
                wall = time.perf_counter() - t0
                if errs:
                    out["cases"][f"batch{bsz}_workers{nw}"] = {"error": errs[0]}
                    continue
                allv = np.sort(np.concatenate([np.asarray(v) for v in lat])) * 1e6
                out["cases"][f"batch{bsz}_workers{nw}" if entry == "batch" else f"workers{nw}"] = {
                    "calls": int(allv.size), "p50_us": float(allv[allv.size // 2]), "p99_us": float(allv[min(allv.size - 1, int(allv.size * 0.99))]),
                    "mean_us": float(allv.mean()), "tiles_per_s": float(allv.size * bsz / wall),
                }
    finally:
        sys.setswitchinterval(old_switch)
        for b in bufs:
            ctx.host_free(b)
        for h in handles:
            L.osmt_worker_destroy(h)
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line under torch.distributed.run with one rank
    per GPU on a free local port, pass rank 0's JSON line through, return the launcher's exit code."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # every rank's stdout / stderr is kept (and still shown): a rank that dies or hangs must be readable afterwards
    log_dir = os.path.join("gpurun_out" if os.path.isdir("gpurun_out") else ".", "bench_rank_logs")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--log-dir", log_dir, "--tee", "3", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def call_with_timeout(fn, seconds, what):
    """fn() on a helper thread; returns (finished, result or exception).  A native call that never returns — a rank stuck inside
    ncclCommInitRank because a peer died on its way there — must not hang the whole job without a word: after `seconds` the
    caller gets (False, None), says so on stderr and goes on without the thing (the helper thread is a daemon and is left behind)."""
    import threading

    box = {}

    def run():
        try:
            box["r"] = fn()
        except BaseException as e:  # noqa: BLE001
            box["e"] = e

    th = threading.Thread(target=run, daemon=True, name=what)
    th.start()
    th.join(seconds)
    if th.is_alive():
        print(f"bench.py: {what} did not return within {seconds:.0f} s — giving up on it (the thread is left behind)", file=sys.stderr, flush=True)
        return False, None
    if "e" in box:
        return True, box["e"]
    return True, box.get("r")


def main():
    args = parse_args()
    if args.pmc_child:
        pmc_child(args.pmc_child)
        return

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = torch.cuda.device_count()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    launched = "WORLD_SIZE" in os.environ
    pinned = any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    if n_dev < args.gpus and not (launched and pinned):
        # never a silent N = 1 run under an N-GPU label.  Not under a launcher that pins one device per rank
        # (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES, common under SLURM): every rank of a correctly provisioned N-GPU
        # job then sees ONE device — there the check is the all-reduced rank count below (rccl_nranks_seen == N)
        sys.exit(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible: refusing to run")
    if launched and n_dev < 1:
        sys.exit("bench.py: no GPU visible to this rank: refusing to run")
    if not launched and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, torch.distributed.run on
        # 127.0.0.1), the way the reference deals tiles to its own workers (src/http_server.rs:50-83,105-108)
        sys.exit(self_launch(args.gpus))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(--nproc-per-node {args.gpus}) or drop the launcher and let bench.py start its own ranks")
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if local_rank >= n_dev:
            local_rank = 0  # one pinned device per rank: it is device 0 of this process
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from osm_renderer_amd import abi, shard, synth
    from osm_renderer_amd.renderer import Context

    ctx = Context(local_rank)
    dev = ctx.device
    solo = rank == 0 and world == 1
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    # The path's only collective: the sum of the per-rank tile counts.  Preferred: the library's own RCCL communicator
    # (osmt_comm_init_rank + osmt_allreduce_tile_count; the unique id travels through torch.distributed, which the
    # harness needs anyway for its barriers).  Any failure falls back to torch.distributed.all_reduce — also RCCL.
    collective = "none (1 GPU)"
    native_comm = False
    comm_hung = False  # the library's communicator never came back: leave through os._exit at the end (its thread is stuck in RCCL)
    nranks_seen = 1  # what an all-reduce(sum) of 1 over the path's collective returns
    if dist is not None:
        collective = "torch.distributed.all_reduce (RCCL)"
        # step 1, no collective of the library yet: can EVERY rank load RCCL through the library?  (a rank that cannot
        # must not leave the others waiting inside ncclCommInitRank)
        try:
            my_uid = shard.comm_unique_id()
            can = 1
        except Exception as e:  # noqa: BLE001
            print(f"rank {rank}: RCCL not loadable through libosmtile ({e})", file=sys.stderr)
            my_uid, can = None, 0
        flag = torch.tensor([can], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            try:
                uid = torch.zeros(abi.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
                if rank == 0:
                    uid.copy_(torch.from_numpy(my_uid))
                dist.broadcast(uid, src=0)
                uid_host = uid.cpu().numpy()

                def native_init():
                    shard.comm_init_rank(ctx, uid_host, rank, world)
                    good = shard.allreduce_tile_count(ctx, 1) == world
                    if good:
                        shard.allreduce_tile_count_enqueue(ctx, 1)
                        good = shard.allreduce_tile_count_result(ctx) == world
                    return good

                # osmt_comm_init_rank has never met a second process on hardware.  If it (or its first all-reduce)
                # hangs, this rank says so after OSMT_BENCH_COMM_TIMEOUT seconds (120) and the job goes on over torch.distributed's
                # communicator — the MIN below turns the native one off on every rank
                finished, res = call_with_timeout(native_init, float(os.environ.get("OSMT_BENCH_COMM_TIMEOUT", "120")),
                                                  f"rank {rank}: osmt_comm_init_rank / first ncclAllReduce of the library's communicator")
                if not finished:
                    comm_hung = True
                    ok = 0
                elif isinstance(res, BaseException):
                    raise res
                else:
                    ok = 1 if res else 0
            except Exception as e:  # noqa: BLE001
                print(f"rank {rank}: native RCCL communicator unavailable ({e}); using torch.distributed", file=sys.stderr)
                ok = 0
            okt = torch.tensor([ok], device=dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            native_comm = bool(okt.item())
        if native_comm:
            nranks_seen = shard.allreduce_tile_count(ctx, 1)
        else:
            one = torch.ones(1, dtype=torch.int64, device=dev)
            dist.all_reduce(one)
            nranks_seen = int(one.item())
        if nranks_seen != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but the all-reduce of 1 over the job's ranks returned {nranks_seen}: refusing to report")
        if native_comm:
            collective = "osmt_allreduce_tile_count_enqueue (library-owned RCCL communicator, ncclAllReduce of one uint64 on the render stream)"

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_sharded(global_tiles_xy, zoom, scale, n_poly, n_line, steps, warmup, maker=None, slots=1):
        """Tile i of the global batch -> rank i mod N.  W untimed + K timed steps bracketed by barrier + synchronize on
        both sides, MAX over ranks; returns the whole-job figures and this rank's objects.  slots > 1: that many resident
        copies of the batch, step i on copy / stream i mod slots (every step still runs project -> pre-pass -> raster over
        its own lists into its own framebuffers; nothing is shared or cached between steps)."""
        mine = global_tiles_xy[shard.shard_indices(len(global_tiles_xy), rank, world)]
        dl = maker(mine) if maker else synth.make_tiles(mine, zoom=zoom, scale=scale, n_poly=n_poly, n_line=n_line)
        scenes = [ctx.upload(dl) for _ in range(slots)]
        outs = [torch.empty((dl.n_jobs, dl.dim, dl.dim, 4), dtype=torch.uint8, device=dev) for _ in range(slots)]
        lanes = [torch.cuda.current_stream()] if slots == 1 else [torch.cuda.Stream() for _ in range(slots)]
        scene, out = scenes[0], outs[0]
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        turn = [0]

        def step(i=None):
            k = turn[0] % slots
            turn[0] += 1
            with torch.cuda.stream(lanes[k]):
                ctx.render_stages(scenes[k], abi.STAGE_PROJECT | abi.STAGE_OPINFO)
                if i is not None:
                    ev[i][0].record()
                ctx.render_stages(scenes[k], abi.STAGE_RASTER, outs[k])
                if i is not None:
                    ev[i][1].record()
                if dist is not None:
                    if native_comm:  # queued behind the raster stage on the same stream, no host synchronisation
                        shard.allreduce_tile_count_enqueue(ctx, dl.n_jobs)
                    else:
                        count.fill_(dl.n_jobs)
                        dist.all_reduce(count)  # RCCL sum of tile counts: the path's only collective

        for _ in range(warmup):
            step()
        sync_all()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        sync_all()
        elapsed = time.perf_counter() - t0
        total = dl.n_jobs
        if dist is not None:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            total = shard.allreduce_tile_count_result(ctx) if native_comm else int(count.item())
            assert total == len(global_tiles_xy), (total, len(global_tiles_xy))
            # every rank rendered exactly ITS share of the round-robin deal (tile i -> rank i mod N), not just the right sum
            mine_n = torch.tensor([dl.n_jobs], dtype=torch.int64, device=dev)
            all_n = [torch.zeros_like(mine_n) for _ in range(world)]
            dist.all_gather(all_n, mine_n)
            want_n = [len(shard.shard_indices(len(global_tiles_xy), r, world)) for r in range(world)]
            got_n = [int(t.item()) for t in all_n]
            if got_n != want_n:
                sys.exit(f"bench.py: tiles per rank {got_n} differ from the round-robin shards {want_n}: refusing to report")
        raster_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        for extra in scenes[1:]:
            extra.free()
        if slots > 1:  # the copies are gone: later callers of step() run on the first one
            scenes[1:], outs[1:], lanes[1:] = [], [], []
            slots = 1
        return {"elapsed": elapsed, "total_tiles": total, "raster_ms": raster_ms, "dl": dl, "scene": scene, "out": out, "step": step}

    # ---- headline ----------------------------------------------------------------------------------
    strong = args.total_tiles > 0
    n_global = args.total_tiles if strong else args.tiles * world
    head = run_sharded(synth.config_tiles(n_global), 15, args.scale, args.n_poly, args.n_line, args.steps, args.warmup)
    dl, scene, out = head["dl"], head["scene"], head["out"]
    alg_bytes = dl.algorithmic_bytes()  # 16*N_pts + 64*N_ops + 8*N_dashes + 4*W*H per tile
    raster_s = head["raster_ms"] / 1e3  # the kernel's own duration: always from the one-batch-at-a-time run (nothing overlaps it there)
    sequential = {"ms_per_step": head["elapsed"] / args.steps * 1e3, "tiles_per_s": head["total_tiles"] * args.steps / head["elapsed"]}
    if args.streams > 1:
        piped = run_sharded(synth.config_tiles(n_global), 15, args.scale, args.n_poly, args.n_line, args.steps, args.warmup, slots=args.streams)
        piped["scene"].free()
        head = dict(head, elapsed=piped["elapsed"], total_tiles=piped["total_tiles"])
        del piped
    achieved = alg_bytes / raster_s / 1e9
    named = args.n_poly == 50 and args.n_line == 40
    if strong:
        workload = (f"BASELINE.json configs[3]: fixed global batch of {n_global} z=15 {dl.dim}x{dl.dim} tiles (x = 19000 + i mod 100, "
                    f"y = 10000 + i / 100), tile i -> rank i mod {world}, RCCL sum of tile counts per step; synthetic 50-poly/200-segment geometry")
    else:
        workload = (f"BASELINE.json configs[1]: batch of {args.tiles} z=15 {dl.dim}x{dl.dim} tiles per GPU, synthetic 50-poly/200-segment "
                    "geometry per tile (SplitMix64, SURVEY.md 8(d)), lat/lon f64 input resident in HBM, RGBA8 framebuffers written to HBM"
                    + (f"; {args.streams} batches in flight (step i on resident copy / stream i mod {args.streams}: the pre-pass of one batch "
                       "overlaps the raster stage of the previous one; one_batch_at_a_time = the same steps strictly one after the other)"
                       if args.streams > 1 else ""))
    result = {
        "metric": "tiles/sec (256x256 z=15)",
        "value": head["total_tiles"] * args.steps / head["elapsed"],
        "unit": "tiles/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": head["elapsed"] / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "tiles_per_step_all_gpus": head["total_tiles"],
            "tiles_this_rank": dl.n_jobs,
            "polygons_per_tile": args.n_poly,
            "polylines_per_tile": args.n_line,
            "scale": args.scale,
            "named_config": bool(named),
            "sharding": "tile i -> rank i mod N; RCCL all-reduce(sum) of tile counts per step",
            "collective": collective,
            "native_comm_timed_out": comm_hung,
            "rccl_nranks_seen": nranks_seen,
            "batches_in_flight": args.streams,
        },
        "one_batch_at_a_time": dict(sequential, what="the same K steps with ONE resident batch on ONE stream: every step waits for the one before; "
                                                     "roofline.avg_launch_ms is measured in this run"),
        "roofline": {
            "kernel": "k_raster (fused fill/stroke/blend/to_rgb) — instruction-issue bound by construction, see roofline_issue; "
                      "the HBM fraction is reported because the metric asks for it, it is not this kernel's ceiling",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "algorithmic_bytes_per_launch": alg_bytes,
            "avg_launch_ms": raster_s * 1e3,
            "timing": "HIP events around the raster stage on the launch stream, averaged over the timed steps",
        },
    }

    # ---- configs[3]: the 10000-tile strong-scaling batch, at every N ---------------------------------
    if not args.no_extra and not strong and args.config4_tiles > 0:
        c4 = run_sharded(synth.config_tiles(args.config4_tiles), 15, 1, 50, 40, steps=max(3, min(args.steps, 10)), warmup=2)
        k4 = max(3, min(args.steps, 10))
        result["config4_strong"] = {
            "workload": f"BASELINE.json configs[3]: {args.config4_tiles} z=15 tiles, tile i -> rank i mod {world} "
                        "(x = 19000 + i mod 100, y = 10000 + i / 100), RCCL sum of tile counts per step",
            "value": c4["total_tiles"] * k4 / c4["elapsed"], "unit": "tiles/s", "scaling": "strong", "n_gpus": world,
            "steps": k4, "ms_per_step": c4["elapsed"] / k4 * 1e3, "tiles_this_rank": c4["dl"].n_jobs,
            "raster_ms_this_rank": c4["raster_ms"],
        }
        c4["scene"].free()
        del c4

    if solo and not args.no_extra:
        # ---- measured HBM copy ceiling (16 B/lane grid-stride copy kernel of the library, 1 GiB in + 1 GiB out) ----
        try:
            copy_gbs, read_gbs = ctx.hbm_copy_probe(1 << 30, 20)
            result["hbm_copy_ceiling"] = {
                "value": copy_gbs, "read_only": read_gbs, "unit": "GB/s", "frac_of_peak": copy_gbs / HBM_PEAK_GBS,
                "read_only_frac_of_peak": read_gbs / HBM_PEAK_GBS,
                "how": "osmt_hbm_copy_probe: nontemporal 16 B/lane grid-stride kernels, 1 GiB per launch, 20 launches each, HIP events on "
                       "the launch stream; value = copy (1 GiB read + 1 GiB written), read_only = the read half alone",
            }
            result["roofline"]["frac_of_copy_ceiling"] = achieved / copy_gbs
        except Exception as e:  # noqa: BLE001
            result["hbm_copy_ceiling"] = {"error": str(e)}

        # ---- configs[2] geometry: the same display lists at @2x --------------------------------------
        r2 = run_sharded(synth.config_tiles(256), 15, 2, 50, 40, steps=10, warmup=2)
        b2 = r2["dl"].algorithmic_bytes()
        result["raster_2x"] = {
            "workload": "BASELINE.json configs[2] geometry: 256 z=15 tiles at @2x (512x512), same 50-poly/200-segment lists",
            "tiles_per_s": 256 * 10 / r2["elapsed"], "ms_per_step": r2["elapsed"] / 10 * 1e3, "k_raster_ms": r2["raster_ms"],
            "algorithmic_bytes_per_launch": b2, "achieved": b2 / (r2["raster_ms"] / 1e3) / 1e9, "unit": "GB/s",
            "frac": b2 / (r2["raster_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS,
        }
        r2["scene"].free()
        del r2

        # ---- configs[4]: dense city -------------------------------------------------------------------
        n5 = args.config5_tiles
        if n5 > 0:
            r5 = run_sharded(synth.config_tiles(n5, x0=79000, y0=40000), 17, 1, 5000, 4000, steps=5, warmup=1,
                             maker=lambda xy: synth.make_tiles(xy, zoom=17, scale=1, n_poly=5000, n_line=4000, radius=(2.0, 12.0), step=12.0))
            b5 = r5["dl"].algorithmic_bytes()
            result["config5"] = {
                "workload": f"BASELINE.json configs[4]: dense city, {n5} z=17 tiles, 5000 polygons + 4000 polylines (20000 segments) per tile",
                "tiles_per_s": n5 * 5 / r5["elapsed"], "ms_per_step": r5["elapsed"] / 5 * 1e3, "k_raster_ms": r5["raster_ms"],
                "ops_per_tile": 9000, "algorithmic_bytes_per_tile": b5 / n5, "algorithmic_bytes_per_launch": b5,
                "achieved": b5 / (r5["raster_ms"] / 1e3) / 1e9, "unit": "GB/s", "frac": b5 / (r5["raster_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS,
            }
            r5["scene"].free()
            del r5
            if args.streams > 1:
                # the same mode as the headline: two resident batches, step i on copy / stream i mod 2 (the pre-pass is 40 % of
                # a dense step: it runs under the raster stage of the batch before); the strictly sequential figure stays beside it
                p5 = run_sharded(synth.config_tiles(n5, x0=79000, y0=40000), 17, 1, 5000, 4000, steps=6, warmup=2, slots=args.streams,
                                 maker=lambda xy: synth.make_tiles(xy, zoom=17, scale=1, n_poly=5000, n_line=4000, radius=(2.0, 12.0), step=12.0))
                p5["scene"].free()
                # one record, one regime.  tiles_per_s / ms_per_step / k_raster_ms / achieved stay the strictly sequential
                # run's (what rounds 1-4 reported under these keys); the headline's mode has keys of its own.
                c5 = result["config5"]
                c5["pipelined"] = {"tiles_per_s": n5 * 6 / p5["elapsed"], "ms_per_step": p5["elapsed"] / 6 * 1e3, "batches_in_flight": args.streams,
                                   "what": f"{args.streams} resident batches, step i on copy / stream i mod {args.streams} (like the headline)"}
                del p5

        # ---- end to end through the host-buffer ABI (upload + kernels + readback per call) -----------
        try:
            # every one-call leg: two untimed calls (the context's buffer cache, the runtime's staging for the pageable display
            # lists and the helper threads of the split upload settle over the first two), then the best of five
            def best_call(fn, warm=2, reps=5):
                for _ in range(warm):
                    fn()
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t0)
                return min(ts)

            pin = ctx.host_alloc((dl.n_jobs, dl.dim, dl.dim, 4))
            raw_s = best_call(lambda: ctx.render_batch_host(dl, out=pin))
            pin3 = ctx.host_alloc((dl.n_jobs, dl.dim * dl.dim * 3))
            rgb_s = best_call(lambda: ctx.render_batch_rgb(dl, out=pin3))
            ctx.host_free(pin3)
            pbuf = ctx.host_alloc((dl.n_jobs * 96 * 1024,))
            _, off = ctx.render_batch_png(dl, out=pbuf, as_bytes=False)
            png_s = best_call(lambda: ctx.render_batch_png(dl, out=pbuf, as_bytes=False), warm=1)
            # the same call from the reference's server shape (http_server.rs:50-83: a pool of workers, one request each):
            # W host threads issue PNG batches back to back; one thread's validation / upload overlaps another's kernels
            import threading

            def workers(n_thr, calls=4):
                bufs = [ctx.host_alloc((dl.n_jobs * 96 * 1024,)) for _ in range(n_thr)]
                def run(b):
                    for _ in range(calls):
                        ctx.render_batch_png(dl, out=b, as_bytes=False)
                # one untimed round of the same shape first: n_thr jobs in flight take n_thr device buffers and pinned staging
                # areas out of the context's caches, and the first concurrent round is the one that has to allocate them
                for timed in (False, True):
                    th = [threading.Thread(target=run, args=(b,)) for b in bufs]
                    t0 = time.perf_counter()
                    for t in th:
                        t.start()
                    for t in th:
                        t.join()
                    dt = time.perf_counter() - t0
                for b in bufs:
                    ctx.host_free(b)
                return n_thr * calls * dl.n_jobs / dt
            pooled = {str(w): workers(w) for w in (2, 4)}
            # ... and from ONE thread with the call split in two (osmt_render_batch_png_begin / _end): batch k + 1 is validated,
            # uploaded and queued while the GPU works on batch k; two jobs in flight, two pinned output buffers
            pb = [ctx.host_alloc((dl.n_jobs * 96 * 1024,)) for _ in range(2)]
            n_pipe = 16
            # an untimed round of the same length first: the second job's buffers come out of the context's caches and the
            # runtime's own staging for the pageable display lists settles (a cold round runs at 3.0 ms per batch, every
            # later one at 2.1: tools/bench_png_begin_end.py); then the timed one
            for n_round in (n_pipe, n_pipe):
                t0 = time.perf_counter()
                prev = ctx.png_begin(dl)
                for k in range(1, n_round):
                    cur = ctx.png_begin(dl)
                    ctx.png_end(prev, pb[(k - 1) & 1])
                    prev = cur
                _, off_p = ctx.png_end(prev, pb[(n_round - 1) & 1])
                pipe_s = (time.perf_counter() - t0) / n_round
            assert int(off_p[-1]) == int(off[-1])
            for b_ in pb:
                ctx.host_free(b_)
            result["end_to_end"] = {
                "what": "wall clock around one osmt_render_batch / osmt_render_batch_png call (validation + H2D of the display lists + all "
                        "kernels + D2H into pinned host memory), best of 5 after two untimed calls; never `value`",
                "tiles": dl.n_jobs,
                "raw_rgba8_pinned_tiles_per_s": dl.n_jobs / raw_s, "raw_rgba8_ms": raw_s * 1e3,
                "raw_rgb8_pinned_tiles_per_s": dl.n_jobs / rgb_s, "raw_rgb8_ms": rgb_s * 1e3,
                "png_files_pinned_tiles_per_s": dl.n_jobs / png_s, "png_ms": png_s * 1e3, "png_bytes_per_tile": float(off[-1]) / dl.n_jobs,
                "png_files_worker_threads_tiles_per_s": pooled,
                "png_files_begin_end_tiles_per_s": dl.n_jobs / pipe_s, "png_begin_end_ms_per_batch": pipe_s * 1e3,
                "png_begin_end_what": "one caller thread, osmt_render_batch_png_begin(k + 1) before osmt_render_batch_png_end(k), 16 batches "
                                      "after an untimed round of 16 (steady state)",
            }
            ctx.host_free(pin)
            ctx.host_free(pbuf)
            # ---- small-batch latency: what a drop-in behind the reference's server lives on --------------------
            # http_server.rs:134-181 renders ONE tile per request per worker; W workers share one Drawer (:50-83).  Here:
            # W host threads on ONE context, each issuing osmt_render_batch_rgb calls of B tiles back to back (its own
            # display list, its own pinned RGB8 buffer); per-call wall clock around the C call only.
            result["end_to_end"]["latency"] = latency_leg(ctx, args)
            result["end_to_end"]["worker_entry"] = latency_leg(ctx, args, batches=(1,), workers=(1, 4, 16), seconds=0.5, entry="worker")
        except Exception as e:  # noqa: BLE001
            result["end_to_end"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- composite pass (configs[2]: @2x 512x512, 8 layers) — rank 0, N = 1 only ------
    if solo and not args.no_composite:
        n, L, dim = args.composite_tiles, 8, 512
        planes = synth.composite_planes(n, L=L, dim=dim, device=dev)
        cout = torch.empty((n, dim, dim, 4), dtype=torch.uint8, device=dev)
        canvas = [0xFC / 255.0, 0xF8 / 255.0, 0xE4 / 255.0, 1.0]
        for _ in range(3):
            ctx.composite(planes, canvas, out=cout)
        reps = 20
        cev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in cev:
            a.record()
            ctx.composite(planes, canvas, out=cout)
            b.record()
        torch.cuda.synchronize()
        c_s = sum(a.elapsed_time(b) for a, b in cev) / reps / 1e3
        c_bytes = n * (L * dim * dim * 32 + dim * dim * 4)  # B_comp, SURVEY.md 8(d)
        result["roofline_composite"] = {
            "kernel": "k_composite<8> (8-layer premultiplied f64 over + to_rgb, 512x512)",
            "bound": "hbm",
            "achieved": c_bytes / c_s / 1e9,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": c_bytes / c_s / 1e9 / HBM_PEAK_GBS,
            "traffic": None,
            "algorithmic_bytes_per_launch": c_bytes,
            "avg_launch_ms": c_s * 1e3,
            "tiles_per_s": n / c_s,
            "tiles_per_launch": n,
        }
        if "hbm_copy_ceiling" in result and "value" in result["hbm_copy_ceiling"]:
            # the composite reads 64 bytes for every byte it writes: its ceiling is the read stream, not the copy
            result["roofline_composite"]["frac_of_read_ceiling"] = c_bytes / c_s / 1e9 / result["hbm_copy_ceiling"]["read_only"]
            result["roofline_composite"]["frac_of_copy_ceiling"] = c_bytes / c_s / 1e9 / result["hbm_copy_ceiling"]["value"]
        del planes, cout

    # ---- PNG files written by the GPU from the framebuffers of the timed steps — rank 0, N = 1 only ----
    if solo and not args.no_png:
        slots, lens = ctx.encode_png_device(out)
        torch.cuda.synchronize()
        reps = 10
        p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p0.record()
        for _ in range(reps):
            slots, lens = ctx.encode_png_device(out)
        p1.record()
        torch.cuda.synchronize()
        p_s = p0.elapsed_time(p1) / reps / 1e3
        png_bytes = int(lens.sum().item())
        p_alg = out.numel() + png_bytes  # RGBA8 read once + files written once
        result["png_encode"] = {
            "kernel": "k_png_encode (Paeth + deflate + Adler-32/CRC-32 on the device, one workgroup per tile)",
            "tiles": int(out.shape[0]),
            "avg_launch_ms": p_s * 1e3,
            "tiles_per_s": out.shape[0] / p_s,
            "png_bytes_per_tile": png_bytes / out.shape[0],
            "ratio_vs_rgba8": out.numel() / png_bytes,
            "bound": "hbm", "achieved": p_alg / p_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": p_alg / p_s / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": p_alg,
        }
        del slots, lens

    # ---- label pass on top of the same area workload — rank 0, N = 1 only ----
    if solo and not args.no_labels:
        from osm_renderer_amd import labels as labels_mod

        n = args.label_tiles
        pool = min(32, n)
        sizes = [(16, 16), (12, 20), (20, 20)]
        rng = np.random.default_rng(1)
        first_img = None
        for h, w in sizes:
            iid = ctx.register_image(rng.integers(0, 256, size=(h, w, 4)).astype(np.uint8))
            first_img = iid if first_img is None else first_img
        base = labels_mod.make_labels(pool, labels_per_tile=24, scale=args.scale, n_images=3, image_sizes=sizes, seed=2)
        base.labels["image_id"] += first_img
        ll = labels_mod.concat_labels([base.subset([i % pool]) for i in range(n)])
        ldl = synth.make_tiles(synth.config_tiles(n), zoom=15, scale=args.scale, n_poly=args.n_poly, n_line=args.n_line)
        lscene = ctx.upload(ldl)
        lout = torch.empty((n, ldl.dim, ldl.dim, 4), dtype=torch.uint8, device=dev)

        def timed(reps=10):
            for _ in range(2):
                ctx.render(lscene, lout)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ctx.render(lscene, lout)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        ms_plain = timed()
        lscene.set_labels(ll)
        ms_lab = timed()
        ok = lscene.label_status()
        d_ms = max(ms_lab - ms_plain, 1e-6)
        result["label_pass"] = {
            "workload": f"{n} config-2 tiles + 24 synthetic labels per tile (TrueType-like outlines flattened like draw_quad; "
                        "40 % with an icon, 30 % rotated), pool of 32 tiles repeated",
            "labels": int(len(ll.labels)),
            "draw_line_calls": int(len(ll.segs)),
            "labels_succeeded": int(ok.sum()),
            "ms_areas_only": ms_plain,
            "ms_with_labels": ms_lab,
            "label_pass_ms": d_ms,
            "labels_per_s": len(ll.labels) / d_ms * 1e3,
            "draw_line_calls_per_s": len(ll.segs) / d_ms * 1e3,
            "tiles_per_s_with_labels": n / ms_lab * 1e3,
            "algorithmic_bytes": ll.algorithmic_bytes(),
        }
        if not args.no_cpu_baseline:
            from oracle import oracle_py

            oracle_py.build()
            nt = min(16, os.cpu_count() or 1)
            n_cpu = min(n, 256)  # enough work for the difference of the two timings to stand clear of the noise
            sub_dl, sub_ll = ldl.subset(range(n_cpu)), ll.subset(range(n_cpu))
            lpool = oracle_py.Pool(nt, args.scale)
            lbuf = np.empty((n_cpu, ldl.dim, ldl.dim, 4), dtype=np.uint8)
            lstat = np.zeros(len(sub_ll.labels), dtype=np.uint8)

            def best(fn, reps=3):
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t0)
                return min(ts)

            t_plain = best(lambda: lpool.render(sub_dl, lbuf))
            t_lab = best(lambda: lpool.render(sub_dl, lbuf, labels=sub_ll, status=lstat))
            lpool.close()
            cpu_s = max(t_lab - t_plain, 1e-9)
            result["label_pass"]["cpu_baseline"] = {
                "value": len(sub_ll.labels) / cpu_s, "unit": "labels/s", "cores": nt, "kind": "port",
                "sample": f"{n_cpu} tiles, {len(sub_ll.labels)} labels: pooled oracle render with labels ({t_lab:.3f} s) minus "
                          f"without ({t_plain:.3f} s), best of 3 each",
            }
        lscene.free()
        del lout

    # ---- CPU baseline: the oracle on the host cores (rank 0, N = 1 only) ---------------
    if solo and not args.no_cpu_baseline:
        from oracle import oracle_py

        oracle_py.build()
        cores = os.cpu_count() or 1
        # Persistent worker pool = the reference's server: one TilePixels per worker, created at start-up
        # (http_server.rs:69-72), tiles dealt round-robin (:105-108).  Pool, canvases and the output array exist
        # BEFORE the timer starts.  Thread sweep: the path rewrites a 47 MB canvas per tile, so it can turn
        # memory-bound before all cores are used; every point of the sweep is reported.
        probe = dl.subset(range(min(8, dl.n_jobs)))
        p1 = oracle_py.Pool(1, args.scale)
        pbuf = np.empty((probe.n_jobs, dl.dim, dl.dim, 4), dtype=np.uint8)
        p1.render(probe, pbuf)
        t = time.perf_counter()
        p1.render(probe, pbuf)
        per_tile = (time.perf_counter() - t) / probe.n_jobs
        p1.close()
        tried, cpu_out, n_first = {}, None, 0
        budget_s = 2.5  # seconds of wall clock per sweep point (5 points), i.e. ~10-30 s of CPU work per point at the quota
        quota = cpu_quota_cores()
        usable = int(min(cores, quota)) if quota else cores  # threads beyond the cgroup quota only get throttled
        points = {1, max(1, usable // 4), max(1, usable // 2), usable, min(cores, 2 * usable)}
        for th in sorted(points):
            n_th = int(min(8192, max(2 * th, budget_s * min(th, usable) / per_tile)))
            sample = synth.make_tiles(synth.config_tiles(n_th), zoom=15, scale=args.scale, n_poly=args.n_poly, n_line=args.n_line)
            pool_t = oracle_py.Pool(th, args.scale)
            buf = np.empty((n_th, dl.dim, dl.dim, 4), dtype=np.uint8)
            buf[:] = 0  # touch the pages before the timer
            t = time.perf_counter()
            pool_t.render(sample, buf)
            dt = time.perf_counter() - t
            tried[th] = {"tiles": n_th, "seconds": dt, "tiles_per_s": n_th / dt}
            pool_t.close()
            if cpu_out is None:
                cpu_out, n_first = buf.copy(), n_th
            del buf
        best_threads = max(tried, key=lambda k: tried[k]["tiles_per_s"])
        n_cmp = min(n_first, dl.n_jobs) if not strong and world == 1 else 0
        match = bool(np.array_equal(out[:n_cmp].cpu().numpy(), cpu_out[:n_cmp])) if n_cmp else None
        result["cpu_baseline"] = {
            "value": tried[best_threads]["tiles_per_s"],
            "unit": "tiles/s",
            "cores": best_threads,
            "host_logical_cpus": cores,
            "cgroup_cpu_quota_cores": cpu_quota_cores(),
            "cpu_model": cpu_model(),
            "sweep": {str(k): v for k, v in tried.items()},
            "kind": "port",
            "sample": "per sweep point ~3 s of the same workload (the batch's own tiles first): C++ oracle = restatement of the reference's "
                      "Rust CPU path incl. its 3x3-tile canvas, NOT the Rust binary; persistent pool, one canvas per worker allocated before "
                      "the timer, output pre-allocated, tiles round-robin; best point reported as value",
            "single_thread_tiles_per_s": 1.0 / per_tile,
            "single_tile_us": per_tile * 1e6,
            "gpu_matches_oracle_on_sample": match,
        }

    if "cpu_baseline" in result and isinstance(result.get("end_to_end"), dict) and "latency" in result["end_to_end"]:
        result["end_to_end"]["latency"]["oracle_single_tile_us"] = result["cpu_baseline"]["single_tile_us"]

    # ---- counters of THIS run: child rocprofv3 passes over the same step (N = 1 only) ------------------
    if solo and not args.no_pmc and named and args.tiles == 1024 and args.scale == 1 and not strong:
        from tools import pmc_pass

        keep = args.keep_pmc or None
        pm = {"how": "child processes: rocprofv3 --kernel-trace --pmc <set> -- python bench.py --pmc-child all (4 composite launches, then 4 config-2 steps), "
                     "per-dispatch averages; FETCH_SIZE / WRITE_SIZE in separate passes, unit KB; FETCH_SIZE x2 = the gfx950 correction "
                     "of MI355X_MICROARCH.md for wide coalesced reads (uncalibrated for k_raster's small scattered reads: both given)"}
        child = ["all" if "roofline_composite" in result else "config2"]
        sq = pmc_pass.run_pass(pmc_pass.SQ_PASS_1, child, keep_dir=keep)
        hit = pmc_pass.pick(sq, "k_raster") if "error" not in sq else None
        if hit:
            v = hit[1]
            valu, salu = v.get("SQ_INSTS_VALU", 0.0), v.get("SQ_INSTS_SALU", 0.0)
            act = v.get("SQ_ACTIVE_INST_VALU", 0.0)  # quad-cycles summed over waves
            busy_ms = act * 4.0 / N_SIMD / CLOCK_HZ * 1e3
            floor4 = valu * 4.0 / N_SIMD / CLOCK_HZ * 1e3
            result["roofline_issue"] = {
                "kernel": "k_raster", "bound": "valu-issue",
                "valu_wave_instr": valu, "salu_wave_instr": salu,
                "valu_busy_quad_cycles": act, "wave_quad_cycles": v.get("SQ_WAVE_CYCLES"), "wait_any_quad_cycles": v.get("SQ_WAIT_ANY"),
                "wait_inst_any_quad_cycles": v.get("SQ_WAIT_INST_ANY"), "busy_cycles": v.get("SQ_BUSY_CYCLES"),
                "floor_ms": busy_ms,
                "floor_ms_at_4_cycles_per_valu": floor4,
                "floor_ms_at_2_cycles_per_valu": floor4 / 2.0,
                "measured_ms": raster_s * 1e3, "profiled_ms": v.get("avg_us", 0.0) / 1e3,
                "frac": busy_ms / (raster_s * 1e3) if raster_s > 0 else None,
                "note": "floor_ms = SQ_ACTIVE_INST_VALU (quad-cycles the SIMDs spent issuing VALU) x 4 / (1024 SIMDs x 2.4 GHz): the time "
                        "the kernel's own VALU stream needs with perfect overlap of everything else; frac = floor_ms / HIP-event kernel time. "
                        "f64 VALU issues at 4 cycles per wave-instruction (16 lanes/cycle), f32/int at 2; both uniform-rate floors are given too",
            }
        else:
            result["roofline_issue"] = {"error": sq.get("error", "k_raster not found in the SQ pass")}
        fe = pmc_pass.run_pass(["FETCH_SIZE"], child, keep_dir=keep)
        wr = pmc_pass.run_pass(["WRITE_SIZE"], child, keep_dir=keep)
        hf = pmc_pass.pick(fe, "k_raster") if "error" not in fe else None
        hw = pmc_pass.pick(wr, "k_raster") if "error" not in wr else None
        if hf and hw and "FETCH_SIZE" in hf[1] and "WRITE_SIZE" in hw[1]:
            f_b, w_b = hf[1]["FETCH_SIZE"] * 1024.0, hw[1]["WRITE_SIZE"] * 1024.0
            result["roofline"]["traffic"] = f_b * 2.0 + w_b
            result["roofline"]["traffic_raw_fetch"] = f_b + w_b
            result["roofline"]["traffic_note"] = "bytes per launch measured in THIS run (child rocprofv3 passes): 2 x FETCH_SIZE + WRITE_SIZE"
            pm["all_kernels"] = {k: {"fetch_kb": fe.get(k, {}).get("FETCH_SIZE"), "write_kb": wr.get(k, {}).get("WRITE_SIZE"),
                                     "avg_us": fe.get(k, {}).get("avg_us")} for k in fe if "osmt" in k or "k_" in k}
        else:
            result["roofline"]["traffic_note"] = "counter pass failed: " + str(fe.get("error") or wr.get("error") or "k_raster not found")
        if "roofline_composite" in result:
            hfc = pmc_pass.pick(fe, "k_composite") if "error" not in fe else None
            hwc = pmc_pass.pick(wr, "k_composite") if "error" not in wr else None
            if hfc and hwc and "FETCH_SIZE" in hfc[1] and "WRITE_SIZE" in hwc[1]:
                result["roofline_composite"]["traffic"] = hfc[1]["FETCH_SIZE"] * 2048.0 + hwc[1]["WRITE_SIZE"] * 1024.0
                result["roofline_composite"]["traffic_note"] = "measured in THIS run: 2 x FETCH_SIZE + WRITE_SIZE (wide coalesced 16 B/lane stream)"
        result["pmc"] = pm
    elif solo and not args.no_pmc:
        result["roofline"]["traffic_note"] = "counter passes only run for the named config (1024 tiles, scale 1, weak mode)"

    # ---- sustained: >= 10 s of back-to-back steps, LAST — after the CPU sweeps and the profiler children, so that a sampler
    # watching the GPU from outside sees the headline workload run uninterrupted at the end of the process ----------------
    if solo and not args.no_extra and args.sustained_seconds > 0:
        n_sus = max(10, int(args.sustained_seconds / max(head["elapsed"] / args.steps, 1e-5)) + 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            head["step"]()
        torch.cuda.synchronize()
        t_sus = time.perf_counter() - t0
        result["sustained"] = {"steps": n_sus, "seconds": t_sus, "tiles_per_s": n_sus * dl.n_jobs / t_sus, "ms_per_step": t_sus / n_sus * 1e3}

    # top-level scalar copies of the figures the round's bars are set on (a driver that keeps only scalars keeps these)
    def _pick(path):
        cur = result
        for k in path:
            if not isinstance(cur, dict) or k not in cur:
                return None
            cur = cur[k]
        return cur if isinstance(cur, (int, float)) else None

    # config5_tiles_per_s: two batches in flight like the headline (what round 5 reported under this name) when that leg ran,
    # config5_sequential_tiles_per_s: one batch at a time (rounds 1-4)
    for name, path in (("config5_tiles_per_s", ("config5", "pipelined", "tiles_per_s") if _pick(("config5", "pipelined", "tiles_per_s")) else ("config5", "tiles_per_s")),
                       ("config5_sequential_tiles_per_s", ("config5", "tiles_per_s")), ("raster_2x_tiles_per_s", ("raster_2x", "tiles_per_s")),
                       ("config4_strong_tiles_per_s", ("config4_strong", "value")), ("label_pass_ms", ("label_pass", "label_pass_ms")),
                       ("sustained_tiles_per_s", ("sustained", "tiles_per_s")), ("composite_hbm_frac", ("roofline_composite", "frac")),
                       ("raster_issue_frac", ("roofline_issue", "frac")), ("k_raster_ms", ("roofline", "avg_launch_ms")),
                       ("png_files_tiles_per_s", ("end_to_end", "png_files_pinned_tiles_per_s")),
                       ("png_files_begin_end_tiles_per_s", ("end_to_end", "png_files_begin_end_tiles_per_s")),
                       ("png_bytes_per_tile", ("end_to_end", "png_bytes_per_tile")),
                       ("worker16_tiles_per_s", ("end_to_end", "worker_entry", "cases", "workers16", "tiles_per_s")),
                       ("worker16_p99_us", ("end_to_end", "worker_entry", "cases", "workers16", "p99_us")),
                       ("worker1_p50_us", ("end_to_end", "worker_entry", "cases", "workers1", "p50_us"))):
        v = _pick(path)
        if v is not None:
            result[name] = v

    if rank == 0:
        print(json.dumps(result), flush=True)
    if comm_hung:
        # a helper thread is still inside RCCL: tearing down the context or the process group under it can block — the line is out
        sys.stderr.flush()
        os._exit(0)
    scene.free()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
