#!/usr/bin/env python
"""bench.py — tiles/sec of the MI355X tile hot path on BASELINE.json configs[1].

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (project -> per-op pre-pass -> fused
fill/stroke/blend raster -> RGBA8 framebuffer) over one batch of 1024 synthetic
z=15 256x256 tiles (50 polygons + 200 stroke segments each) PER GPU, display
lists already resident in HBM, framebuffers written to HBM.  With N > 1 (launched
by torch.distributed.run, one rank per GPU) tile i of the global batch belongs to
rank i mod N (weak scaling: 1024 tiles per GPU), no data-path collective; one
RCCL all-reduce of the tile count per step is the only communication.

Rank 0 prints ONE JSON line with the whole-job tiles/s plus
  roofline            the dominant kernel (k_raster): algorithmic bytes / kernel time vs 8 TB/s
  roofline_composite  the 8-layer @2x composite pass (the kernel the >= 40 % HBM target is on)
  cpu_baseline        the C++ oracle (restatement of the reference's Rust CPU path, NOT the
                      Rust binary) on the host cores, bounded sample of the same workload
  png_encode          SURVEY.md 8(f) N3: the framebuffers of the step turned into PNG files on the GPU
  label_pass          SURVEY.md 8(f) N1: the same tiles with 24 synthetic labels per tile on top
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tiles", type=int, default=1024, help="tiles per GPU per step")
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--composite-tiles", type=int, default=64)
    ap.add_argument("--n-poly", type=int, default=50, help="diagnostic: polygons per tile (default = the named config)")
    ap.add_argument("--n-line", type=int, default=40, help="diagnostic: polylines per tile (default = the named config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-composite", action="store_true")
    ap.add_argument("--no-labels", action="store_true")
    ap.add_argument("--no-png", action="store_true")
    ap.add_argument("--label-tiles", type=int, default=1024)
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from osm_renderer_amd import abi, synth
    from osm_renderer_amd.renderer import Context

    ctx = Context(local_rank)
    dev = ctx.device

    # ---- workload: tile i of the global batch -> rank i mod world --------------------
    global_tiles = synth.config_tiles(args.tiles * world)
    mine = global_tiles[rank::world]
    dl = synth.make_tiles(mine, zoom=15, scale=args.scale, n_poly=args.n_poly, n_line=args.n_line)
    scene = ctx.upload(dl)
    out = torch.empty((dl.n_jobs, dl.dim, dl.dim, 4), dtype=torch.uint8, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    alg_bytes = dl.algorithmic_bytes()  # 16*N_pts + 64*N_ops + 8*N_dashes + 4*W*H per tile

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        ctx.render_stages(scene, abi.STAGE_PROJECT | abi.STAGE_OPINFO)
        if i is not None:
            ev[i][0].record()
        ctx.render_stages(scene, abi.STAGE_RASTER, out)
        if i is not None:
            ev[i][1].record()
        if dist is not None:
            count.fill_(dl.n_jobs)
            dist.all_reduce(count)  # RCCL sum of tile counts (the path's only collective)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        total_tiles_per_step = int(count.item())
        assert total_tiles_per_step == args.tiles * world, (total_tiles_per_step, args.tiles, world)
    else:
        total_tiles_per_step = dl.n_jobs

    raster_ms = [a.elapsed_time(b) for a, b in ev]
    raster_avg_s = sum(raster_ms) / len(raster_ms) / 1e3
    achieved = alg_bytes / raster_avg_s / 1e9

    result = {
        "metric": "tiles/sec (256x256 z=15)",
        "value": total_tiles_per_step * args.steps / elapsed,
        "unit": "tiles/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE.json configs[1]: batch of {args.tiles} z=15 {dl.dim}x{dl.dim} tiles per GPU, "
            "synthetic 50-poly/200-segment geometry per tile (SplitMix64, SURVEY.md 8(d)), lat/lon f64 input "
            "resident in HBM, RGBA8 framebuffers written to HBM",
            "tiles_per_gpu": args.tiles,
            "polygons_per_tile": args.n_poly,
            "polylines_per_tile": args.n_line,
            "scale": args.scale,
            "sharding": "tile i -> rank i mod N; RCCL all-reduce(sum) of tile counts per step",
        },
        "roofline": {
            "kernel": "k_raster (fused fill/stroke/blend/to_rgb; LDS/ALU-bound by construction, see DESIGN.md)",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "algorithmic_bytes_per_launch": alg_bytes,
            "avg_launch_ms": raster_avg_s * 1e3,
        },
    }

    # ---- composite pass (configs[2]: @2x 512x512, 8 layers) — rank 0, N = 1 only ------
    if rank == 0 and world == 1 and not args.no_composite:
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
        c_bytes = n * (L * dim * dim * 32 + dim * dim * 4)  # B_comp, SURVEY.md §8(d)
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
        del planes, cout

    # ---- PNG files written by the GPU from the framebuffers of the timed steps — rank 0, N = 1 only ----
    if rank == 0 and world == 1 and not args.no_png:
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
            "kernel": "k_png_encode_fast (Paeth + fixed-Huffman run-length deflate + Adler-32/CRC-32, one workgroup per tile)",
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
    if rank == 0 and world == 1 and not args.no_labels:
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

            def best(fn, reps=3):
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t0)
                return min(ts)

            t_plain = best(lambda: oracle_py.render_batch(sub_dl, threads=nt))
            t_lab = best(lambda: oracle_py.render_batch(sub_dl, threads=nt, labels=sub_ll))
            cpu_s = max(t_lab - t_plain, 1e-9)
            result["label_pass"]["cpu_baseline"] = {
                "value": len(sub_ll.labels) / cpu_s, "unit": "labels/s", "cores": nt, "kind": "port",
                "sample": f"{n_cpu} tiles, {len(sub_ll.labels)} labels: oracle render with labels ({t_lab:.3f} s) minus "
                          f"without ({t_plain:.3f} s), best of 3 each",
            }
        lscene.free()
        del lout

    # ---- CPU baseline: the oracle on the host cores (rank 0, N = 1 only) ---------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_py

        cores = os.cpu_count() or 1
        probe = dl.subset(range(min(8, dl.n_jobs)))
        t = time.perf_counter()
        oracle_py.render_batch(probe, threads=1)
        per_tile = (time.perf_counter() - t) / probe.n_jobs
        # bounded sample (~10-30 s of CPU work in total): at least 8 tiles per thread so the per-thread
        # canvas allocation is amortised, at most 4096 tiles; tiles beyond the step's own batch continue
        # the same generator.  The reference's path is memory-bound (a 47 MB canvas is rewritten per
        # tile), so more threads is not always faster: a few thread counts are timed, the best is reported.
        tried = {}
        cpu_out = None
        n_sample = 0
        for th in sorted({cores, max(1, cores // 2), max(1, cores // 4), max(1, cores // 8), max(1, cores // 16)}):
            n_th = int(min(8192, 32 * th))  # 32 tiles per thread: steady state, not canvas allocation
            sample = synth.make_tiles(synth.config_tiles(n_th * world)[rank::world], zoom=15, scale=args.scale,
                                      n_poly=args.n_poly, n_line=args.n_line)
            t = time.perf_counter()
            o = oracle_py.render_batch(sample, threads=th)
            tried[th] = n_th / (time.perf_counter() - t)
            if cpu_out is None:
                cpu_out, n_sample = o, n_th
        best_threads = max(tried, key=tried.get)
        cpu_s = 1.0 / tried[best_threads]  # seconds per tile at the best thread count
        n_cmp = min(n_sample, dl.n_jobs)
        cpu_out = cpu_out[:n_cmp]
        n_sample_cmp = n_cmp
        gpu_out = out[:n_sample_cmp].cpu().numpy()
        result["cpu_baseline"] = {
            "value": 1.0 / cpu_s,
            "unit": "tiles/s",
            "cores": best_threads,
            "host_logical_cpus": cores,
            "tiles_per_s_by_threads": {str(k): v for k, v in tried.items()},
            "kind": "port",
            "sample": f"32 tiles per thread (max 8192) of the same workload (the batch's own tiles first), C++ oracle (restatement of the reference's Rust "
            f"CPU path incl. its 3x3-tile canvas), best of {sorted(tried)} threads, one canvas per thread, tiles round-robin; "
            f"single-thread probe {1.0 / per_tile:.1f} tiles/s",
            "single_thread_tiles_per_s": 1.0 / per_tile,
            "gpu_matches_oracle_on_sample": bool(np.array_equal(gpu_out, cpu_out)),
        }

    # HBM bytes per launch from the PMC counters cannot be read in-process: they are collected by
    # separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this same command and
    # committed as profiles/r01_hbm_traffic.json (with the gfx950 FETCH_SIZE correction noted there).
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
            tr = json.load(f)
        if args.tiles == 1024 and args.scale == 1 and args.n_poly == 50 and args.n_line == 40:
            result["roofline"]["traffic"] = tr["k_raster"]["traffic_bytes_per_launch_fetch_x2"]
            result["roofline"]["traffic_note"] = "bytes/launch, rocprofv3 PMC pass of an earlier run of this command"
        if "png_encode" in result and "k_png_encode_fast" in tr and args.tiles == 1024 and args.scale == 1:
            result["png_encode"]["traffic"] = tr["k_png_encode_fast"]["traffic_bytes_per_launch_fetch_x2"]
        if "label_pass" in result and "k_label_cover" in tr and args.label_tiles == 1024 and args.scale == 1:
            result["label_pass"]["traffic_k_label_cover"] = tr["k_label_cover"]["traffic_bytes_per_launch_fetch_x2"]
        if "roofline_composite" in result and args.composite_tiles == 64:
            result["roofline_composite"]["traffic"] = tr["k_composite"]["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        print(json.dumps(result))
    scene.free()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
