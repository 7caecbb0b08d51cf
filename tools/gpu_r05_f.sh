#!/bin/bash
# Round-5 run F: does a PINNED caller buffer change how one-tile requests of several threads overlap?  (bench.py's Python latency leg — pinned
# buffers — shows 4 threads at the throughput of 1 through osmt_render_batch_rgb; the native bench — pageable buffers — shows 3.4 x)
TAG=${1:-r05_f}
O=gpurun_out/$TAG; mkdir -p $O
{ echo "# pageable caller buffers"; timeout 200 bash tools/worker_bench.sh 1 4 16; echo "# OSMT_BENCH_PINNED=1"; OSMT_BENCH_PINNED=1 timeout 200 bash tools/worker_bench.sh 1 4 16; } > $O/worker_pinned.txt 2>&1; cat $O/worker_pinned.txt
