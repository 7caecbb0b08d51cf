#!/bin/bash
# Round-5 run K: groups of the worker entry in flight (OSMT_WORKER_INFLIGHT 1 .. 4) now that small groups are zero-copy; then 2 x 200 s more fuzz under poison
TAG=${1:-r05_k}
O=gpurun_out/$TAG; mkdir -p $O
for f in 1 2 3 4; do echo "# OSMT_WORKER_INFLIGHT=$f"; OSMT_WORKER_INFLIGHT=$f timeout 200 bash tools/worker_bench.sh 16 32 64 2>&1 | grep worker_render; done > $O/worker_inflight.txt 2>&1; cat $O/worker_inflight.txt
timeout 300 python tools/fuzz_parity.py 200 5501 > $O/fuzz_areas.txt 2>&1; tail -1 $O/fuzz_areas.txt
timeout 300 python tools/fuzz_parity.py 200 5502 labels > $O/fuzz_labels.txt 2>&1; tail -1 $O/fuzz_labels.txt
