#!/bin/bash
# round-4 run f: worker entry after the pinned-pool fix and the polling wait; group trace; quick stage times with the memset folded away
O=gpurun_out/r04_f; mkdir -p $O
for m in 1 2 3; do echo "OSMT_WORKER_INFLIGHT=$m"; OSMT_WORKER_INFLIGHT=$m timeout 200 bash tools/worker_bench.sh 1 4 16 32 2>&1 | grep '"entry"'; done > $O/worker_inflight.txt 2>&1; cat $O/worker_inflight.txt
echo "no polling wait:"; OSMT_SPIN_SYNC=0 timeout 100 bash tools/worker_bench.sh 1 16 2>&1 | grep '"entry"' | tee $O/worker_nospin.txt
OSMT_TRACE_WORKER=1 timeout 100 bash tools/worker_bench.sh 16 2> $O/worker_trace.err > /dev/null; python3 - <<'PY'
import re,collections
rows=[tuple(map(float,m.groups())) for m in re.finditer(r"group: (\d+) requests, (\d+) tiles, merge\+staging (\d+) us, render (\d+) us", open('gpurun_out/r04_f/worker_trace.err').read())]
by=collections.defaultdict(list)
for r,t,mg,rd in rows: by[int(t)].append((mg,rd))
print("groups", len(rows))
for t in sorted(by): v=by[t]; print(f"tiles {t:3d}: {len(v):5d} groups, merge {sum(a for a,_ in v)/len(v):6.1f} us, render {sum(b for _,b in v)/len(v):7.1f} us")
PY
timeout 300 python tools/time_variants.py base > $O/stage_times.txt 2>&1; cat $O/stage_times.txt
timeout 120 python -m pytest tests/test_gpu_worker.py tests/test_gpu_threads.py -x -q 2>&1 | tail -2
