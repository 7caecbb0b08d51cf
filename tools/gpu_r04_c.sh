#!/bin/bash
# round-4 run c: one-lane-per-pixel walk: parity suite, stage times vs round 3, fuzz, work counters; worker entry at 2 / 4 / 8 groups in flight
O=gpurun_out/r04_c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
OSMT_TIME_BIG=1 timeout 600 python tools/time_variants.py base r3 abl1 abl3 > $O/stage_times.txt 2>&1; cat $O/stage_times.txt
OSMT_LIB=$PWD/osm_renderer_amd/libosmtile_dbg.so timeout 120 python tools/dbg_counts.py config2 > $O/dbg_counts.txt 2>&1; cat $O/dbg_counts.txt
timeout 200 python tools/fuzz_parity.py 120 41 > $O/fuzz_areas.txt 2>&1; tail -3 $O/fuzz_areas.txt
for m in 2 4 8; do echo "OSMT_WORKER_INFLIGHT=$m"; OSMT_WORKER_INFLIGHT=$m timeout 200 bash tools/worker_bench.sh 1 4 16 32 2>&1 | grep worker_render; done > $O/worker_inflight.txt 2>&1; cat $O/worker_inflight.txt
