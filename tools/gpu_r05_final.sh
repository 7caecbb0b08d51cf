#!/bin/bash
# Round-5 closing run on HEAD: the full GPU suite under the poisoned allocator (log with the commit hash in its first line), the bench
# line, kernel traces + counter passes of config 2 / config 5 (256) / @2x / composite / the small-batch (FOLD) instantiation on a one-tile and
# a 64-tile batch, labels, native worker bench, one-tile latency, smoke, 2 x 150 s of fuzz under poison.
TAG=${1:-r05_final}
O=gpurun_out/$TAG; mkdir -p $O
( echo "HEAD $(cat .git_head 2>/dev/null)  (pytest -m gpu, OSMT_POISON_ALLOC=1 via tests/conftest.py)"; timeout 1500 python -m pytest tests -m gpu -q --timeout=600 --durations=6 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -14 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err
timeout 600 python tools/prof_workload.py config2 $O/config2 kt,sq1,sq2,fetch,write > $O/prof.log 2>&1
timeout 900 python tools/prof_workload.py config5:256 $O/config5 kt,sq1,fetch,write >> $O/prof.log 2>&1
timeout 600 python tools/prof_workload.py config2:1 $O/fold_single_tile kt,sq1,fetch,write >> $O/prof.log 2>&1
timeout 600 python tools/prof_workload.py config2:64 $O/fold_64_tiles kt,sq1,fetch,write >> $O/prof.log 2>&1
timeout 600 python tools/prof_workload.py raster_2x:256 $O/raster_2x kt,sq1,fetch,write >> $O/prof.log 2>&1
timeout 600 python tools/prof_workload.py composite $O/composite kt,sq1,fetch,write >> $O/prof.log 2>&1
OSMT_TIME_BIG=1 timeout 600 python tools/time_variants.py base r4 > $O/stage_times.txt 2>&1; cat $O/stage_times.txt
timeout 300 bash tools/worker_bench.sh 1 4 16 32 > $O/worker_bench.txt 2>&1; cat $O/worker_bench.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
bash tools/bench_label_variants.sh > $O/labels_kernel_trace.txt 2>&1; tail -4 $O/labels_kernel_trace.txt
timeout 120 python tools/prof_single_tile.py > $O/single_tile.txt 2>&1; tail -1 $O/single_tile.txt
timeout 260 python tools/fuzz_parity.py 150 5201 > $O/fuzz_areas.txt 2>&1; tail -1 $O/fuzz_areas.txt
timeout 260 python tools/fuzz_parity.py 150 5202 labels > $O/fuzz_labels.txt 2>&1; tail -1 $O/fuzz_labels.txt
