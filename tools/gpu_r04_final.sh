#!/bin/bash
# Round-4 evidence run: full GPU suite, the bench line, kernel traces + counter passes of configs 2 / 5 / @2x / composite / labels,
# native worker bench, stage times vs the round-3 kernels, fuzz.
TAG=${1:-r04_final}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err
timeout 600 python tools/prof_workload.py config2 $O/config2 kt,sq1,sq2,fetch,write > $O/prof.log 2>&1
timeout 900 python tools/prof_workload.py config5:256 $O/config5 kt,sq1,fetch,write >> $O/prof.log 2>&1
timeout 600 python tools/prof_workload.py raster_2x:256 $O/raster_2x kt,sq1,fetch,write >> $O/prof.log 2>&1
timeout 600 python tools/prof_workload.py composite $O/composite kt,sq1,fetch,write >> $O/prof.log 2>&1
OSMT_TIME_BIG=1 timeout 600 python tools/time_variants.py base r3 > $O/stage_times.txt 2>&1; cat $O/stage_times.txt
timeout 300 bash tools/worker_bench.sh 1 4 16 32 > $O/worker_bench.txt 2>&1; cat $O/worker_bench.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
bash tools/bench_label_variants.sh > $O/labels_kernel_trace.txt 2>&1; tail -4 $O/labels_kernel_trace.txt
timeout 250 python tools/fuzz_parity.py 150 4101 > $O/fuzz_areas.txt 2>&1; tail -1 $O/fuzz_areas.txt
timeout 250 python tools/fuzz_parity.py 150 4102 labels > $O/fuzz_labels.txt 2>&1; tail -1 $O/fuzz_labels.txt
timeout 120 python tools/prof_single_tile.py > $O/single_tile.txt 2>&1; tail -3 $O/single_tile.txt
