#!/bin/bash
# Round-4 closing run on the final library (RGB8 packed in k_raster, two worker groups in flight): GPU suite, bench line, config-2 trace +
# counters, worker bench, single-tile latency, smoke, a short fuzz of each kind.  (Traces of config 5 / @2x / composite / labels: gpu_r04_final.sh.)
TAG=${1:-r04_final3}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err
timeout 600 python tools/prof_workload.py config2 $O/config2 kt,sq1,sq2,fetch,write > $O/prof.log 2>&1
timeout 300 bash tools/worker_bench.sh 1 4 16 32 > $O/worker_bench.txt 2>&1; cat $O/worker_bench.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 120 python tools/prof_single_tile.py > $O/single_tile.txt 2>&1; tail -1 $O/single_tile.txt
timeout 150 python tools/fuzz_parity.py 90 4201 > $O/fuzz_areas.txt 2>&1; tail -1 $O/fuzz_areas.txt
timeout 150 python tools/fuzz_parity.py 90 4202 labels > $O/fuzz_labels.txt 2>&1; tail -1 $O/fuzz_labels.txt
