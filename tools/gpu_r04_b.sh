#!/bin/bash
# round-4 run b: worker entry tests + bench, ablation ladder of the new k_raster, work counters, issue counters
O=gpurun_out/r04_b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_worker.py -x -q > $O/pytest_worker.log 2>&1; echo "pytest rc $?" >> $O/pytest_worker.log; tail -5 $O/pytest_worker.log
timeout 300 bash tools/worker_bench.sh 1 4 16 > $O/worker_bench.txt 2>&1; cat $O/worker_bench.txt
timeout 600 python tools/time_variants.py base abl1 abl2 abl3 abl4 abl6 > $O/ablation.txt 2>&1; cat $O/ablation.txt
OSMT_LIB=$PWD/osm_renderer_amd/libosmtile_dbg.so timeout 120 python tools/dbg_counts.py config2 > $O/dbg_counts.txt 2>&1; cat $O/dbg_counts.txt
timeout 400 python tools/prof_workload.py config2 $O/config2 kt,sq1,sq2 > $O/prof.log 2>&1; grep -A3 "k_raster" $O/config2_pmc.txt | head -40
