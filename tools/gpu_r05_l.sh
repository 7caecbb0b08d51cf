#!/bin/bash
# Round-5 run L: where the wall clock of one 1024-tile osmt_render_batch_png call goes (host phases via OSMT_TRACE_UPLOAD)
TAG=${1:-r05_l}
O=gpurun_out/$TAG; mkdir -p $O
OSMT_POISON_ALLOC=0 timeout 300 python tools/bench_e2e_breakdown.py > $O/e2e_breakdown.txt 2>&1; cat $O/e2e_breakdown.txt
OSMT_POISON_ALLOC=0 OSMT_TRACE_UPLOAD=1 timeout 300 python tools/bench_png_begin_end.py 1024 8 > $O/png_trace.txt 2>&1; grep -c "osmt upload" $O/png_trace.txt; grep "osmt upload" $O/png_trace.txt | tail -6; grep -v "osmt upload" $O/png_trace.txt
