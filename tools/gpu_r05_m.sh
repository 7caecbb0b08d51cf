#!/bin/bash
# Round-5 run M: split upload (a helper thread copies the caller's arrays while the calling thread validates and builds the tables)
TAG=${1:-r05_m}
O=gpurun_out/$TAG; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_fullsize_and_errors.py tests/test_gpu_empty_tiles.py tests/test_gpu_multi.py tests/test_gpu_png_device.py tests/test_gpu_worker.py tests/test_gpu_zero_copy.py -m gpu -q --timeout=300 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
OSMT_POISON_ALLOC=0 timeout 300 python tools/bench_e2e_breakdown.py > $O/e2e_breakdown.txt 2>&1; cat $O/e2e_breakdown.txt
OSMT_POISON_ALLOC=0 OSMT_TRACE_UPLOAD=1 timeout 300 python tools/bench_png_begin_end.py 1024 8 > $O/png_trace.txt 2>&1; grep "osmt upload" $O/png_trace.txt | tail -3; grep -v "osmt upload" $O/png_trace.txt
