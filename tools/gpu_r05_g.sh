#!/bin/bash
# Round-5 run G: k_raster writing a small request's pixels straight into PINNED caller memory (zero copy) against the copy path
TAG=${1:-r05_g}
O=gpurun_out/$TAG; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_worker.py tests/test_gpu_empty_tiles.py tests/test_gpu_fullsize_and_errors.py tests/test_gpu_threads.py tests/test_gpu_labels.py tests/test_gpu_host_mirror.py -m gpu -q --timeout=300 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
{ echo "# pinned caller buffers, zero copy (default: up to 8 tiles)"; OSMT_BENCH_PINNED=1 timeout 200 bash tools/worker_bench.sh 1 4 16 32
  echo "# pinned caller buffers, OSMT_ZERO_COPY_TILES=0 (device framebuffer + asynchronous copy)"; OSMT_ZERO_COPY_TILES=0 OSMT_BENCH_PINNED=1 timeout 200 bash tools/worker_bench.sh 1 4 16 32
  echo "# pageable caller buffers (zero copy never applies to the caller's buffer; the worker entry's pinned staging takes it for groups of up to 8 tiles)"; timeout 200 bash tools/worker_bench.sh 1 4 16 32
  echo "# pageable caller buffers, OSMT_ZERO_COPY_TILES=0"; OSMT_ZERO_COPY_TILES=0 timeout 200 bash tools/worker_bench.sh 1 4 16 32; } > $O/worker_zero_copy.txt 2>&1; cat $O/worker_zero_copy.txt
