#!/bin/bash
# Round-5 run H: (1) the scene-isolation test six times in fresh processes + twice inside its file (poison fills on a highest-priority stream);
# (2) zero-copy threshold 8 / 16 / 32 tiles through the native worker bench (pageable caller buffers: the worker entry's pinned staging takes it)
TAG=${1:-r05_h}
O=gpurun_out/$TAG; mkdir -p $O
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest "tests/test_gpu_fullsize_and_errors.py::test_worker_threads_with_their_own_scenes_do_not_wait_for_each_other" -m gpu -q 2>&1 | tail -1; done > $O/isolation_test.txt 2>&1
for i in 1 2; do timeout 600 python -m pytest tests/test_gpu_worker.py tests/test_gpu_fullsize_and_errors.py tests/test_gpu_threads.py -m gpu -q 2>&1 | tail -1; done >> $O/isolation_test.txt 2>&1
cat $O/isolation_test.txt
for z in 8 16 32; do echo "# OSMT_ZERO_COPY_TILES=$z, pageable caller buffers"; OSMT_ZERO_COPY_TILES=$z timeout 200 bash tools/worker_bench.sh 16 32 2>&1 | grep worker_render; done > $O/zero_copy_threshold.txt 2>&1; cat $O/zero_copy_threshold.txt
