#!/bin/bash
# Round-5 run O: k_opinfo with four waves per workgroup sharing ONE atomicAdd per arena cursor ('base'; ow8: eight) against one per wave ('ow1' = the closing run's kernel)
TAG=${1:-r05_o}
O=gpurun_out/$TAG; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity_ops.py tests/test_gpu_empty_tiles.py tests/test_gpu_parity_tiles.py tests/test_gpu_fullsize_and_errors.py tests/test_reference_golden_patches.py tests/test_gpu_worker.py -m gpu -q --timeout=300 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
OSMT_TIME_BIG=1 timeout 900 python tools/time_variants.py base ow1 ow8 base ow1 > $O/stage_times.txt 2>&1; cat $O/stage_times.txt
