#!/bin/bash
# Round-5 run E: one 48-byte record per virtual segment (k_opinfo -> k_prebin) instead of six arrays: 'base' (AoS) against 'head' (the closing run's library)
TAG=${1:-r05_e}
O=gpurun_out/$TAG; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity_ops.py tests/test_gpu_empty_tiles.py tests/test_gpu_parity_tiles.py tests/test_gpu_fullsize_and_errors.py tests/test_reference_golden_patches.py -m gpu -q --timeout=300 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
OSMT_TIME_BIG=1 timeout 900 python tools/time_variants.py base head base head > $O/stage_times.txt 2>&1; cat $O/stage_times.txt
