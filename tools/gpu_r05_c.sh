#!/bin/bash
# Round-5 run C: fixed-slot list heads + pre-pass fixes; variants: nofix (FIXK=0), p384 / p5 (k_prebin occupancy), r4 (round 4's library)
TAG=${1:-r05_c}
O=gpurun_out/$TAG; mkdir -p $O
( echo "HEAD $(cat .git_head 2>/dev/null)"; timeout 900 python -m pytest tests/test_gpu_parity_ops.py tests/test_gpu_empty_tiles.py tests/test_gpu_parity_tiles.py tests/test_gpu_fullsize_and_errors.py tests/test_gpu_labels.py tests/test_gpu_worker.py tests/test_reference_golden_patches.py -m gpu -q --timeout=300 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -8 $O/pytest.log
OSMT_TIME_BIG=1 timeout 900 python tools/time_variants.py base nofix p384 p5 r4 base > $O/stage_times.txt 2>&1; cat $O/stage_times.txt
timeout 200 python tools/fuzz_parity.py 90 5101 > $O/fuzz_areas.txt 2>&1; tail -2 $O/fuzz_areas.txt
