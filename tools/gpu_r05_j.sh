#!/bin/bash
# Round-5 run J: two consecutive stroke ops walked in ONE pass into two alpha planes (OSMT_V_PAIR; 14 KB of LDS per wave: 11 waves per CU)
TAG=${1:-r05_j}
O=gpurun_out/$TAG; mkdir -p $O
( OSMT_LIB=$PWD/osm_renderer_amd/libosmtile_pair.so timeout 900 python -m pytest tests/test_gpu_parity_ops.py tests/test_gpu_empty_tiles.py tests/test_gpu_parity_tiles.py tests/test_gpu_fullsize_and_errors.py tests/test_reference_golden_patches.py tests/test_gpu_labels.py -m gpu -q --timeout=300 ) > $O/pytest_pair.log 2>&1; echo "pytest rc $?" >> $O/pytest_pair.log; tail -5 $O/pytest_pair.log
OSMT_TIME_BIG=1 timeout 900 python tools/time_variants.py base pair pair4 base pair > $O/stage_times.txt 2>&1; cat $O/stage_times.txt
OSMT_LIB=$PWD/osm_renderer_amd/libosmtile_pair.so timeout 200 python tools/fuzz_parity.py 60 5401 > $O/fuzz_pair.txt 2>&1; tail -1 $O/fuzz_pair.txt
