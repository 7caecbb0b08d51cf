#!/bin/bash
# Round-5 run D: round 4's kernels + empty-tile fix + FILL_RMAX 384 + compact stroke constants + k_sublist preloads ('base') against round 4's library ('r4')
TAG=${1:-r05_d}
O=gpurun_out/$TAG; mkdir -p $O
( echo "HEAD $(cat .git_head 2>/dev/null)"; timeout 900 python -m pytest tests/test_gpu_parity_ops.py tests/test_gpu_empty_tiles.py tests/test_gpu_parity_tiles.py tests/test_gpu_fullsize_and_errors.py tests/test_gpu_labels.py tests/test_reference_golden_patches.py -m gpu -q --timeout=300 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
OSMT_TIME_BIG=1 timeout 900 python tools/time_variants.py base r4 base r4 > $O/stage_times.txt 2>&1; cat $O/stage_times.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err
python -c "
import json;d=json.load(open('$O/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','k_raster_ms','config5_tiles_per_s','raster_2x_tiles_per_s','label_pass_ms','raster_issue_frac','worker16_tiles_per_s','png_files_tiles_per_s') if k in d})
print(d['one_batch_at_a_time']); print(d['config5']); print({k.split('(')[0][-12:]:(round(v['fetch_kb']),round(v['write_kb']),round(v['avg_us'],1)) for k,v in d['pmc']['all_kernels'].items()})"
