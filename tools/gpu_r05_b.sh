#!/bin/bash
# Round-5 run B: the sorted stroke-record layout (k_raster without the key filter) + compact stroke constants:
# full GPU suite under poison, stage times against the round-4 library in the same run, the bench line.
TAG=${1:-r05_b}
O=gpurun_out/$TAG; mkdir -p $O
( echo "HEAD $(cat .git_head 2>/dev/null)"; timeout 1500 python -m pytest tests -m gpu -q --timeout=300 --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -25 $O/pytest.log
OSMT_TIME_BIG=1 timeout 600 python tools/time_variants.py base r4 base r4 > $O/stage_times.txt 2>&1; cat $O/stage_times.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err
python -c "
import json;d=json.load(open('$O/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','k_raster_ms','config5_tiles_per_s','raster_2x_tiles_per_s','label_pass_ms','raster_issue_frac','worker16_tiles_per_s','png_files_tiles_per_s') if k in d})
print(d['one_batch_at_a_time']); print({k:(v['fetch_kb'],v['write_kb'],v['avg_us']) for k,v in d['pmc']['all_kernels'].items()})"
