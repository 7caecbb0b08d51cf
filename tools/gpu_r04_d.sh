#!/bin/bash
# round-4 run d: flattened pre-pass load chains; full suite; stage times; batches in flight; multi-GPU host share
O=gpurun_out/r04_d; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
OSMT_TIME_BIG=1 timeout 600 python tools/time_variants.py base r3 > $O/stage_times.txt 2>&1; cat $O/stage_times.txt
for s in 1 2 3 4; do timeout 300 python bench.py --streams $s --no-extra --no-cpu-baseline --no-pmc --no-composite --no-labels --no-png 2>$O/bench_s$s.err | tail -1 > $O/bench_s$s.json; python -c "
import json,sys; d=json.load(open('$O/bench_s$s.json')); print('streams', $s, 'value', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'seq', d.get('one_batch_at_a_time',{}).get('ms_per_step'), 'k_raster', round(d['roofline']['avg_launch_ms'],4))"; done
timeout 600 python tools/bench_multi_host.py 10000 1 2 4 8 > $O/multi_host.txt 2>&1; cat $O/multi_host.txt
