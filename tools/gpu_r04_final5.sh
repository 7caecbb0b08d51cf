#!/bin/bash
# Round-4 last run (non-temporal framebuffer stores): GPU suite, bench line, config-2 trace + counters, stage times against round 3, a fuzz minute.
TAG=${1:-r04_final5}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err
timeout 600 python tools/prof_workload.py config2 $O/config2 kt,sq1,sq2,fetch,write > $O/prof.log 2>&1
OSMT_TIME_BIG=1 timeout 300 python tools/time_variants.py base r3 > $O/stage_times.txt 2>&1; cat $O/stage_times.txt
timeout 100 python tools/fuzz_parity.py 50 4301 > $O/fuzz_areas.txt 2>&1; tail -1 $O/fuzz_areas.txt
