#!/bin/bash
# Round-5 closing run, fourth part, on HEAD after k_opinfo's shared arena reservations: full GPU suite, bench line, traces + counters of config 2 and config 5 (256), stage times against round 4, smoke, a fuzz minute
TAG=${1:-r05_final4}
O=gpurun_out/$TAG; mkdir -p $O
( echo "HEAD $(cat .git_head 2>/dev/null)  (pytest -m gpu -x, OSMT_POISON_ALLOC=1 via tests/conftest.py)"; timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -1 $O/bench.err
timeout 600 python tools/prof_workload.py config2 $O/config2 kt,sq1,fetch,write > $O/prof.log 2>&1
timeout 900 python tools/prof_workload.py config5:256 $O/config5 kt,write >> $O/prof.log 2>&1
OSMT_TIME_BIG=1 timeout 600 python tools/time_variants.py base r4 > $O/stage_times.txt 2>&1; cat $O/stage_times.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 200 python tools/fuzz_parity.py 60 5701 > $O/fuzz_areas.txt 2>&1; tail -1 $O/fuzz_areas.txt
