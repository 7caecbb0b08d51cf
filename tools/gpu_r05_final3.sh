#!/bin/bash
# Round-5 closing run, third part, on HEAD after the split upload (host side only; kernels = gpu_r05_final.sh): full GPU suite, bench line, smoke, a fuzz minute of each kind
TAG=${1:-r05_final3}
O=gpurun_out/$TAG; mkdir -p $O
( echo "HEAD $(cat .git_head 2>/dev/null)  (pytest -m gpu, OSMT_POISON_ALLOC=1 via tests/conftest.py)"; timeout 1500 python -m pytest tests -m gpu -q --timeout=600 --durations=6 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -14 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 200 python tools/fuzz_parity.py 60 5601 > $O/fuzz_areas.txt 2>&1; tail -1 $O/fuzz_areas.txt
timeout 200 python tools/fuzz_parity.py 60 5602 labels > $O/fuzz_labels.txt 2>&1; tail -1 $O/fuzz_labels.txt
