#!/bin/bash
# Round-5 first run: the full GPU suite under poison (tests/conftest.py), then the bench line.
TAG=${1:-r05_a}
O=gpurun_out/$TAG; mkdir -p $O
( echo "HEAD $(cat .git_head 2>/dev/null)"; timeout 1500 python -m pytest tests -m gpu -q --durations=15 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -30 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err; cat $O/bench.json | head -c 3000
