#!/bin/bash
# Round-5 closing run, second part, after the host-side changes (zero-copy output for small requests into pinned memory, poison fills on a
# priority stream; the kernels are those of gpu_r05_final.sh): full GPU suite, bench line, native worker bench, one-tile latency, smoke, fuzz.
TAG=${1:-r05_final2}
O=gpurun_out/$TAG; mkdir -p $O
( echo "HEAD $(cat .git_head 2>/dev/null)  (pytest -m gpu, OSMT_POISON_ALLOC=1 via tests/conftest.py)"; timeout 1500 python -m pytest tests -m gpu -q --timeout=600 --durations=6 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -14 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err
{ echo "# pageable caller buffers"; timeout 300 bash tools/worker_bench.sh 1 4 16 32; echo "# OSMT_BENCH_PINNED=1"; OSMT_BENCH_PINNED=1 timeout 300 bash tools/worker_bench.sh 1 4 16 32; } > $O/worker_bench.txt 2>&1; cat $O/worker_bench.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 120 python tools/prof_single_tile.py > $O/single_tile.txt 2>&1; tail -1 $O/single_tile.txt
timeout 200 python tools/fuzz_parity.py 100 5301 > $O/fuzz_areas.txt 2>&1; tail -1 $O/fuzz_areas.txt
timeout 200 python tools/fuzz_parity.py 100 5302 labels > $O/fuzz_labels.txt 2>&1; tail -1 $O/fuzz_labels.txt
