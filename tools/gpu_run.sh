#!/bin/bash
# ONE script for every GPU-box run of a round (replaces the 33 one-off tools/gpu_rNN_*.sh of rounds 3-5).  Runs from the repository
# root on the box (`gpurun -- 'bash tools/gpu_run.sh <tag> <step> [<step> ...]'`); everything lands under gpurun_out/<tag>/.
#
#   pytest[:<pytest args>]      the GPU suite (default: tests -m gpu -x -q), under OSMT_POISON_ALLOC=1 (tests/conftest.py)
#   variants:<v1,v2,...>        tools/time_variants.py over libosmtile_<v>.so ("base" = libosmtile.so); BIG=1 adds 256 config-5 tiles
#   fuzz[:<seconds>[:<seed>]]   tools/fuzz_parity.py (areas) ; fuzzlabels[:<seconds>[:<seed>]] with a label pass per tile
#   bench[:<bench.py args>]     python bench.py -> bench.json (+ the scalars the round's bars are set on)
#   prof:<workload>[@<passes>]  tools/prof_workload.py: kernel trace + counter passes (kt,sq1,sq2,sq3,fetch,write) of a bench.py --pmc-child workload ("config5:256")
#   smoke                       __graft_entry__.smoke()
#   worker                      tools/worker_bench.sh (native request threads)
#   sh:<command>                anything else, logged to extra_<n>.log
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
echo "HEAD $(cat .git_head 2>/dev/null || git rev-parse --short HEAD 2>/dev/null)" > $O/head.txt
n=0
for step in "$@"; do
  n=$((n+1))
  kind=${step%%:*}; arg=""; [ "$step" != "$kind" ] && arg=${step#*:}
  case $kind in
    pytest)
      ( cat $O/head.txt; timeout 2400 python -m pytest ${arg:-tests -m gpu -x -q --timeout=600} ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log ;;
    variants)
      OSMT_TIME_BIG=${BIG:-} timeout 1200 python tools/time_variants.py ${arg//,/ } > $O/variants_$n.txt 2>&1; cat $O/variants_$n.txt ;;
    fuzz|fuzzlabels)
      secs=${arg%%:*}; seed=${arg#*:}; [ -z "$arg" ] && secs=60; [ "$seed" = "$arg" ] && seed=$((6000 + n))
      extra=""; [ $kind = fuzzlabels ] && extra="labels"
      timeout $((secs + 240)) python tools/fuzz_parity.py $secs $seed $extra > $O/${kind}_$n.txt 2>&1; tail -1 $O/${kind}_$n.txt ;;
    bench)
      timeout 1500 python bench.py $arg > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err
      python - <<PY
import json
try:
    d = json.load(open("$O/bench.json"))
    keys = ("value", "ms_per_step", "k_raster_ms", "raster_issue_frac", "config5_tiles_per_s", "config5_sequential_tiles_per_s", "raster_2x_tiles_per_s",
            "label_pass_ms", "composite_hbm_frac", "png_files_tiles_per_s", "png_files_begin_end_tiles_per_s", "png_bytes_per_tile",
            "worker16_tiles_per_s", "worker1_p50_us", "sustained_tiles_per_s")
    print({k: (round(d[k], 4) if isinstance(d.get(k), float) else d.get(k)) for k in keys})
except Exception as e:
    print("no bench line:", e)
PY
      ;;
    prof)
      wl=${arg%%@*}; passes=${arg#*@}; [ "$passes" = "$arg" ] && passes="kt,sq1,fetch,write"
      timeout 1500 python tools/prof_workload.py $wl $O/${wl//:/_} $passes >> $O/prof.log 2>&1; tail -2 $O/prof.log ;;
    smoke)
      timeout 600 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt ;;
    worker)
      timeout 900 bash tools/worker_bench.sh > $O/worker_bench.txt 2>&1; tail -12 $O/worker_bench.txt ;;
    sh)
      bash -c "$arg" > $O/extra_$n.log 2>&1; tail -30 $O/extra_$n.log ;;
    *) echo "unknown step $step" ;;
  esac
done
