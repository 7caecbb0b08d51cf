#!/bin/bash
# GPU run r03_a: full -m gpu suite, baseline stage times (+ the 32x8 sub-tile variant), profiles of config 5 / @2x / config 2
O=gpurun_out/r03_a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 400 python tools/time_variants.py base subh8 > $O/variants.txt 2>&1
timeout 500 python tools/prof_workload.py config5:64 $O/config5 kt,sq1,fetch,write > $O/prof.log 2>&1
timeout 300 python tools/prof_workload.py raster_2x:256 $O/raster_2x kt,sq1 >> $O/prof.log 2>&1
timeout 400 python tools/prof_workload.py config2 $O/config2 kt,sq1,sq2 >> $O/prof.log 2>&1
tail -3 $O/pytest.log; cat $O/variants.txt
