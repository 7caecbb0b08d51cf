#!/bin/bash
# usage: tools/gpu_quick.sh <tag> "<pytest args or empty>" "<variants for time_variants.py or empty>" [extra command]
TAG=$1; PYT=$2; VARS=$3; EXTRA=$4
O=gpurun_out/$TAG; mkdir -p $O
if [ -n "$PYT" ]; then timeout 1200 python -m pytest $PYT > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -15 $O/pytest.log; fi
if [ -n "$VARS" ]; then timeout 900 python tools/time_variants.py $VARS > $O/variants.txt 2>&1; cat $O/variants.txt; fi
if [ -n "$EXTRA" ]; then bash -c "$EXTRA" > $O/extra.log 2>&1; tail -40 $O/extra.log; fi
