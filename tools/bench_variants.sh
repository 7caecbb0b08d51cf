#!/bin/bash
# usage: tools/bench_variants.sh "<bench args>" v1 v2 ...   (variants = libosmtile_<v>.so; "base" = libosmtile.so)
ARGS="$1"; shift
for v in "$@"; do
  if [ "$v" = base ]; then L=osm_renderer_amd/libosmtile.so; else L=osm_renderer_amd/libosmtile_$v.so; fi
  OSMT_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-composite $ARGS 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$ARGS', round(d['value']), round(d['roofline']['avg_launch_ms'],3))"
done
