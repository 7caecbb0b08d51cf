#!/bin/bash
for v in "$@"; do
  if [ "$v" = base ]; then L=osm_renderer_amd/libosmtile.so; else L=osm_renderer_amd/libosmtile_$v.so; fi
  OSMT_LIB=$PWD/$L python bench.py --no-cpu-baseline --tiles 8 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['roofline_composite']; print('$v', round(d['achieved']), round(d['frac'],4), round(d['avg_launch_ms'],4))"
done
