#!/bin/bash
# label_pass_ms (bench.py label leg) for the main library and every libosmtile_labl*.so diagnostic variant
cd "$(dirname "$0")/.."
for lib in osm_renderer_amd/libosmtile.so osm_renderer_amd/libosmtile_labl*.so; do
  [ -f "$lib" ] || continue
  OSMT_LIB=$PWD/$lib timeout 200 python bench.py --no-cpu-baseline --no-pmc --no-png --no-extra --no-composite --steps 5 2>/dev/null \
    | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$lib', 'label_pass_ms', round(r['label_pass']['label_pass_ms'],3))"
done
