#!/bin/bash
# Times k_raster (config 2, 1024 tiles) for the main library and every libosmtile_<variant>.so built with
# osm_renderer_amd.build.build_variant (diagnostic -D switches: OSMT_ABL ablations, OSMT_V_* tuning knobs).
# Ablated variants render WRONG pixels on purpose; only their kernel time is of interest.
cd "$(dirname "$0")/.."
for lib in osm_renderer_amd/libosmtile.so osm_renderer_amd/libosmtile_*.so; do
  [ -f "$lib" ] || continue
  OSMT_LIB=$PWD/$lib timeout 120 python bench.py --no-extra --no-pmc --no-cpu-baseline --no-labels --no-png --no-composite --steps 10 --warmup 2 2>/dev/null \
    | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$lib', 'k_raster_ms', round(r['roofline']['avg_launch_ms'],4), 'step_ms', round(r['ms_per_step'],4))"
done
