#!/bin/bash
# k_label_cover time (rocprofv3 kernel trace) for the main library and every libosmtile_labl*.so diagnostic variant
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for lib in osm_renderer_amd/libosmtile.so osm_renderer_amd/libosmtile_labl*.so; do
  [ -f "$lib" ] || continue
  n=$(basename $lib .so)
  rm -rf /tmp/lv_$n
  OSMT_LIB=$PWD/$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/lv_$n -o lv -- python tools/bench_labels.py 1024 24 5 > /tmp/lv_$n.json 2>/dev/null
  echo "== $n $(python -c "import json;r=json.load(open('/tmp/lv_$n.json'));print('label_pass_ms',r['label_pass_ms'])")"
  python tools/rocpd_summary.py $(find /tmp/lv_$n -name '*.db' | head -1) | grep -E "k_label|k_raster" | head -6 | cut -c1-110
done
