#!/bin/bash
# HBM traffic of the label kernels: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md), KiB per dispatch
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/plh_$ctr
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/plh_$ctr -o p -- python tools/bench_labels.py 1024 24 2 > /dev/null 2>/tmp/plh_$ctr.err || tail -3 /tmp/plh_$ctr.err
  python tools/rocpd_summary.py $(find /tmp/plh_$ctr -name '*.db' | head -1) | grep -E "$ctr" | grep -E "k_label|k_raster" | cut -c1-110
done
