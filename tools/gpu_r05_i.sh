#!/bin/bash
# Round-5 run I: do small requests get faster when the runtime's copies run as shader blits instead of on the SDMA engines (HSA_ENABLE_SDMA=0)?
TAG=${1:-r05_i}
O=gpurun_out/$TAG; mkdir -p $O
{ echo "# HSA_ENABLE_SDMA=0, pageable caller buffers"; HSA_ENABLE_SDMA=0 timeout 300 bash tools/worker_bench.sh 1 4 16 32
  echo "# HSA_ENABLE_SDMA=0, OSMT_BENCH_PINNED=1"; HSA_ENABLE_SDMA=0 OSMT_BENCH_PINNED=1 timeout 300 bash tools/worker_bench.sh 1 4 16 32
  echo "# HSA_ENABLE_SDMA=0, OSMT_BENCH_PINNED=1, OSMT_ZERO_COPY_TILES=0"; HSA_ENABLE_SDMA=0 OSMT_ZERO_COPY_TILES=0 OSMT_BENCH_PINNED=1 timeout 300 bash tools/worker_bench.sh 1 4 16; } > $O/worker_sdma_off.txt 2>&1; cat $O/worker_sdma_off.txt
