#!/bin/bash
# builds and runs tools/worker_bench.cpp against the in-tree libosmtile.so (run on a GPU box); args: thread counts
set -e
cd "$(dirname "$0")/.."
mkdir -p tests/_build
g++ -O2 -std=c++17 -pthread -o tests/_build/worker_bench tools/worker_bench.cpp -Losm_renderer_amd -losmtile \
    -Wl,-rpath,$PWD/osm_renderer_amd -Wl,-rpath-link,/opt/rocm/lib
TORCH_LIB=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
LD_LIBRARY_PATH=$TORCH_LIB:/opt/rocm/lib:$LD_LIBRARY_PATH tests/_build/worker_bench "$@"
