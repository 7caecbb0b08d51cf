#!/bin/bash
# does the buffer cache limit (32 GB) explain why bench.py's one-call PNG leg is slower than the same call measured alone?
O=gpurun_out/${1:-r05_n}; mkdir -p $O
OSMT_CACHE_GB=200 timeout 900 python bench.py > $O/bench_cache200.json 2> $O/bench.err; echo "bench rc $?"
python -c "
import json;d=json.load(open('$O/bench_cache200.json'));e=d['end_to_end']
print('cache 200 GB', {k:round(v) for k,v in e.items() if isinstance(v,(int,float))})"
timeout 900 python bench.py --no-extra > /dev/null 2>&1
OSMT_POISON_ALLOC=0 timeout 300 python tools/bench_png_begin_end.py 1024 8 2>&1 | grep -v amdgpu
