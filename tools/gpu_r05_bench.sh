#!/bin/bash
# the bench line of HEAD (bench.py's one-call legs: best of five after two untimed calls)
O=gpurun_out/${1:-r05_bench}; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -1 $O/bench.err
python -c "
import json;d=json.load(open('$O/bench.json'));e=d['end_to_end']
print({k:d[k] for k in ('value','ms_per_step','k_raster_ms','config5_tiles_per_s','png_files_tiles_per_s','png_files_begin_end_tiles_per_s','worker16_tiles_per_s')})
print({k:round(v) for k,v in e.items() if isinstance(v,(int,float))})"
