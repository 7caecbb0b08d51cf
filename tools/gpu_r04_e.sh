#!/bin/bash
# round-4 run e: PNG begin/end, list-boundary parity tests, fuzz with image fills + shared rings, the bench line
O=gpurun_out/r04_e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_png_device.py tests/test_gpu_parity_ops.py tests/test_gpu_worker.py -x -q > $O/pytest_new.log 2>&1; echo "pytest rc $?" >> $O/pytest_new.log; tail -5 $O/pytest_new.log
timeout 200 python tools/fuzz_parity.py 100 77 > $O/fuzz_areas.txt 2>&1; tail -3 $O/fuzz_areas.txt
timeout 200 python tools/fuzz_parity.py 80 78 labels > $O/fuzz_labels.txt 2>&1; tail -3 $O/fuzz_labels.txt
timeout 900 python bench.py --no-pmc > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_e/bench.json'))
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,(dict,list))})
e=d.get('end_to_end',{})
print({k:v for k,v in e.items() if not isinstance(v,(dict,list))})
print(e.get('worker_entry',{}).get('cases'))
print(e.get('latency',{}).get('cases',{}).get('batch1_workers1'), e.get('latency',{}).get('cases',{}).get('batch1_workers16'))
PY
