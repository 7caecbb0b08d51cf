for m in 2 6; do
OSMT_WORKER_INFLIGHT=$m OSMT_TRACE_WORKER=1 timeout 100 bash tools/worker_bench.sh 16 2> gpurun_out/wt_$m.err | grep worker_render
python3 - $m <<'PY'
import re,collections,sys
m=sys.argv[1]
rows=[tuple(map(float,x.groups())) for x in re.finditer(r"group: (\d+) requests, (\d+) tiles, merge\+staging (\d+) us, render (\d+) us", open(f'gpurun_out/wt_{m}.err').read())]
by=collections.defaultdict(list)
for r,t,mg,rd in rows: by[int(t)].append((mg,rd))
print("inflight",m,"groups", len(rows))
for t in sorted(by):
    v=sorted(b for _,b in by[t]); mg=sorted(a for a,_ in by[t])
    print(f"tiles {t:3d}: {len(v):5d} groups, merge p50 {mg[len(mg)//2]:6.0f} max {mg[-1]:7.0f} us, render p50 {v[len(v)//2]:7.0f} p99 {v[int(len(v)*0.99)]:7.0f} max {v[-1]:7.0f} us")
PY
done
