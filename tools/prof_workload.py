#!/usr/bin/env python
"""Kernel trace + counter passes of ONE bench workload, as text summaries for profiles/.

    python tools/prof_workload.py <workload> <out_prefix> [passes]

<workload> is a `bench.py --pmc-child` name ("config2", "config5:64", "raster_2x:256", "all"); writes
<out_prefix>_kernel_trace.txt and <out_prefix>_pmc.txt (SQ issue / wait counters, FETCH_SIZE, WRITE_SIZE — each in its
own rocprofv3 pass with only --kernel-trace beside it).  passes: comma list out of kt,sq1,sq2,sq3,fetch,write (default all
but sq2 / sq3).
"""
import contextlib
import io
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import pmc_pass, rocpd_summary  # noqa: E402


def summary(db):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rocpd_summary.main(db)
    return buf.getvalue()


def main():
    workload, prefix = sys.argv[1], sys.argv[2]
    passes = (sys.argv[3] if len(sys.argv) > 3 else "kt,sq1,fetch,write").split(",")
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    if "kt" in passes:
        out_dir = tempfile.mkdtemp(prefix="osmt_kt_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        cmd = [pmc_pass.rocprof(), "--kernel-trace", "--stats", "-d", out_dir, "-o", "kt", "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--pmc-child", workload]
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        dbs = [os.path.join(d, f) for d, _, fs in os.walk(out_dir) for f in fs if f.endswith(".db")]
        with open(prefix + "_kernel_trace.txt", "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --pmc-child {workload}\n")
            if dbs:
                f.write(summary(dbs[0]))
            else:
                f.write(f"# no database (rc {r.returncode}): {r.stderr.decode(errors='replace')[-400:]}\n")
    sets = {"sq1": pmc_pass.SQ_PASS_1, "sq2": pmc_pass.SQ_PASS_2, "sq3": pmc_pass.SQ_PASS_3, "fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"]}
    with open(prefix + "_pmc.txt", "w") as f:
        for name in passes:
            if name not in sets:
                continue
            keep = tempfile.mkdtemp(prefix="osmt_keep_", dir="/tmp")
            res = pmc_pass.run_pass(sets[name], [workload], timeout=600, keep_dir=keep)
            f.write(f"# pass {name}: rocprofv3 --kernel-trace --pmc {' '.join(sets[name])} -- python bench.py --pmc-child {workload}\n")
            if "error" in res:
                f.write(f"# error: {res['error']}\n")
                continue
            for db in sorted(os.listdir(keep)):
                f.write(summary(os.path.join(keep, db)))
            f.write("\n")


if __name__ == "__main__":
    main()
