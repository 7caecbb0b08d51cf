"""rocprofv3 counter passes run as CHILD processes of bench.py, so that the HBM traffic and the issue counters in
the bench line belong to the SAME run (same box, same binary) and not to an earlier profile.

Each pass is `rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --pmc-child <workload>` (counters in their
own pass, only --kernel-trace beside them; FETCH_SIZE and WRITE_SIZE in separate passes: MI355X_MICROARCH.md
"rocprofv3 PMC slots").  The child renders a few steps of the named workload and exits; the parent reads the rocpd
SQLite database the profiler leaves behind.  Everything is bounded by a timeout and every failure is reported as
{"error": ...} instead of an exception: a counter pass must never cost the bench line.
"""
import glob
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SQ_PASS_1 = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
             "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]
SQ_PASS_3 = ["SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES"]  # lane packing: thread-cycles / (64 x active cycles)
SQ_PASS_2 = ["SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_WAVES",
             "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"]


def rocprof():
    for cand in (shutil.which("rocprofv3"), "/opt/rocm/bin/rocprofv3"):
        if cand and os.path.exists(cand):
            return cand
    return None


def read_db(path):
    """{kernel_name: {"calls": n, "avg_us": .., counter: per-dispatch average}} from a rocpd database."""
    c = sqlite3.connect(path)
    out = {}
    for name, n, avg in c.execute("select name, count(*), avg(duration) from kernels group by name"):
        out[name] = {"calls": int(n), "avg_us": float(avg) / 1e3}
    try:
        rows = c.execute(
            "select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection "
            "group by kernel_name, counter_name"
        ).fetchall()
    except sqlite3.Error:
        rows = []
    for name, ctr, n, s in rows:
        out.setdefault(name, {})[ctr] = float(s) / max(int(n), 1)
        out[name].setdefault("pmc_dispatches", int(n))
    return out


def run_pass(counters, child_args, timeout=240, keep_dir=None):
    """One profiler pass; returns the per-kernel dict of read_db() or {"error": ...}."""
    exe = rocprof()
    if exe is None:
        return {"error": "rocprofv3 not found"}
    out_dir = tempfile.mkdtemp(prefix="osmt_pmc_", dir="/tmp")
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    cmd = [exe, "--kernel-trace", "--pmc", *counters, "-d", out_dir, "-o", "p", "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--pmc-child", *child_args]
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
        dbs = glob.glob(os.path.join(out_dir, "**", "*.db"), recursive=True)
        if not dbs:
            return {"error": f"no rocpd database (rc {r.returncode}): {r.stderr.decode(errors='replace')[-300:]}"}
        res = {}
        for db in dbs:
            for k, v in read_db(db).items():
                res.setdefault(k, {}).update(v)
        if keep_dir:
            os.makedirs(keep_dir, exist_ok=True)
            for db in dbs:  # one database per pass: name it after the pass's first counter
                shutil.copy(db, os.path.join(keep_dir, f"{counters[0]}_{os.path.basename(db)}"))
        return res
    except subprocess.TimeoutExpired:
        return {"error": f"rocprofv3 pass timed out after {timeout} s"}
    except Exception as e:  # noqa: BLE001 — never let a counter pass kill the bench line
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def pick(res, needle):
    """The entry of the kernel whose name contains `needle` (most calls wins), or None."""
    best = None
    for name, v in res.items():
        if needle in name and isinstance(v, dict) and (best is None or v.get("calls", 0) > best[1].get("calls", 0)):
            best = (name, v)
    return best
