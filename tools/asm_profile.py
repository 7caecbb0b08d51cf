#!/usr/bin/env python
"""Static instruction profile of one kernel of osmt_kernels.hip (CPU only; hipcc cross-compiles gfx950).

    python tools/asm_profile.py <kernel substring> [-D FLAG ...] [--src FILE] [--by-block]

Compiles the file device-only with line tables, then counts the instructions of the kernel whose mangled name contains
the substring, per source line (file:line of the innermost .loc) and per class (VALU / SALU / LDS / VMEM / SMEM /
branch).  Static counts: multiply by the loop trip counts (tools/dbg_counts.py) for a dynamic estimate.  --by-block
lists basic blocks in program order instead, with the source lines each block spans.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "BR"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store") or op.startswith("s_dcache"):
        return "SMEM"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "WAIT"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"):
        return "VMEM"
    if op.startswith("v_"):
        return "VALU"
    return "OTHER"


def main():
    args = sys.argv[1:]
    needle = args[0]
    defs, src, by_block = [], os.path.join(ROOT, "osm_renderer_amd", "csrc", "osmt_kernels.hip"), False
    i = 1
    while i < len(args):
        if args[i] == "-D":
            defs.append("-D" + args[i + 1])
            i += 2
        elif args[i] == "--src":
            src = args[i + 1]
            i += 2
        elif args[i] == "--by-block":
            by_block = True
            i += 1
        else:
            i += 1
    out = tempfile.mktemp(suffix=".s", dir="/tmp")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only",
           "-gline-tables-only", "-S", "-o", out, src] + defs
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    files = {}
    lines = open(out).read().splitlines()
    os.unlink(out)
    in_k = False
    cur = ("?", 0)
    per_line = collections.defaultdict(collections.Counter)
    blocks = []  # (label, Counter, set(lines))
    total = collections.Counter()
    for ln in lines:
        s = ln.strip()
        m = re.match(r"\.file\s+(\d+)\s+\"([^\"]*)\"\s+\"([^\"]*)\"", s)
        if m:
            files[int(m.group(1))] = m.group(3)
            continue
        m = re.match(r"\.file\s+(\d+)\s+\"([^\"]*)\"", s)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(2))
            continue
        if re.match(r"^_Z\w*:", ln):
            in_k = needle in ln
            if in_k:
                blocks.append((ln.rstrip(":"), collections.Counter(), set()))
            continue
        if not in_k:
            continue
        if s.startswith(".Lfunc_end"):
            in_k = False
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (os.path.basename(files.get(int(m.group(1)), "?")), int(m.group(2)))
            continue
        if re.match(r"^\.LBB\d+_\d+:", s):
            blocks.append((s.split(":")[0], collections.Counter(), set()))
            continue
        if not s or s.startswith(".") or s.startswith(";") or s.startswith("//"):
            continue
        op = s.split()[0]
        c = classify(op)
        per_line[cur][c] += 1
        total[c] += 1
        blocks[-1][1][c] += 1
        blocks[-1][2].add(cur)
    cols = ["VALU", "SALU", "LDS", "VMEM", "SMEM", "BR", "WAIT"]
    print("# static instruction counts; total:", dict(total), "sum", sum(total.values()))
    if by_block:
        for label, cnt, lns in blocks:
            if not sum(cnt.values()):
                continue
            srcs = sorted(lns)
            span = ", ".join(f"{f}:{l}" for f, l in srcs[:6]) + (" ..." if len(srcs) > 6 else "")
            print(f"{label:14s} " + " ".join(f"{c}={cnt[c]:<4d}" for c in cols if cnt[c]) + f"   [{span}]")
        return
    print(f"{'file:line':28s} " + " ".join(f"{c:>5s}" for c in cols))
    for key in sorted(per_line):
        cnt = per_line[key]
        print(f"{key[0] + ':' + str(key[1]):28s} " + " ".join(f"{cnt[c]:5d}" for c in cols))


if __name__ == "__main__":
    main()
