#!/bin/bash
# kernel timeline of single-query searches (rocprofv3 --kernel-trace; the program itself after `--`): per kernel
# duration and the gap to the kernel before it.  usage: trace_single.sh [rows]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/trace_single; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o run -- python3 tools/trace_single.py ${1:-10000000} > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/out.txt
python3 - <<PY
import sqlite3, re
c = sqlite3.connect("$OUT/run_results.db")
rows = list(c.execute("select name, start, end from kernels order by start"))
rows = [r for r in rows if "vl" in r[0]][-12:]
prev_end = None
for name, s, e in rows:
    m = re.search(r"k_\w+", name)
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(m.group(0) if m else name[:40]):28s} dur {(e - s) / 1e3:9.1f} us   gap before {gap:8.1f} us")
    prev_end = e
PY
