#!/bin/bash
# per-kernel timeline of one batch of the MFMA filter (last batch of the run)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/trace_$1; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o run -- python3 tools/bench_mfma.py --config $1 --reps 2 --check 1 > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import sqlite3, re
c = sqlite3.connect("$OUT/run_results.db")
rows = c.execute("select name, start, end, duration, grid_x, grid_y, workgroup_x from kernels order by start").fetchall()
# last occurrence of k_queries_bf16 starts the last batch pass
idx = max(i for i, r in enumerate(rows) if "k_queries_bf16" in r[0])
t0 = rows[idx][1]
for r in rows[idx: idx + 12]:
    nm = re.sub(r"\(anonymous namespace\)::", "", r[0])
    m = re.search(r"\d+(k_\w+?)I((?:Li\d+E)+)E", nm)
    nm = (m.group(1) + "<" + ",".join(re.findall(r"Li(\d+)E", m.group(2))) + ">") if m else nm.split("(")[0][-40:]
    print(f"{(r[1]-t0)/1e3:9.1f} us  +{r[3]/1e3:8.1f} us  grid {r[4]//max(r[6],1)}x{r[5]}  {nm}")
PY
