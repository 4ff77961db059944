#!/bin/bash
# kernel-trace timing of diagnostic builds (their results are wrong on purpose; only k_mfma_* durations are read)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/variants; mkdir -p $OUT
for T in "$@"; do
  if [ $T = base ]; then unset VL_LIB_PATH; else export VL_LIB_PATH=$GRAFT_REPO_ROOT/vectorlite_amd/libvl_$T.so; fi
  rm -rf $OUT/$T
  rocprofv3 --kernel-trace -d $OUT/$T -o run -- python3 tools/bench_mfma.py --config ${CFG:-c5} --rows ${ROWS:-4000000} --nq ${NQ:-1024} --reps 2 --check 1 > $OUT/$T.json 2> $OUT/$T.err
  python3 - <<PY
import sys
import sqlite3, re, collections
c = sqlite3.connect("$OUT/$T/run_results.db")
agg = collections.defaultdict(list)
for name, dur in c.execute("select name, duration from kernels where name like '%k_mfma_%' order by start"):
    m = re.search(r"k_mfma_\w+<[^>]*>", name) or re.search(r"k_mfma_\w+?I\w+?E", name)
    agg[m.group(0) if m else name[:70]].append(dur)
for k, v in agg.items():
    per_batch = len(v) // 4 if len(v) >= 4 else 1   # 4 batches per run (2 warm + 2 timed)
    last = v[-per_batch:]
    print("$T", k, "launches/batch", per_batch, "sum_us", round(sum(last) / 1e3, 1), "each_us", [round(x / 1e3) for x in last])
PY
done
