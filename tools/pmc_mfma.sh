#!/bin/bash
# PMC passes over the MFMA batch filter (one counter set per rocprofv3 run; --pmc alone, no tracing).
# usage: tools/pmc_mfma.sh <tag> [extra env assignments...]   -> gpurun_out/pmc_<tag>/<set>/
set -o pipefail
export TMPDIR=/tmp
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
i=0
for S in "${SETS[@]}"; do
  rocprofv3 --pmc $S -d $OUT/s$i -o run -- python3 tools/bench_mfma.py --config ${CFG:-c5} --rows ${ROWS:-4000000} --nq ${NQ:-1024} --reps 2 --check 4 > $OUT/s$i.json 2> $OUT/s$i.err || { echo "set $i failed"; tail -3 $OUT/s$i.err; }
  i=$((i+1))
done
python3 - <<PY
import sqlite3, glob, re, collections
for db in sorted(glob.glob("$OUT/s*/run_results.db")):
    c = sqlite3.connect(db)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, counter, value in c.execute("select kernel_name, counter_name, value from counters_collection"):
        if "mfma" not in name: continue
        m = re.search(r"(k_mfma_\w+<[^>]*>)", name)
        a = agg[(m.group(1) if m else name[:60], counter)]
        a[0] += 1; a[1] += value
    for (k, cn), a in sorted(agg.items()):
        print(f"{k:40s} {cn:28s} n={a[0]:4d} avg={a[1]/a[0]:.4g}")
PY
