#!/bin/bash
# Round-2 evidence for the batched MFMA filter: configs 5 and 3 -- bench JSON (roofline object), rocprofv3 kernel trace,
# and separate PMC passes (MFMA busy / waits; FETCH_SIZE).  The program itself follows `--`.  Outputs: gpurun_out/r2mf/.
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2mf
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for C in c5 c3; do
  python3 tools/bench_mfma.py --config $C --reps 5 > $OUT/${C}_bench.json 2> $OUT/${C}_bench.err || exit 1
  rocprofv3 --kernel-trace --stats -d $OUT/${C}_stats -o run -- python3 tools/bench_mfma.py --config $C --reps 3 --check 4 > $OUT/${C}_prof.json 2> $OUT/${C}_prof.err || exit 2
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/${C}_pmc1 -o run -- python3 tools/bench_mfma.py --config $C --reps 2 --check 4 > $OUT/${C}_pmc1.json 2> $OUT/${C}_pmc1.err || exit 3
  rocprofv3 --pmc FETCH_SIZE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/${C}_pmc2 -o run -- python3 tools/bench_mfma.py --config $C --reps 2 --check 4 > $OUT/${C}_pmc2.json 2> $OUT/${C}_pmc2.err || exit 4
  echo "[r2mf] $C done"
done
