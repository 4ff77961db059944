#!/bin/bash
# A/B of the MFMA filter's schedule knobs (environment) on one box: tools/stages_probe.sh c5|c3 "VAR=.. VAR=.." ...
cd $GRAFT_REPO_ROOT
CFG=$1; shift
run() { echo "== $*"; env $* timeout -k 10 200 python tools/bench_mfma.py --config $CFG 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['filter_kernels_ms_per_batch'], d['ms_per_batch'], d['roofline']['frac'], d['parity'][:12])"; }
run X=1
for s in "$@"; do run $s; done
run X=1
