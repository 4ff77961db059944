cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/bench_mfma.py --config c5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['filter_kernels_ms_per_batch'], d['ms_per_batch'], d['roofline']['frac'], d['parity'][:12])"; }
run VL_MFMA_STAGES=3
run VL_MFMA_STAGES=4
run VL_MFMA_STAGES=4 VL_MFMA_STAGE1=1 VL_MFMA_STAGE2=2 VL_MFMA_STAGE3=6
run VL_MFMA_STAGES=4 VL_MFMA_STAGE1=1 VL_MFMA_STAGE2=4 VL_MFMA_STAGE3=9
run VL_MFMA_STAGES=3 VL_MFMA_STAGE1=1 VL_MFMA_STAGE2=4
run VL_MFMA_STAGES=3
