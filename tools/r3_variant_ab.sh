#!/bin/bash
# A/B of a diagnostic build against the shipped library: config 3's shard and config 5 through tools/bench_mfma.py,
# alternating, filter time per batch (HIP events).  usage: tools/r3_variant_ab.sh <tag> [c3|c5 ...]
cd $GRAFT_REPO_ROOT
TAG=$1; shift
for C in ${@:-c3 c5}; do
  for rep in 1 2; do
    for T in base $TAG; do
      if [ $T = base ]; then unset VL_LIB_PATH; else export VL_LIB_PATH=$GRAFT_REPO_ROOT/vectorlite_amd/libvl_$T.so; fi
      python3 tools/bench_mfma.py --config $C --reps 5 --check 8 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(\"$C $T filter\", d[\"filter_kernels_ms_per_batch\"], d[\"roofline\"][\"frac\"], \"whole\", d[\"ms_per_batch\"], d[\"parity\"][:24])"
    done
  done
done
