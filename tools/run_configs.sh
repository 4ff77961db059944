#!/usr/bin/env bash
# BASELINE.json's configurations in one go on a single MI355X (config 3: this GPU's 1/8 shard; the 8-GPU launch line
# is in tools/bench_sharded.py).  Results go to stdout; nothing here is part of the driver's contract (bench.py is).
#   bash tools/run_configs.sh 2>/dev/null | tee gpurun_out/configs.txt
set -o pipefail
cd "$(dirname "$0")/.."
echo "== headline: flat cosine N=10M d384 k10, single query (bench.py)";            python bench.py | cut -c1-600
echo "== config 2: flat cosine N=1M d384 k10, single query";                        python bench.py --rows 1000000 --no-cpu-baseline --no-checks | cut -c1-400
echo "== config 3 (one rank's shard): flat L2 1.25M x 768, batch 1024";             python tools/bench_sharded.py --rows 1250000 --dim 768 --batch 1024 --steps 10
echo "== config 4: HNSW cosine N=1M d384, ef 10 (reference) .. 128";                python tools/hnsw_eval.py --rows 1000000 --dim 384 --latent 16 --nq 2000 | tail -5
echo "== config 5: batched flat Q=4096 N=10M d384 (bf16 MFMA filter + exact finalize)"; python tools/bench_mfma.py --config c5 | cut -c1-700
echo "== concurrent callers, coalesced";                                            python tools/bench_coalesce.py | tail -3
