"""bench.py as the driver runs it: a fresh child process, the supervisor form, small sizes.
Every field the measurement contract names must be on the ONE stdout line, for argument pairs that
used to break the post-processing (few steps, no warmup)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

# the rehearsal form of the other BASELINE configurations: the same code, small sizes
SMALL_OTHER = ["--sustained-s", "0.3", "--c2-rows", "50000", "--c5-queries", "256", "--c3-rows", "160000", "--c3-batch", "128",
               "--c4-rows", "20000", "--c4-queries", "64"]


def _run(extra, timeout=600):
    env = dict(os.environ)
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True,
                       env=env, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line
    return json.loads(lines[0]), r.stderr


@pytest.mark.parametrize("steps,warmup", [(3, 1), (1, 0)])
def test_bench_child_process_prints_the_whole_contract_line(steps, warmup):
    out, err = _run(["--gpus", "1", "--rows", "200000", "--steps", str(steps), "--warmup", str(warmup),
                     "--cpu-sample-rows", "20000", "--cpu-queries", "4"] + SMALL_OTHER)
    assert "errors" not in out, out["errors"]
    assert out["metric"].startswith("flat-cosine QPS") and out["unit"] == "queries/s"
    assert out["n_gpus"] == 1 and out["steps"] == steps and out["warmup"] == warmup
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert out["dtype"] == "f32" and out["data"] == "synthetic" and out["scaling"] == "weak"
    assert out["vs_baseline"] is None and out["higher_is_better"] is True
    assert "workload" in out["config"] and "model" not in out["config"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["launches_timed"] == steps  # HIP events saw exactly the timed launches
    assert rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["algorithmic_bytes_per_launch"] == 200000 * 384 * 4
    # kernel time fits inside the step time
    assert rf["avg_launch_ms"] <= out["ms_per_step"] * 1.05
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    par = out["parity"]
    assert par["ids_bit_exact"] == "4/4" and par["max_abs_score_diff"] == 0.0 and par["recall_at_10"] == 1.0
    assert out["config"]["fast_vs_exact_full_size"].startswith("4/4")
    assert out["config"]["search_paths_seen"] == [1]  # PATH_FAST: the f32 scan is what was timed
    assert out["config"]["prewarm_queries_untimed"] >= 10
    assert "value_first_5_steps" in out
    # the other BASELINE configurations ride on the same line, each with its own roofline object
    sus = out["value_sustained"]
    assert sus["queries"] >= 64 and sus["value"] > 0 and sus["k_scan_avg_launch_ms"] > 0
    oc = out["config"]["other_configs"]
    assert set(oc) == {"c2", "c5", "c3_shard", "c3_full_one_card", "c4_hnsw"}
    assert oc["c2"]["roofline"]["bound"] == "hbm" and 0 < oc["c2"]["roofline"]["frac"] < 1 and oc["c2"]["fast_vs_exact"].startswith("4/4")
    assert oc["c5"]["roofline"]["bound"] == "mfma" and oc["c5"]["roofline"]["frac"] > 0
    assert oc["c5"]["rows_identical_to_single_search"].startswith("16/16")
    # `value` takes its inputs from HBM (the contract); the host form -- staging + PCIe inside the call -- rides beside it
    assert oc["c5"]["host_queries_pcie_inclusive"]["identical_to_device_queries"] is True
    assert oc["c5"]["host_queries_pcie_inclusive"]["ms_per_batch"] > 0 and oc["c5"]["ms_per_batch"] > 0
    c3 = oc["c3_shard"]
    assert c3["n_gpus"] == 1 and c3["rows_per_rank"] == 20000 and c3["roofline"]["bound"] == "mfma" and c3["roofline"]["frac"] > 0
    assert c3["own_rows_match_single_search"] == "4/4" and c3["transport"].startswith("RCCL")
    assert set(c3["exchange_ms_per_batch"]) >= {"ncclAllGather", "merge_kernel_and_d2h", "local_search_host_clock"}
    # the same batch taken from device memory (vl_shard_search_batch_dev): timed beside the host form, identical answer
    assert c3["device_queries"]["identical_to_host_queries"] is True and c3["device_queries"]["ms_per_batch"] > 0
    assert c3["ms_per_batch"] == c3["device_queries"]["ms_per_batch"] and c3["host_queries_pcie_inclusive"]["ms_per_batch"] > 0
    # config 3 at its own shape: every shard of the corpus on this card, the two halves of vl_shard_search_batch around the merge
    cf = oc["c3_full_one_card"]
    assert cf["shards"] == 8 and cf["rows_per_shard"] == [20000] * 8 and cf["ms_per_batch_per_shard"] > 0 and cf["merge_ms_for_all_records"] > 0
    assert cf["second_pass_identical"] is True and cf["scores_sorted"] is True and cf["every_shards_rows_match_its_own_single_search"] == "4/4"
    assert out["nccl_ranks_seen"] == 1 and c3["nccl_ranks_seen"] == 1     # what RCCL's communicator itself reports (vl_comm_world)
    cc = out["concurrent_16_threads"]
    assert cc["threads"] == 16 and cc["identical_to_lone_search"] == "16/16" and cc["value"] > 0
    assert cc["passes"] <= cc["queries"] and cc["latency_ms"]["p99"] >= cc["latency_ms"]["p50"] > 0
    nt = cc["native_threads"]                  # the same loop from pthreads (no GIL between a return and the next call)
    assert "skipped" in nt or (nt["identical_to_lone_search"] == "16/16" and nt["value"] > 0 and nt["without_adaptive_gather"]["value"] > 0)
    off = cc["without_adaptive_gather"]        # the same loop with the leader's gather off: window 0 as rounds 1-3 ran it
    assert off["identical_to_lone_search"] == "16/16" and off["leader_waits"] == 0 and off["value"] > 0
    c4 = oc["c4_hnsw"]
    assert c4["parity"].startswith("unpinned") and set(c4["data"]) == {"latent16", "clustered", "iid_gaussian"}
    for dist_name in ("latent16", "clustered"):   # the construction beam's sweep: 128 (rounds 1-3), 200, 400 (the default)
        sw = c4["data"][dist_name]["ef_construction_sweep"]
        assert set(sw) == {"128", "200", "400"} and all(0.0 <= v["recall_at_10_strict_beam"] <= v["recall_at_10_ef128"] + 0.05 for v in sw.values())
        c10 = c4["data"][dist_name]["cpu_hnsw_walk_same_graph_strict_beam"]   # SURVEY H5, like for like: the reference's beam on both sides
        assert c10["ef"] == 10 and c10["cpu_ms_per_query"] > 0 and c10["gpu_lone_query_ms_same_beam"] > 0
    for dist_name in ("latent16", "clustered", "iid_gaussian"):
        d = c4["data"][dist_name]
        for ef in ("ef10", "ef32", "ef128"):
            assert 0.0 <= d[ef]["recall_at_10_vs_exact_f64_order"] <= 1.0 and d[ef]["roofline"]["frac"] >= 0
        assert d["ef128"]["recall_at_10_vs_exact_f64_order"] >= d["ef10"]["recall_at_10_vs_exact_f64_order"] - 0.05
        cw = d["cpu_hnsw_walk_same_graph"]   # config 4's "recall@10 vs CPU HNSW": the checker's walk of the exported graph
        assert "skipped" not in cw, cw
        assert cw["cores"] == 1 and cw["cpu_ms_per_query"] > 0 and 0.0 <= cw["cpu_recall_at_10_vs_u64_distance_order"] <= 1.0
        assert abs(cw["cpu_recall_at_10_vs_u64_distance_order"] - cw["gpu_recall_at_10_same_queries"]) <= 0.25
    full = cb["full_size_check"]
    assert full["ran"] is True and full["gpu_answer_bit_identical"].startswith("2/2")
    assert rf["kernel_variant"]["query_in_kernarg"] == 1


def test_bench_two_ranks_without_an_external_launcher():
    """`--gpus 2` alone starts two ranks (rehearsal: both on this card, gloo rendezvous)."""
    env_before = os.environ.get("VL_BENCH_REHEARSE")
    os.environ["VL_BENCH_REHEARSE"] = "1"
    try:
        out, err = _run(["--gpus", "2", "--rows", "100000", "--steps", "4", "--warmup", "1", "--c3-rows", "60000", "--c3-batch", "128"])
    finally:
        if env_before is None:
            os.environ.pop("VL_BENCH_REHEARSE", None)
        else:
            os.environ["VL_BENCH_REHEARSE"] = env_before
    assert out["n_gpus"] == 2 and out["steps"] == 4
    assert "replicas" in out["config"]["parallelism"]
    assert out["value"] > 0 and out["roofline"]["launches_timed"] == 4
    assert "cpu_baseline" not in out  # rank 0 at N = 1 only
    # after the timed region the ranks also answer one row-sharded batch together (device merge; RCCL on real multi-GPU)
    rs = out["config"]["other_configs"]["c3_row_sharded"]
    assert rs["n_gpus"] == 2 and rs["rows_per_rank"] == 30000 and rs["dim"] == 768 and rs["queries"] == 128
    assert rs["identical_on_every_rank"] is True and rs["own_rows_match_single_search"] == "4/4"
    assert rs["roofline"]["bound"] == "mfma" and rs["roofline"]["whole_call"]["frac"] > 0
    assert "errors" not in out, out.get("errors")
