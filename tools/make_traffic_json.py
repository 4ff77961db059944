#!/usr/bin/env python3
"""profiles/traffic.json entries for the batch filter (k_mfma_rows, pass 1) from tools/r4_traffic.sh's outputs:
traffic per batch = average FETCH_SIZE per dispatch x dispatches / batches, x 1024 (KB) x 2 (gfx950: FETCH_SIZE tallies 64 B
per 128-B request, MI355X_MICROARCH.md HBM / rocprofv3 section).   usage: python tools/make_traffic_json.py gpurun_out/r4t c3 c5"""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, cfgs = sys.argv[1], sys.argv[2:]
path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(path))
if cfgs and cfgs[0] == "hnsw":  # config 4: one entry per beam width (tools/r4_traffic_hnsw.sh)
    out = []
    for ef in cfgs[1:] or ["10", "32", "128"]:
        tgt = json.loads(open(os.path.join(src, f"c4_ef{ef}_target.json")).read().strip().splitlines()[-1])
        kname = f"vl::k_hnsw_search<0, {tgt['list_slots']}>"
        row = None
        for r in csv.DictReader(open(os.path.join(src, f"c4_ef{ef}_pmc_fetch_size.csv"))):
            if r["kernel"].startswith(kname) and r["counter"] == "FETCH_SIZE":
                row = r
        assert row and int(row["dispatches"]) == tgt["batches"], (kname, row, tgt)
        per_batch = float(row["avg_value"]) * 1024 * 2
        evq = tgt["distance_evals_per_query"]
        beam = max(int(ef), 10)
        alg = tgt["queries"] * ((evq - beam) * tgt["dim"] * 4 + beam * tgt["dim"] * 8)  # navigation rows f32, the final beam's f64
        out.append({"config": "c4", "workload": {k: tgt[k] for k in ("data", "rows", "dim", "queries", "ef", "ef_construction")},
                    "kernel": row["kernel"], "counter": "FETCH_SIZE", "dispatches": int(row["dispatches"]),
                    "distance_evals_per_query": evq, "traffic_bytes_per_batch": int(per_batch), "algorithmic_bytes_per_batch": int(alg),
                    "traffic_over_algorithmic": round(per_batch / alg, 4),
                    "correction": "x2 (gfx950 FETCH_SIZE counts 64 B per 128-B request), KB -> bytes x1024",
                    "source": f"profiles/r04_c4_ef{ef}_rocprofv3_pmc_fetch_size.csv: rocprofv3 --pmc FETCH_SIZE of tools/pmc_hnsw_target.py --ef {ef} (tools/r4_traffic_hnsw.sh)"})
    tj["k_hnsw_search"] = out
    json.dump(tj, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))
    sys.exit(0)
entries = [e for e in tj.get("k_mfma_rows", []) if e["config"] not in cfgs]
for c in cfgs:
    tgt = json.loads(open(os.path.join(src, f"{c}_target.json")).read().strip().splitlines()[-1])
    kname = f"vl::k_mfma_rows<{tgt['plan']['ksteps']}, 1, {tgt['metric']}"
    row = None
    for r in csv.DictReader(open(os.path.join(src, f"{c}_pmc_fetch_size.csv"))):
        if r["kernel"].startswith(kname) and r["counter"] == "FETCH_SIZE":
            row = r
    assert row, (kname, "not in the PMC summary")
    disp, avg_kb = int(row["dispatches"]), float(row["avg_value"])
    expect = tgt["batches"] * tgt["sequences_per_batch"] * tgt["plan"]["stages"]
    assert disp == expect, (disp, expect)
    per_batch = avg_kb * 1024 * 2 * disp / tgt["batches"]
    ldb = tgt["plan"]["ksteps"] * 16
    alg = tgt["rows"] * ldb * 2 * tgt["sequences_per_batch"]  # every sequence streams the bf16 slab once
    entries.append({"config": c, "workload": {"rows": tgt["rows"], "dim": tgt["dim"], "queries": tgt["queries"], "metric": tgt["metric"]},
                    "kernel": row["kernel"], "plan": tgt["plan"], "counter": "FETCH_SIZE", "dispatches": disp, "batches": tgt["batches"],
                    "avg_kb_per_dispatch": avg_kb, "traffic_bytes_per_batch": int(per_batch),
                    "algorithmic_bytes_per_batch": int(alg), "traffic_over_algorithmic": round(per_batch / alg, 4),
                    "correction": "x2 (gfx950 FETCH_SIZE counts 64 B per 128-B request), KB -> bytes x1024",
                    "source": f"profiles/r04_{c}_rocprofv3_pmc_fetch_size.csv: rocprofv3 --pmc FETCH_SIZE of tools/pmc_filter_target.py --config {c} (tools/r4_traffic.sh)"})
tj["k_mfma_rows"] = entries
json.dump(tj, open(path, "w"), indent=1)
print(json.dumps(entries, indent=1))
