#!/usr/bin/env python3
"""PMC evidence for the wide matrix-core path of the IVF-PQ search (ivf_pq_wide.hip; 1M x 768, the reference's default pq_dim 384,
10 k queries, 32 probes, k 10 - bench.py's PQ-768 line).
  python scripts/pmc_wide.py run                      # builds the index and runs 4 searches (the profiled command)
  python scripts/pmc_wide.py summarize DIR OUT.json   # per-kernel counter sums of the rocprofv3 --pmc passes under DIR
HBM bytes = (FETCH_SIZE x 2 + WRITE_SIZE) KiB as MI355X_MICROARCH.md prescribes for gfx950 (scripts/pmc_c4_c2.py, bench.py)."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import numpy as np, torch
    import bench, cuvs_amd
    from cuvs_amd.neighbors import ivf_pq
    dev = torch.device("cuda", 0)
    res = cuvs_amd.common.Resources()
    rows, nq = 1_000_000, 10000
    x = torch.empty((rows, 768), dtype=torch.float16, device=dev)
    bench.gen_rows(rows, 768, 1234, dev, latent=32, n_modes=4096, out=x, spread=0.7)
    q = torch.empty((nq, 768), dtype=torch.float16, device=dev)
    bench.gen_rows(nq, 768, 4321, dev, latent=32, n_modes=4096, out=q, spread=0.7)
    idx = ivf_pq.build(ivf_pq.IndexParams(n_lists=1024, kmeans_n_iters=10, kmeans_trainset_fraction=0.5), x, resources=res)
    sp = ivf_pq.SearchParams(n_probes=32, lut_dtype=np.float16, internal_distance_dtype=np.float32, max_internal_batch_size=nq)
    for _ in range(4):
        ivf_pq.search(sp, idx, q, 10, resources=res)
    res.sync(); torch.cuda.synchronize()


TAGS = ("pqw_filter_kernel", "pq_rescore_blocks_kernel", "pq_rescore_wave_kernel", "pqw_head_bound_kernel", "pqw_bprep_kernel", "pqw_head_survivors_kernel")


def summarize(d, out):
    agg = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            name = next((t for t in TAGS if t in kn), None)
            if name is None:
                continue
            if name == "pqw_filter_kernel":
                name += "<EMIT> (bound-only head phase)" if ", true>" in kn else " (tail pairs)"
            c = agg.setdefault(name, {}).setdefault(r["Counter_Name"], [0.0, 0])
            c[0] += float(r["Counter_Value"]); c[1] += 1
    res = {}
    for name, cs in agg.items():
        per = {k: v[0] / max(v[1], 1) for k, v in cs.items()}
        cyc = per.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        hbm = (2.0 * per.get("FETCH_SIZE", 0.0) + per.get("WRITE_SIZE", 0.0)) * 1024.0
        line = {"dispatches_seen": max(v[1] for v in cs.values()), "hbm_bytes_per_dispatch": int(hbm), "cycles_per_dispatch": int(cyc),
                "tcc_hit_rate": round(per.get("TCC_HIT_sum", 0.0) / max(per.get("TCC_HIT_sum", 0.0) + per.get("TCC_MISS_sum", 0.0), 1.0), 4)}
        if cyc > 0:
            line["kernel_ms_at_2p4ghz"] = round(cyc / 2.4e9 * 1e3, 3)
            line["hbm_gbs_at_2p4ghz_cycles"] = round(hbm / (cyc / 2.4e9) / 1e9, 1)
            line["valu_busy_of_measured_cycles"] = round(per.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (cyc * 1024), 4)
            line["mfma_busy_of_measured_cycles"] = round(per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 4)
            line["lds_busy_of_measured_cycles"] = round(per.get("SQ_LDS_IDX_ACTIVE", 0.0) / (cyc * 256), 4)
        line["raw_per_dispatch"] = {k: round(v, 1) for k, v in per.items()}
        res[name] = line
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        summarize(sys.argv[2], sys.argv[3])
