#!/usr/bin/env python3
"""PMC evidence for the C4 (CAGRA 10M x 768 fp16) and C2 (IVF-Flat 10M x 128 fp32) search kernels.

  python scripts/pmc_c4_c2.py build  [--c4-rows N]   # builds both indexes once, saves them + the queries under /tmp/pmc_idx
  python scripts/pmc_c4_c2.py run                    # loads them and runs 3 searches each (the profiled command)
  python scripts/pmc_c4_c2.py summarize DIR OUT.json # per-kernel counter sums of rocprofv3 --pmc passes under DIR -> JSON

rocprofv3 --pmc passes (counters only, one group per run: MI355X_MICROARCH.md) wrap the `run` step:
scripts/gpu_r04_pmc.sh. HBM bytes = FETCH_SIZE x 2 (gfx950: the counter reports 64-byte units, the unit is 32... - the guide's
correction) x 1024 / ... are computed in summarize exactly as bench.py does for the PQ scan."""
import csv
import glob
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

D = "/tmp/pmc_idx"


ONLY_C2 = os.environ.get("PMC_ONLY_C2") == "1"  # round 6: C2 alone (scripts/gpu_r06_pmc_c2.sh)


def build(c4_rows):
    import cuvs_amd
    from cuvs_amd.neighbors import cagra, ivf_flat

    os.makedirs(D, exist_ok=True)
    dev = torch.device("cuda", 0)
    res = cuvs_amd.common.Resources()
    if ONLY_C2:
        return build_c2(res, dev)
    x = torch.empty((c4_rows, 768), dtype=torch.float16, device=dev)
    bench.gen_rows(c4_rows, 768, 1234, dev, latent=24, n_modes=1, out=x)
    q = torch.empty((10000, 768), dtype=torch.float16, device=dev)
    bench.gen_rows(10000, 768, 4321, dev, latent=24, n_modes=1, out=q)
    t0 = time.time()
    idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=128, graph_degree=64), x, resources=res)
    res.sync()
    print(f"C4 built in {time.time() - t0:.1f}s", flush=True)
    cagra.save(f"{D}/c4.idx", idx, include_dataset=True, resources=res)
    torch.save(q.cpu(), f"{D}/c4_q.pt")
    del idx, x, q
    torch.cuda.empty_cache()
    build_c2(res, dev)


def build_c2(res, dev):
    from cuvs_amd.neighbors import ivf_flat

    x = bench.gen_rows(10_000_000, 128, 1234, dev)
    q = bench.gen_rows(10000, 128, 4321, dev)
    idx = ivf_flat.build(ivf_flat.IndexParams(n_lists=4096, kmeans_trainset_fraction=0.1), x, resources=res)
    res.sync()
    ivf_flat.save(f"{D}/c2.idx", idx, resources=res)
    torch.save(q.cpu(), f"{D}/c2_q.pt")
    print("C2 built", flush=True)


def run():
    import cuvs_amd
    from cuvs_amd.neighbors import cagra, ivf_flat

    dev = torch.device("cuda", 0)
    res = cuvs_amd.common.Resources()
    dd = torch.empty((10000, 10), dtype=torch.float32, device=dev)
    if not ONLY_C2:
        idx = cagra.load(f"{D}/c4.idx", resources=res)
        q = torch.load(f"{D}/c4_q.pt").to(dev)
        nb = torch.empty((10000, 10), dtype=torch.int32, device=dev)
        for algo in ("single_cta", "multi_cta"):
            sp = cagra.SearchParams(itopk_size=64, algo=algo)
            for _ in range(3):
                cagra.search(sp, idx, q, 10, neighbors=nb, distances=dd, resources=res)
        res.sync()
        del idx
        torch.cuda.empty_cache()
    idx = ivf_flat.load(f"{D}/c2.idx", resources=res)
    q = torch.load(f"{D}/c2_q.pt").to(dev)
    nb = torch.empty((10000, 10), dtype=torch.int64, device=dev)
    sp = ivf_flat.SearchParams(n_probes=64)
    for _ in range(4):  # the first search builds the fp16 residual copy
        ivf_flat.search(sp, idx, q, 10, neighbors=nb, distances=dd, resources=res)
    res.sync()
    torch.cuda.synchronize()


def summarize(d, out):
    agg = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            name = None
            for tag in ("cagra_search_multi_kernel", "cagra_search_kernel", "ivf_flat_scan_kernel", "pq_filter_kernel", "flat_filter2_kernel", "flat_rescore_kernel"):
                if tag in kn:
                    name = tag
            if name == "flat_filter2_kernel" and ", true>" in kn:
                name = "flat_filter2_kernel<EMIT> (bound-only head phase)"
            if name is None:
                continue
            a = agg.setdefault(name, {})
            c = a.setdefault(r["Counter_Name"], [0.0, 0])
            c[0] += float(r["Counter_Value"]); c[1] += 1
    res = {}
    for name, cs in agg.items():
        per = {k: v[0] / max(v[1], 1) for k, v in cs.items()}  # per dispatch
        cyc = per.get("GRBM_GUI_ACTIVE", 0.0) / 8.0            # summed over the 8 XCDs
        hbm = (2.0 * per.get("FETCH_SIZE", 0.0) + per.get("WRITE_SIZE", 0.0)) * 1024.0  # gfx950: FETCH_SIZE x 2, KiB units
        line = {"dispatches_seen": max(v[1] for v in cs.values()), "hbm_bytes_per_dispatch": int(hbm),
                "cycles_per_dispatch": int(cyc),
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
    if sys.argv[1] == "build":
        build(int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000)
    elif sys.argv[1] == "run":
        run()
    else:
        summarize(sys.argv[2], sys.argv[3])
