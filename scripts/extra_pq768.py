import sys, os, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, cuvs_amd
res = cuvs_amd.common.Resources(); dev = torch.device("cuda:0")
print(json.dumps(bench.extra_pq768(res, dev, int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000)), flush=True)
