import sys, os, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, cuvs_amd
res = cuvs_amd.common.Resources(); dev = torch.device("cuda:0")
print(json.dumps(bench.extra_c4(res, dev, 10_000_000, 24, modes=4096, spread=0.7)), flush=True)
