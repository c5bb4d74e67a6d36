import sys, os, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, cuvs_amd
res = cuvs_amd.common.Resources(); dev = torch.device("cuda:0")
print(json.dumps(bench.extra_flat768(res, dev)), flush=True)
print(json.dumps(bench.extra_cagra128(res, dev)), flush=True)
