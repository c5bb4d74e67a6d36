"""World-N check of the shard communicator's collectives over RCCL (one process per GPU), each against a numpy expectation:
cuvsAmdShardAllGatherTopK (ncclAllGather of the packed [Q, k] blocks + the R-way merge) for select-min and select-max, and - through a
tiny list-sharded IVF-PQ search - the in-place probe all-gather (ncclAllGather of uint32 words) and the bound all-reduce
(ncclAllReduce min). Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/rccl_collectives_check.py
Every rank prints one line; a rank that waits longer than the launcher's timeout is killed by scripts/gpu_first_8gpu.sh."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuvs_amd  # noqa: E402
from cuvs_amd.neighbors import ivf_pq, ivf_pq_sharded as sh  # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
res = cuvs_amd.common.Resources()
comm = sh.ShardComm.from_torch(res)

# ---- all_gather_topk: every rank's block is a function of (seed, rank): all ranks can compute the expectation
nq, k = 513, 10
parts_d, parts_i = [], []
for r in range(world):
    g = np.random.default_rng(100 + r)
    d = np.sort(g.random((nq, k), dtype=np.float32), axis=1)
    i = g.integers(0, 1 << 40, size=(nq, k)).astype(np.int64)
    parts_d.append(d); parts_i.append(i)
for select_min in (True, False):
    mine_d = parts_d[rank] if select_min else np.ascontiguousarray(parts_d[rank][:, ::-1])
    mine_i = parts_i[rank] if select_min else np.ascontiguousarray(parts_i[rank][:, ::-1])
    od, oi = comm.all_gather_topk(torch.from_numpy(mine_d).to(dev), torch.from_numpy(mine_i).to(dev), select_min=select_min, resources=res)
    res.sync()
    pd = parts_d if select_min else [np.ascontiguousarray(p[:, ::-1]) for p in parts_d]
    pi = parts_i if select_min else [np.ascontiguousarray(p[:, ::-1]) for p in parts_i]
    ed, ei = sh.merge_gathered(pd, pi, k, select_min=select_min)
    assert (od.cpu().numpy() == ed).all() and (oi.cpu().numpy() == ei).all(), f"rank {rank}: all_gather_topk(select_min={select_min}) differs"

# ---- a list-sharded search with the communicator attached: probe all-gather + bound all-reduce inside cuvsIvfPqSearch
g = np.random.default_rng(7)
x = (g.random((40000, 64), dtype=np.float32) * 1.9 + 0.1)
q = (g.random((600, 64), dtype=np.float32) * 1.9 + 0.1)
xt, qt = torch.from_numpy(x).to(dev), torch.from_numpy(q).to(dev)
ip = ivf_pq.IndexParams(n_lists=32, pq_dim=32, kmeans_n_iters=10, add_data_on_build=False)
shard = sh.build(ip, xt, rank, world, resources=res)
sh.extend(shard, xt, torch.arange(len(x), dtype=torch.int64, device=dev), resources=res)
sh.attach_comm(shard, comm)
sp = ivf_pq.SearchParams(n_probes=12, max_internal_batch_size=32768)
d, i = sh.search(sp, shard, qt, 10, comm, resources=res)
res.sync()
whole = ivf_pq.build(ivf_pq.IndexParams(n_lists=32, pq_dim=32, kmeans_n_iters=10), xt, resources=res)
wd, wi = ivf_pq.search(sp, whole, qt, 10, resources=res)
res.sync()
same_d = bool((np.sort(d.cpu().numpy(), 1) == np.sort(wd.cpu().numpy(), 1)).all())
assert same_d, f"rank {rank}: the merged answer of the {world} shards differs from the unsharded index"
sh.attach_comm(shard, None)
dist.barrier()
print(f"rank {rank} of {world}: RCCL collectives OK (all_gather_topk min / max, sharded search == unsharded)", flush=True)
comm.close()
dist.destroy_process_group()
