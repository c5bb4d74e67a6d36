"""Row-range sharded search over the native communicator (include/cuvs_amd/shard.h): the reference's SHARDED mode
(cpp/src/neighbors/mg/snmg.cuh:128-166 build, :248-375 search) for any index type, one process per GPU.

Every rank builds a complete index over ITS rows, searches ALL queries, translates the local row ids to global ids on
the device (cuvsAmdShardTranslateIds, snmg.cuh:420-429) and takes part in ONE ncclAllGather of the [Q, k] blocks
(cuvsAmdShardAllGatherTopK) followed by the R-way merge - instead of the reference's ncclSend/ncclRecv fan-in to a
root (:298-340). `cuvs_amd.mg` is the same algorithm over torch.distributed (the gloo-testable twin)."""
import ctypes as C

import torch

from .._lib import check, lib
from ..common import auto_sync_resources
from ..mg import shard_rows  # noqa: F401  (row range of a rank: ceil(N / R) rows each)


@auto_sync_resources
def translate_ids(local_ids, row_offset, out=None, resources=None):
    """[Q, k] uint32 (torch.int32 storage, the CAGRA output) or int64 local ids -> int64 global ids, on the device."""
    ids = local_ids.contiguous()
    bits = 32 if ids.dtype in (torch.int32, torch.uint32) else 64
    assert bits == 32 or ids.dtype == torch.int64
    if out is None:
        out = torch.empty(ids.shape, dtype=torch.int64, device=ids.device)
    check(lib().cuvsAmdShardTranslateIds(resources.get_c_obj(), C.c_void_p(ids.data_ptr()), C.c_int(bits),
                                         C.c_int64(ids.numel()), C.c_int64(row_offset), C.c_void_p(out.data_ptr())))
    return out


class RowShard:
    """One rank's shard: `module` is cuvs_amd.neighbors.{brute_force, ivf_flat, ivf_pq, cagra}, `index` its index over the
    rows [row_offset, row_offset + n_local) of the corpus."""

    def __init__(self, module, index, row_offset, comm, select_min=True):
        self.module, self.index, self.row_offset, self.comm, self.select_min = module, index, row_offset, comm, select_min

    def search(self, search_fn, queries, k, resources=None):
        """search_fn(index, queries, k) -> (distances [Q, k] float32, neighbors [Q, k]) on the local shard; returns the
        merged (distances, int64 global neighbors), replicated on every rank."""
        d, i = search_fn(self.index, queries, k)
        gi = translate_ids(i, self.row_offset, resources=resources)
        return self.comm.all_gather_topk(d.contiguous(), gi, select_min=self.select_min, resources=resources)
