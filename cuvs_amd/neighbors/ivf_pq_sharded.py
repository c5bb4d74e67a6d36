"""List-sharded multi-GPU IVF-PQ: one process per GPU, the per-rank top-k blocks all-gathered by RCCL inside the
library (include/cuvs_amd/shard.h, cuvs_amd/csrc/shard_comm.hip). Reference: the sharded search of
cpp/src/neighbors/mg/snmg.cuh:248-375 (row-range shards, NCCL send/recv fan-in) - here the split by IVF list with one
global coarse quantizer: list L lives on rank L % world - or, dealt by size, on owners[L] (deal_lists: greedy LPT over the
rows per list) - every rank ranks all centres and scans the probes it owns.

Only the rendezvous (128 bytes of id from rank 0 to the others) needs an out-of-band channel - `ShardComm.from_torch`
uses the torch.distributed store the launcher already provides; the data path never touches torch.distributed."""
import ctypes as C

import numpy as np
import torch

from .._lib import check, lib
from ..common import auto_sync_resources
from . import ivf_pq

ID_BYTES = 128


def owner(list_id, world, owners=None):
    return list_id % world if owners is None else int(owners[list_id])


def deal_lists(weights, world):
    """Greedy longest-processing-time dealing of the lists (rows per list) to `world` ranks - cuvsAmdShardDealLists, the same
    table on every rank for the same weights. Returns int32 [n_lists]."""
    w = np.ascontiguousarray(weights, dtype=np.uint64)
    owners = np.empty(len(w), np.int32)
    check(lib().cuvsAmdShardDealLists(w.ctypes.data_as(C.c_void_p), C.c_uint32(len(w)), C.c_int(world),
                                      owners.ctypes.data_as(C.c_void_p)))
    return owners


@auto_sync_resources
def list_histogram(index, rows, counts=None, resources=None):
    """Adds the number of `rows` (device tensor [n, dim]) falling into every list to `counts` (uint64 [n_lists])."""
    from .._lib import Tensor

    if counts is None:
        counts = np.zeros(index.n_lists, np.uint64)
    t = Tensor(rows.contiguous())
    check(lib().cuvsAmdIvfPqListHistogram(resources.get_c_obj(), index._p, t.ptr, counts.ctypes.data_as(C.c_void_p)))
    return counts


@auto_sync_resources
def row_labels(index, rows, resources=None):
    """The list of every row (device tensor [n, dim]) as a device int64 tensor [n] (cuvsAmdIvfPqRowLabels)."""
    from .._lib import Tensor

    t = Tensor(rows.contiguous())
    out = torch.empty(rows.shape[0], dtype=torch.int32, device=rows.device)
    check(lib().cuvsAmdIvfPqRowLabels(resources.get_c_obj(), index._p, t.ptr, C.c_void_p(out.data_ptr())))
    return out.to(torch.int64)


def set_list_owners(index, owners, rank, world):
    """Marks the (still empty) index as the shard of `rank` with an explicit owner per list (same table on every rank)."""
    o = np.ascontiguousarray(owners, dtype=np.int32)
    check(lib().cuvsAmdIvfPqSetListOwners(index._p, o.ctypes.data_as(C.c_void_p), C.c_uint32(len(o)), C.c_int(rank), C.c_int(world)))


class ShardComm:
    def __init__(self, rank, world, unique_id, resources):
        assert len(unique_id) == ID_BYTES
        self.rank, self.world = rank, world
        self._c = C.c_void_p()
        buf = (C.c_char * ID_BYTES).from_buffer_copy(unique_id)
        check(lib().cuvsAmdShardCommCreate(resources.get_c_obj(), buf, C.c_int(rank), C.c_int(world), C.byref(self._c)))

    @staticmethod
    def unique_id(host_staged=False):
        """128-byte rendezvous id: RCCL's (one process per GPU, xGMI) or - host_staged=True - the name of the mapped file
        of the host-staged transport (ranks of one host that share a device; cuvsAmdShardCommGetUniqueIdHostStaged)."""
        buf = (C.c_char * ID_BYTES)()
        check((lib().cuvsAmdShardCommGetUniqueIdHostStaged if host_staged else lib().cuvsAmdShardCommGetUniqueId)(buf))
        return bytes(buf.raw)

    @classmethod
    def from_torch(cls, resources, group=None, host_staged=False):
        """rank / world / id exchange through an initialised torch.distributed group (control plane only)."""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id(host_staged) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(rank, world, box[0], resources)

    def close(self):
        if self._c:
            check(lib().cuvsAmdShardCommDestroy(self._c))
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @auto_sync_resources
    def all_gather_topk(self, distances, neighbors, select_min=True, out=None, resources=None):
        """[Q, k] float32 / int64 device tensors of this rank -> merged [Q, k] (on every rank)."""
        nq, k = distances.shape
        od, oi = out if out is not None else (torch.empty_like(distances), torch.empty_like(neighbors))
        fn = lib().cuvsAmdShardAllGatherTopK
        fn.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        check(fn(resources.get_c_obj(), self._c, distances.data_ptr(), neighbors.data_ptr(), nq, k, int(select_min),
                 od.data_ptr(), oi.data_ptr()))
        return od, oi


@auto_sync_resources
def build(index_params, trainset, rank, world, owners=None, resources=None):
    """Train the model (identical on every rank: same trainset, deterministic k-means) and mark the empty index as the
    shard of `rank`: list L on rank L % world, or on owners[L] (deal_lists). index_params must have
    add_data_on_build=False."""
    assert not index_params.add_data_on_build, "list-sharded build trains first and adds rows with extend()"
    index = ivf_pq.build(index_params, trainset, resources=resources)
    if owners is None:
        check(lib().cuvsAmdIvfPqSetListShard(index._p, C.c_int(rank), C.c_int(world)))
    else:
        set_list_owners(index, owners, rank, world)
    return index


def attach_comm(index, comm):
    """Lets the shard's searches all-reduce the per-query bounds between their two scan phases (collective: every rank
    must then run the same searches). Results do not change, per-rank pruning does."""
    check(lib().cuvsAmdIvfPqSetShardComm(index._p, comm._c if comm is not None else None))
    index._shard_comm = comm  # keep it alive


def extend(index, rows, ids, resources=None):
    """Adds the rows that fall into this rank's lists (the others are dropped); ids = global row ids."""
    return ivf_pq.extend(index, rows, ids, resources=resources)


def search(search_params, index, queries, k, comm, select_min=True, resources=None):
    """Global n_probes nearest lists per query, scan of the owned ones, ONE all-gather + merge."""
    d, i = ivf_pq.search(search_params, index, queries, k, resources=resources)
    return comm.all_gather_topk(d, i, select_min=select_min, resources=resources)


def merge_gathered(d_parts, i_parts, k, select_min=True):
    """CPU twin of the device merge (shard_comm.hip): parts [world, Q, k] -> the k best of every query; which of several
    equal k-th distances survive is decided by (rank, position), the winners come out ordered by (distance, id)."""
    d = np.concatenate(list(d_parts), axis=1)
    i = np.concatenate(list(i_parts), axis=1)
    key = d if select_min else -d
    out_d = np.empty((d.shape[0], k), d.dtype)
    out_i = np.empty((d.shape[0], k), i.dtype)
    for q in range(d.shape[0]):
        sel = np.argsort(key[q], kind="stable")[:k]            # (distance, column) - column = rank-major position
        o = np.lexsort((i[q, sel], key[q, sel]))                # winners by (distance, id)
        out_d[q], out_i[q] = d[q, sel][o], i[q, sel][o]
    return out_d, out_i
