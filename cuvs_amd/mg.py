"""Single-node multi-GPU search: one process per GPU over torch.distributed (RCCL on ROCm, gloo on CPU tests).

Reference: cpp/src/neighbors/mg/snmg.cuh — SHARDED mode (:128-166 build, :248-375 search: every rank searches all
queries on its row-range shard, ids are translated by the shard offset, partial top-k lists are merged) and
REPLICATED mode (:576-648, queries split across ranks, no communication).

MI355X design: the per-rank result is tiny (Q*k*(4+8) bytes; 1.2 MB at Q=10k, k=10), so instead of the
reference's ncclSend/ncclRecv fan-in to a root (or its log2(R) merge tree) every rank takes part in ONE
all_gather over xGMI and merges the R*k candidates per query locally — results end up replicated, latency is one
collective. No other collective touches the data path.
"""
import torch
import torch.distributed as dist


def shard_rows(n_rows, rank, world):
    """Row range of `rank` (snmg.cuh:128-166: ceil(N/R) rows per rank, the last rank takes the remainder)."""
    per = (n_rows + world - 1) // world
    start = min(n_rows, rank * per)
    return start, min(n_rows, start + per)


def translate_ids(ids, row_offset, invalid_mask=None):
    """Local shard row ids -> global ids (snmg.cuh:420-429); invalid entries stay untouched."""
    if invalid_mask is None:
        invalid_mask = (ids < 0) | (ids == torch.iinfo(torch.int64).max)
    return torch.where(invalid_mask, ids, ids + row_offset)


def merge_parts(dists, ids, k, select_min=True):
    """Merge [Q, P] candidate lists into the best k per query, ordered by (distance, id)
    (knn_merge_parts.cuh:27-103 semantics; ties -> smaller id)."""
    big = torch.iinfo(torch.int64).max
    key_ids = torch.where(ids < 0, torch.full_like(ids, big), ids)
    o1 = torch.argsort(key_ids, dim=1, stable=True)
    d1 = torch.gather(dists, 1, o1)
    i1 = torch.gather(ids, 1, o1)
    o2 = torch.argsort(d1, dim=1, stable=True, descending=not select_min)
    return torch.gather(d1, 1, o2)[:, :k].contiguous(), torch.gather(i1, 1, o2)[:, :k].contiguous()


def all_gather_merge(local_dists, local_ids, k, select_min=True, group=None):
    """One all_gather of the per-shard top-k + local merge. local_*: [Q, k]; returns replicated [Q, k]."""
    world = dist.get_world_size(group)
    if world == 1:
        return merge_parts(local_dists, local_ids, k, select_min)
    packed_d = [torch.empty_like(local_dists) for _ in range(world)]
    packed_i = [torch.empty_like(local_ids) for _ in range(world)]
    dist.all_gather(packed_d, local_dists.contiguous(), group=group)
    dist.all_gather(packed_i, local_ids.contiguous(), group=group)
    return merge_parts(torch.cat(packed_d, dim=1), torch.cat(packed_i, dim=1), k, select_min)


class ShardedIndex:
    """SHARDED-mode index: `module` is cuvs_amd.neighbors.{brute_force, ivf_flat, ivf_pq, cagra}."""

    def __init__(self, module, index, row_offset, select_min=True):
        self.module, self.index, self.row_offset, self.select_min = module, index, row_offset, select_min

    @classmethod
    def build(cls, module, build_fn, dataset_rows, n_total, group=None):
        """build_fn(local_rows) -> index. dataset_rows: this rank's rows (already sliced with shard_rows)."""
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        start, _ = shard_rows(n_total, rank, world)
        return cls(module, build_fn(dataset_rows), start)

    def search(self, search_fn, queries, k, group=None):
        """search_fn(index, queries, k) -> (distances, neighbors) on the local shard."""
        d, i = search_fn(self.index, queries, k)
        i = translate_ids(i.to(torch.int64), self.row_offset)
        if not dist.is_initialized():
            return d, i
        return all_gather_merge(d, i, k, self.select_min, group)


def replicated_query_slice(n_queries, rank, world):
    """REPLICATED mode: contiguous slice of the query batch owned by `rank` (snmg.cuh:576-648, ROUND_ROBIN over
    batches degenerates to this for one batch per rank)."""
    return shard_rows(n_queries, rank, world)
