"""Corpus sharded across the GPUs of one node (one process per GPU, torch.distributed).

Each rank holds a contiguous index range of the reference-fingerprint database and scans only
that.  The single exchange step is a MAX all-reduce of one int64 per query:
``key = float_bits(best score) << 32 | (0xFFFFFFFF - global index)``.  Scores are >= 0, so their
IEEE bit patterns order like the values, and the complemented index makes the LOWEST index win
ties -- the strict '<' of the upstream best-match loop (LBAudioDetectiveTests.m:80-83).  With the
"nccl" backend (RCCL on ROCm) the 8-byte message travels over xGMI; "gloo" is used by the CPU
tests of this reduction logic.
"""
from __future__ import annotations

# True: the helpers below run their collectives whenever torch.distributed is initialised, also in a group of ONE rank
# (bench.py --force-dist: the N > 1 code paths executed on a single GPU); False: a lone rank skips them.
FORCE_COLLECTIVES = False


def _collective(group=None) -> bool:
    """Is there a process group this call should talk to?"""
    try:
        import torch.distributed as dist
    except ImportError:                 # a torch built without distributed support: one rank, no group
        return False
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return FORCE_COLLECTIVES or dist.get_world_size(group) > 1


def shard_range(n_entries: int, rank: int, world_size: int):
    """Contiguous [begin, end) of rank's shard; sizes differ by at most one."""
    base, extra = divmod(n_entries, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def make_key(score_bits: int, global_index: int) -> int:
    return (score_bits << 32) | (0xFFFFFFFF - global_index)


def decode_key(key: int):
    """(index, score) from a reduced key; index -1 when nothing scored above 0."""
    import struct
    key &= 0xFFFFFFFFFFFFFFFF
    score = struct.unpack("<f", struct.pack("<I", key >> 32))[0]
    if key == 0 or not score > 0.0:
        return -1, score
    return 0xFFFFFFFF - (key & 0xFFFFFFFF), score


def allreduce_best(key_tensor, group=None):
    """In-place MAX all-reduce of an int64 key tensor (any device the backend supports)."""
    import torch.distributed as dist
    if _collective(group):
        dist.all_reduce(key_tensor, op=dist.ReduceOp.MAX, group=group)
    return key_tensor


def gather_packed(local_packed, group=None):
    """Optional last step of sharded fingerprinting: every rank contributes its [n_local, count, 32]
    packed sub-fingerprints (160 B per one-second clip) and receives all of them in rank order.  Ranks
    may hold different numbers of clips.  The fingerprint path itself needs no collective."""
    import torch
    import torch.distributed as dist
    if not _collective(group):
        return local_packed
    world = dist.get_world_size(group)
    n = torch.tensor([local_packed.shape[0]], dtype=torch.int64, device=local_packed.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    biggest = max(counts)
    padded = torch.zeros((biggest,) + tuple(local_packed.shape[1:]), dtype=local_packed.dtype, device=local_packed.device)
    padded[: local_packed.shape[0]] = local_packed
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def broadcast_fingerprint(fp, src: int = 0, group=None, device=None):
    """Send rank `src`'s query fingerprint to every rank (the 160-byte "query broadcast" of the sharded
    compare).  `fp` is a Fingerprint on the source rank and may be None elsewhere; returns a Fingerprint
    on every rank.  Two small collectives: the shape, then the Booleans."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from .api import Fingerprint
    if not _collective(group):
        return fp
    if device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    shape = torch.zeros(2, dtype=torch.int64, device=device)
    if dist.get_rank(group) == src:
        bools = fp.to_bools()
        shape[0], shape[1] = bools.shape[0], bools.shape[1]
    dist.broadcast(shape, src=src, group=group)
    n, length = int(shape[0]), int(shape[1])
    data = torch.zeros(max(1, n * length), dtype=torch.uint8, device=device)
    if dist.get_rank(group) == src and n * length:
        data[: n * length] = torch.from_numpy(np.ascontiguousarray(bools).reshape(-1)).to(device)
    dist.broadcast(data, src=src, group=group)
    if dist.get_rank(group) == src:
        return fp
    out = Fingerprint(length)
    rows = data[: n * length].cpu().numpy().reshape(n, length)
    for row in rows:
        out.add_subfingerprint(row)
    return out


def make_comm(rank: int = 0, world_size: int = 1, group=None):
    """The library's own RCCL communicator for the sharded query (LBAudioDetectiveCommInitRank).  Rank 0 obtains
    the 128-byte id; torch.distributed -- plumbing, whatever backend it runs on -- carries it to the other ranks.
    The current CUDA/HIP device must already be this rank's."""
    from .api import Comm
    uid, err = [None], None
    if rank == 0:
        try:
            uid[0] = Comm.unique_id()
        except Exception as e:          # noqa: BLE001 -- the other ranks are waiting in the broadcast: tell them
            err = e
    if world_size > 1 or _collective(group):
        # (one rank without FORCE_COLLECTIVES never touches torch.distributed; FORCE_COLLECTIVES must be set identically on
        # every rank BEFORE any helper of this module is called, or the ranks' broadcast calls do not pair up)
        import torch.distributed as dist
        dist.broadcast_object_list(uid, src=0, group=group)
    if uid[0] is None:
        raise err if err is not None else RuntimeError("rank 0 could not obtain an RCCL unique id")
    return Comm(world_size, uid[0], rank)


class ShardedCorpus:
    """This rank's shard of a global corpus plus the collective top-1 query.  With `comm` (make_comm) the
    exchange step runs inside the library (ncclAllReduce of the uint64 keys, LBAudioDetectiveCorpusQuerySharded);
    without it the keys are reduced through torch.distributed (the CPU tests use gloo)."""

    def __init__(self, subfingerprint_length: int, subfingerprints_per_entry: int, n_entries_global: int,
                 rank: int = 0, world_size: int = 1, group=None, comm=None):
        from .api import Corpus
        self.rank, self.world_size, self.group, self.comm = rank, world_size, group, comm
        self.n_entries_global = n_entries_global
        self.begin, self.end = shard_range(n_entries_global, rank, world_size)
        self.local = Corpus(subfingerprint_length, subfingerprints_per_entry, max(1, self.end - self.begin))

    def append_packed_device(self, packed, stream=None):
        self.local.append_packed_device(packed, stream)

    def query_batch(self, fps, range_: int = 0, keys_out=None):
        """Collective: several queries, one pass over every shard, one all-reduce of len(fps) int64 keys."""
        if self.comm is not None:
            return self.local.query_batch_sharded(fps, self.comm, self.begin, range_)
        import torch
        if keys_out is None:
            keys_out = torch.zeros(len(fps), dtype=torch.int64, device="cuda")
        self.local.query_batch_keys_device(fps, keys_out, range_, index_base=self.begin)
        allreduce_best(keys_out, self.group)
        return [decode_key(int(k)) for k in keys_out.tolist()]

    def query(self, fp, range_: int = 0, key_out=None):
        """Collective: every rank calls it with the same query; returns (global index, score)."""
        if self.comm is not None:
            return self.local.query_sharded(fp, self.comm, self.begin, range_)
        import torch
        if key_out is None:
            key_out = torch.zeros(1, dtype=torch.int64, device="cuda")
        self.local.query_key_device(fp, key_out, range_, index_base=self.begin)
        allreduce_best(key_out, self.group)
        return decode_key(int(key_out.item()))
