"""Utterance-level data parallelism: one process per GPU.

Utterances are independent for MFCC, scoring and every trellis, so they are block-partitioned
over ranks and nothing is exchanged inside the time recursion.  Exactly two collectives exist:
the all-gather of decoded state paths and the sum all-reduce of packed Baum-Welch statistics.
On GPUs both run on RCCL over xGMI inside liblnasr_hip.so (`Context.comm_init`); the ranks find
each other through `lnasr.rendezvous` (loopback TCP, no framework).  `TcpComm` carries the same two
calls over that rendezvous group on the CPU so that the sharding logic can be rehearsed without
GPUs (`bench.py --dry-launch`, tests); the world-2 `gloo` test keeps its own transport under tests/.
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous block partition: ranks < n_items % world get one extra item."""
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_items, world):
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


class RcclComm:
    """The product transport: RCCL communicator owned by a lnasr Context."""

    def __init__(self, ctx, rank, world, unique_id):
        self.ctx, self.rank, self.world = ctx, rank, world
        ctx.comm_init(unique_id, rank, world)

    def allgather_u32(self, local):
        return self.ctx.allgather_u32(local)

    def allreduce_sum_f64(self, buf):
        return self.ctx.allreduce_f64(buf)


class TcpComm:
    """CPU transport over a lnasr.rendezvous.TcpGroup -- rehearsals and tests only (numpy, no framework)."""

    def __init__(self, group):
        self.group, self.rank, self.world = group, group.rank, group.world

    def allgather_u32(self, local):
        parts = self.group.allgather(np.ascontiguousarray(local, dtype=np.uint32))
        return np.stack(parts).astype(np.uint32)

    def allreduce_sum_f64(self, buf):
        parts = self.group.allgather(np.ascontiguousarray(buf, dtype=np.float64))
        buf[...] = np.sum(np.stack(parts), axis=0)          # same order on every rank: identical bits everywhere
        return buf


def gather_paths(comm, local_paths, n_total, T):
    """local_paths [U_local, T] (equal-size shards are sent as they are; ragged shards are padded to the
    largest shard) -> [n_total, T] on every rank, in global utterance order."""
    sizes = shard_sizes(n_total, comm.world)
    umax = max(sizes)
    local = np.asarray(local_paths, dtype=np.uint32)
    if local.shape[0] < umax:
        local = np.concatenate([local, np.zeros((umax - local.shape[0], T), dtype=np.uint32)])
    allp = np.asarray(comm.allgather_u32(local)).reshape(comm.world, umax, T)
    return np.concatenate([allp[r, :sizes[r]] for r in range(comm.world)], axis=0)


def allreduce_stats(comm, stats):
    """Sum the packed sufficient-statistics vector over ranks (in place)."""
    return comm.allreduce_sum_f64(np.ascontiguousarray(stats, dtype=np.float64))
