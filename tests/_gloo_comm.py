"""torch.distributed (gloo) transport for the world-2 CPU test of the sharding logic -- test infrastructure only; the
package itself is framework-free (lnasr/dist.py: RcclComm on GPUs, TcpComm for rehearsals)."""
import numpy as np


class GlooComm:
    """CPU transport over an initialised torch.distributed (gloo) group."""

    def __init__(self, rank, world):
        import torch.distributed as dist
        self.dist, self.rank, self.world = dist, rank, world

    def allgather_u32(self, local):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(local).astype(np.int64))
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return np.stack([o.numpy() for o in outs]).astype(np.uint32)

    def allreduce_sum_f64(self, buf):
        import torch
        t = torch.from_numpy(buf)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return buf
