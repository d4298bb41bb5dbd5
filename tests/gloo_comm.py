"""Test-only transport for cosmogp_amd.multigpu.CommBase: torch.distributed with the gloo backend, so that the
sharding / gathering / failure-agreement logic of the N > 1 path runs on CPU with world size 2.  The product's
own communicator is multigpu.Comm (RCCL inside libcosmogp_hip, no PyTorch)."""
import numpy as np

from cosmogp_amd.multigpu import CommBase


class GlooComm(CommBase):
    def __init__(self):
        import torch.distributed as dist
        self._dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.backend = 'gloo'

    def _allgather(self, block):
        import torch
        send = torch.from_numpy(np.ascontiguousarray(block, dtype=np.float64))
        recv = torch.empty((self.world,) + tuple(send.shape), dtype=send.dtype)
        self._dist.all_gather_into_tensor(recv.view(-1, *send.shape[1:]), send)
        return recv.numpy()
