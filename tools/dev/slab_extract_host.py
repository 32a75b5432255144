"""dev: SlabExtractor.extract() of rank 0 of 2 with the collectives stubbed out (halo / records = local copies, the
all-gather = a device copy): wall time per step and, under a kernel trace, the gaps the host leaves on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import torch.distributed as dist
from primitive3d_amd.fields import perlin_grid
from primitive3d_amd.slab import SlabExtractor
dev = torch.device("cuda", 0)
shape = (1024, 512, 512)
ex = SlabExtractor(shape, 0, 2, dev)
ex.fill_local(lambda x0, x1: perlin_grid(shape, period=64, seed=0, device=dev, x0=x0, x1=x1))
halo = perlin_grid(shape, period=64, seed=0, device=dev, x0=512, x1=513)[0]
ex.halo_recv_buffer().copy_(halo)
rec_fake = None


class _Op:
    def __init__(self, op, tensor, peer):
        self.tensor = tensor


def _batch(ops):
    return []


def _all_gather(out, inp, async_op=False):
    out.zero_()
    out.view(-1)[:inp.numel()].copy_(inp.view(-1))   # (a rank contributes several header words since round 3)


dist.get_backend = lambda: "nccl"
dist.P2POp = _Op
dist.batch_isend_irecv = _batch
dist.all_gather_into_tensor = _all_gather
dist.isend = dist.irecv = None
thresh, lower, upper = 0.0, [0.0, 0.0, 0.0], [float(s) for s in shape]
for _ in range(5): res = ex.extract(thresh, lower, upper)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 20
for _ in range(K): res = ex.extract(thresh, lower, upper)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("extract() with stubbed collectives: %.1f us per step (V=%d F=%d)" % ((t1 - t0) / K * 1e6, res.vertices.shape[0], res.faces.shape[0]))
