"""dev: GPU + host time of ONE rank's share of the slab path (rank 0 of 2 on a 1024x512x512 grid: 512 planes + halo),
without the transport: the halo plane and the neighbour's records are local copies."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from primitive3d_amd.fields import perlin_grid
from primitive3d_amd.slab import SlabExtractor
dev = torch.device("cuda", 0)
shape = (1024, 512, 512)
ex = SlabExtractor(shape, 0, 2, dev)
ex.fill_local(lambda x0, x1: perlin_grid(shape, period=64, seed=0, device=dev, x0=x0, x1=x1))
halo = perlin_grid(shape, period=64, seed=0, device=dev, x0=512, x1=513)[0]
thresh, lower, upper = 0.0, [0.0, 0.0, 0.0], [float(s) for s in shape]
rank_counts = torch.zeros(2, dtype=torch.int64, device=dev)
def step():
    ex.phase_interior(thresh, lower, upper)
    ex.halo_recv_buffer().copy_(halo)
    be = ex.backend
    be.stream_rest(ex.grid, thresh, lower, upper, ex.shape, ex.x0, ex.has_halo)
    rank_counts[0:1].copy_(be.header_vertex_count())
    be.launch_finalize()
    nv, nf, verts, faces = be.finish_on_device(rank_counts, 0)
    return verts, faces
for _ in range(5): out = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 20
for _ in range(K): out = step()
torch.cuda.synchronize(); t1 = time.perf_counter()
print("slab rank step: %.1f us  (V=%d F=%d)" % ((t1 - t0) / K * 1e6, out[0].shape[0], out[1].shape[0]))
