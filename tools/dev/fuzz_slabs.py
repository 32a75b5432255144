"""dev: randomized fuzz of the slab path (all ranks in one process) against the oracle of the whole grid."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from oracle import oracle_extract
from primitive3d_amd import capi
from primitive3d_amd.slab import SlabExtractor
from tests.test_gpu_parity import _assert_same_mesh
from tests.ws_keys import vertex_keys_from_workspace
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
gpu = torch.device("cuda", 0)
for it in range(int(os.environ.get("N", "25"))):
    world = int(rng.integers(2, 5))
    rx = int(rng.integers(world, 90)); ry = int(rng.integers(2, 30)); rz = int(rng.choice([rng.integers(2, 70), rng.integers(100, 300), rng.integers(513, 650)]))
    x, y, z = np.meshgrid(np.arange(rx), np.arange(ry), np.arange(rz), indexing="ij")
    g = (np.sin(x * 0.45) + np.cos(y * 0.5) + np.sin(z * 0.21) + 0.2 * rng.standard_normal((rx, ry, rz))).astype(np.float32)
    thresh, lower, upper = float(rng.uniform(-0.3, 0.3)), [0.5, -1.0, 2.0], [3.0, 4.0, 9.0]
    full = torch.from_numpy(g).to(gpu)
    exs = [SlabExtractor(g.shape, r, world, gpu) for r in range(world)]
    for e in exs: e.fill_local(lambda x0, x1: full[x0:x1])
    for e in exs: e.phase_interior(thresh, lower, upper)
    for r in range(world - 1): exs[r].halo_recv_buffer().copy_(exs[r + 1].halo_send_buffer())
    counts = [e.phase_extract(thresh, lower, upper) for e in exs]
    for r in range(world - 1): exs[r].records_recv_buffer().copy_(exs[r + 1].records_send_buffer())
    rank_counts = torch.tensor([c[0] for c in counts], dtype=torch.int64, device=gpu)
    dev_path = bool(rng.integers(0, 2))
    outs = []
    for e in exs:
        if dev_path:
            outs.append((e._verts, e.backend.faces_from_rank_counts(rank_counts, e.rank)))
        else:
            r = e.phase_faces(counts); outs.append((r.vertices, r.faces))
    torch.cuda.synchronize()
    allv, allf, allk = [], [], []
    for e, (v, f) in zip(exs, outs):
        lshape = tuple(e.grid.shape)
        k = vertex_keys_from_workspace(e.backend._ws.cpu().numpy(), lshape, v.shape[0], capi.debug_layout(*lshape), halo_last_plane=e.has_halo)
        allk.append((k // 3 + e.x0 * ry * rz) * 3 + k % 3); allv.append(v.cpu().numpy()); allf.append(f.cpu().numpy())
    _assert_same_mesh((np.concatenate(allv), np.concatenate(allf), np.concatenate(allk)), oracle_extract(g, thresh, lower, upper))
print("slab fuzz ok")
