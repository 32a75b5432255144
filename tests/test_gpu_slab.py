"""Multi-GPU slab path validated on ONE device: all ranks of the algorithm run sequentially in one
process (primitive3d_amd.slab.extract_in_process), tensor copies standing in for the RCCL send/recv
pairs.  The merged mesh must equal the oracle's mesh of the whole grid."""
import numpy as np
import pytest
import torch

from oracle import canonical_mesh, oracle_extract
from tests.test_gpu_parity import _assert_same_mesh
from tests.ws_keys import vertex_keys_from_workspace

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("device_bases", [False, True])
@pytest.mark.parametrize("world", [2, 3, 4])
def test_slabs_merge_to_the_whole_grid_mesh(gpu, world, device_bases):
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    from primitive3d_amd.slab import SlabExtractor, slab_bounds
    g = perlin_grid((61, 21, 150), period=12, seed=5).numpy()   # 61 planes: world 2 gets slabs long enough to be split
    thresh, lower, upper = 0.02, [0.5, -1.0, 2.0], [3.0, 4.0, 9.0]
    full = torch.from_numpy(g).to(gpu)
    shape = g.shape
    exs = [SlabExtractor(shape, r, world, gpu) for r in range(world)]
    for e in exs:
        e.fill_local(lambda x0, x1: full[x0:x1])
    splits = [e.phase_interior(thresh, lower, upper) for e in exs]  # interior planes before the halo arrives
    for r in range(world - 1):
        exs[r].halo_recv_buffer().copy_(exs[r + 1].halo_send_buffer())
    counts = [e.phase_extract(thresh, lower, upper) for e in exs]
    for r in range(world - 1):
        exs[r].records_recv_buffer().copy_(exs[r + 1].records_send_buffer())
    if device_bases:  # the id bases come from the all-gathered counts ON THE DEVICE (p3d_mc_slab.rank_counts)
        from primitive3d_amd.slab import SlabResult
        rank_counts = torch.tensor([c[0] for c in counts], dtype=torch.int64, device=gpu)
        res = [SlabResult(e._verts, e.backend.faces_from_rank_counts(rank_counts, e.rank), rank=e.rank,
                          rank_counts=rank_counts) for e in exs]
    else:
        res = [e.phase_faces(counts) for e in exs]
    torch.cuda.synchronize()

    rx, ry, rz = shape
    allv, allf, allk = [], [], []
    for e, out in zip(exs, res):
        lshape = tuple(e.grid.shape)
        ws = e.backend._ws.cpu().numpy()
        k = vertex_keys_from_workspace(ws, lshape, out.vertices.shape[0], capi.debug_layout(*lshape),
                                       halo_last_plane=e.has_halo)
        lin, ax = k // 3, k % 3
        allk.append((lin + e.x0 * ry * rz) * 3 + ax)  # local voxel index -> global
        allv.append(out.vertices.cpu().numpy())
        allf.append(out.faces.cpu().numpy())
        assert out.vertex_base == sum(c[0] for c in counts[:e.rank])
        assert int(e.backend.header_vertex_count()[0]) == counts[e.rank][0]  # what the all-gather would read
    hip = (np.concatenate(allv), np.concatenate(allf), np.concatenate(allk))
    _assert_same_mesh(hip, oracle_extract(g, thresh, lower, upper))


def test_single_rank_slab_equals_plain_call(gpu):
    from primitive3d_amd.slab import extract_in_process
    g = np.random.default_rng(3).standard_normal((9, 8, 70)).astype(np.float32)
    (out,) = extract_in_process(torch.from_numpy(g).to(gpu), 1, 0.0)
    rv, rf, _ = oracle_extract(g, 0.0)
    assert out.vertices.shape == rv.shape and out.faces.shape == rf.shape
