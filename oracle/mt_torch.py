"""The tensor-op formulation of marching tetrahedra (what prim3d/utility/marching_tetrahedras.py:147-234 runs: the
reference is pure PyTorch, so on a ROCm box its work IS this chain of ATen ops), restated from oracle/mt_oracle.py op
for op with torch instead of numpy.  TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/__init__.py): the parity tests
check it against the reference-made goldens (tests/test_tetra_cpu.py) and tools/bench_next.py times it on the GPU
beside the HIP library as "the reference's op chain on this device".  Never imported by the product.
"""
import torch

from .mt_oracle import BASE_TET_EDGES, NUM_TRIANGLES, TRIANGLE_TABLE


def mt_torch(vertices: torch.Tensor, tets: torch.Tensor, sdf: torch.Tensor, return_tet_idx: bool = False):
    """vertices [N,3] f32, tets [T,4] i64 (orientation fixed IN PLACE, :147-148), sdf [N] f32 on any device."""
    dev = vertices.device
    table = torch.as_tensor(TRIANGLE_TABLE, device=dev)
    ntab = torch.as_tensor(NUM_TRIANGLES, device=dev)
    base = torch.as_tensor(BASE_TET_EDGES, device=dev)
    with torch.no_grad():
        m = torch.cat([torch.ones(tets.shape + (1,), device=dev, dtype=torch.float64), vertices[tets].double()], -1)
        flip = torch.linalg.det(m) < 0                                   # :147 (float64 here, see mt_oracle.py)
        tets[flip, :2] = tets[flip][:, [1, 0]]                           # :148
        occ = sdf > 0                                                    # :151
        occ4 = occ[tets.reshape(-1)].reshape(-1, 4)
        s = occ4.sum(-1)
        valid = (s > 0) & (s < 4)                                        # :153-154
        edges = tets[valid][:, base].reshape(-1, 2)                      # :157
        edges = torch.stack([edges.min(1).values, edges.max(1).values], 1)   # :67-83
        uniq, inv = torch.unique(edges, dim=0, return_inverse=True)      # :160
        cross = occ[uniq.reshape(-1)].reshape(-1, 2).sum(-1) == 1        # :163-164
        mapping = torch.full((uniq.shape[0],), -1, dtype=torch.long, device=dev)
        mapping[cross] = torch.arange(int(cross.sum()), device=dev)      # :165-170
        idx_map = mapping[inv].reshape(-1, 6)                            # :171
        pairs = uniq[cross]
    p = vertices[pairs.reshape(-1)].reshape(-1, 2, 3)                    # :178
    sd = sdf[pairs.reshape(-1)].reshape(-1, 2, 1).clone()                # :179
    sd[:, -1] *= -1                                                      # :180
    den = sd.sum(1, keepdim=True)                                        # :182
    sd = torch.flip(sd, [1]) / den                                       # :184
    verts = (p * sd).sum(1)                                              # :185
    with torch.no_grad():
        case = (occ4[valid] * (2 ** torch.arange(4, device=dev))).sum(-1)    # :194-195
        ntri = ntab[case]
        one, two = ntri == 1, ntri == 2
        faces = torch.cat([torch.gather(idx_map[one], 1, table[case[one]][:, :3]).reshape(-1, 3),
                           torch.gather(idx_map[two], 1, table[case[two]][:, :6]).reshape(-1, 3)], 0)   # :205-224
        if return_tet_idx:
            tid = torch.arange(tets.shape[0], device=dev)[valid]         # :226-234
            return verts, faces, torch.cat([tid[one], tid[two].repeat_interleave(2)], 0)
    return verts, faces
