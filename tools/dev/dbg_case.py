import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.cases import small_cases
from tests.test_gpu_parity import _hip_extract_fused
from oracle import oracle_extract, canonical_mesh
name = sys.argv[1]
g, thresh, lower, upper = small_cases()[name]
hip = _hip_extract_fused(torch.device('cuda:0'), g, thresh, lower, upper)
ref = oracle_extract(g, thresh, lower, upper)
hk, hv, hf = canonical_mesh(*hip); rk, rv, rf = canonical_mesh(*ref)
print('keys equal', np.array_equal(hk, rk), 'faces equal', np.array_equal(hf, rf))
bad = np.where(~((hv == rv) | (np.isnan(hv) & np.isnan(rv))).all(1))[0]
print('bad verts', len(bad), 'of', len(hk))
rx, ry, rz = g.shape
for i in bad[:12]:
    k = hk[i]; ax = k % 3; lin = k // 3; z = lin % rz; y = (lin // rz) % ry; x = lin // (rz * ry)
    print((x, y, z), 'axis', ax, 'hip', hv[i], 'ref', rv[i])
