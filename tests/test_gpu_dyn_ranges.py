"""The DYNAMIC plane hand-out of the streaming kernel (k_fused DYN, primitive3d_amd/csrc/range_sched.h): persistent
blocks that own ranges of planes, claim them plane by plane and steal from each other.  By default only grids with long
ranges take that path (1024^3-class); here P3D_FUSED_DYN=2 forces it on every shape that has at least 2 planes per block,
and the results must be what the fixed x-slabs give: the oracle's mesh, bit for bit.

Replaces the same reference lines as the rest of the streaming kernel (marching_cubes.cu:4-46, 70-138)."""
import numpy as np
import pytest
import torch

from oracle import oracle_extract
from tests.test_gpu_parity import _assert_same_mesh, _hip_extract_fused

pytestmark = pytest.mark.gpu


def _sphere(shape, c, r):
    X, Y, Z = np.meshgrid(*[np.arange(n, dtype=np.float32) for n in shape], indexing="ij")
    return ((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2 - r * r).astype(np.float32)


def _perlin(shape, period, seed):
    from primitive3d_amd.fields import perlin_grid
    return perlin_grid(shape, period=period, seed=seed).numpy()


CASES = {
    # (grid, thresh): rows of >= 5 chunks take the 8 x 3 tile, 3-4 chunks the 4 x 6 tile -- the two that have a DYN variant;
    # every shape has at least 2 planes per resident block (1024), or the launch falls back to fixed slabs
    "noise_600x40x512": lambda: (np.random.default_rng(1).standard_normal((600, 40, 512)).astype(np.float32), 0.0),
    "noise_800x26x330_ragged": lambda: (np.random.default_rng(2).standard_normal((800, 26, 330)).astype(np.float32), 0.3),
    "sphere_700x48x320_skewed_work": lambda: (_sphere((700, 48, 320), (60, 20, 100), 37.5), 0.0),   # most blocks idle: all steal
    "sphere_600x30x1200_three_z_tiles": lambda: (_sphere((600, 30, 1200), (250, 15, 600), 14.25), 0.0),
    "plateau_2100x13x321": lambda: (np.where(np.random.default_rng(3).random((2100, 13, 321)) < 0.5, 0.0, 1.0).astype(np.float32), 0.0),
    "perlin_1100x180x200_4x6_tile": lambda: (_perlin((1100, 180, 200), 12, 4), -0.05),
}


def _dyn_launches():
    from primitive3d_amd import capi
    return capi.debug_counters()["dynamic_launches"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_dyn_hand_out_matches_oracle(gpu, tuning_env, name):
    g, thresh = CASES[name]()
    lower, upper = [0.5, -1.0, 2.0], [3.0, 4.0, 9.0]
    tuning_env("P3D_FUSED_DYN", 2)
    before = _dyn_launches()
    hip = _hip_extract_fused(gpu, g, thresh, lower, upper)
    assert _dyn_launches() > before, "the launch fell back to fixed slabs: the case does not test the hand-out"
    _assert_same_mesh(hip, oracle_extract(g, thresh, lower, upper))


def test_dyn_and_fixed_slabs_agree_at_512_cubed(gpu, tuning_env, built):
    """The headline grid through both launch shapes: same counts, same mesh (as a sorted triangle soup of bit patterns)."""
    from bench import soup_hashes
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    g = perlin_grid(512, period=64, seed=0, device=gpu)
    out = {}
    for mode in (0, 2):
        tuning_env("P3D_FUSED_DYN", mode)
        before = _dyn_launches()
        v, f = capi.extract_fused(g, 0.0, [0.0] * 3, [512.0] * 3)
        assert (_dyn_launches() > before) == (mode == 2)
        torch.cuda.synchronize()
        out[mode] = soup_hashes(v, f)
        assert (v.shape[0], f.shape[0]) == (5223442, 10400444)
    assert torch.equal(out[0][0], out[2][0]) and torch.equal(out[0][1], out[2][1])


def test_dyn_stack_of_items(gpu, tuning_env, built):
    """A batch of grids is one stack of planes: under DYN the tile column carries the item, ranges never straddle items."""
    from bench import soup_hashes
    from primitive3d_amd.fields import perlin_grid
    shape = (416, 100, 200)
    grids = torch.stack([perlin_grid(shape, period=10 + 3 * b, seed=20 + b) for b in range(4)]).half()
    tuning_env("P3D_FUSED_DYN", 2)
    before = _dyn_launches()
    v, f, vo, fo = built.marching_cubes_batched(grids.to(gpu), 0.02)
    torch.cuda.synchronize()
    assert _dyn_launches() > before
    vo, fo = vo.tolist(), fo.tolist()
    for b in range(grids.shape[0]):
        rv, rf, _ = oracle_extract(grids[b].float().numpy(), 0.02, [0.0] * 3, [float(n) for n in shape], want_keys=False)
        vb, fb = v[vo[b]:vo[b + 1]], f[fo[b]:fo[b + 1]]
        assert tuple(vb.shape) == rv.shape and tuple(fb.shape) == rf.shape, (b, vb.shape, rv.shape)
        hg, kg = soup_hashes(vb, fb)
        ho, ko = soup_hashes(torch.from_numpy(rv).to(gpu), torch.from_numpy(rf).to(gpu))
        assert torch.equal(kg, ko) and torch.equal(hg, ho), b


def test_dyn_calls_back_to_back_reuse_cleared_tables(gpu, tuning_env, built):
    """The hand-out table of call n+1 is cleared by the streaming kernel of call n (a ring of four per stream): twelve calls
    in a row on alternating shapes, every one with the right counts."""
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    from tests.test_gpu_configs import torch_counts
    tuning_env("P3D_FUSED_DYN", 2)
    grids = [perlin_grid((384, 96, 512), period=32, seed=5, device=gpu), perlin_grid((600, 48, 320), period=16, seed=6, device=gpu)]
    want = [torch_counts(g, 0.0) for g in grids]
    before = _dyn_launches()
    for i in range(12):
        g = grids[i & 1]
        v, f = capi.extract_fused(g, 0.0, [0.0] * 3, [float(s) for s in g.shape])
        assert (v.shape[0], f.shape[0]) == want[i & 1], i
    assert _dyn_launches() >= before + 12


@pytest.mark.parametrize("knobs", [
    {"P3D_FUSED_NMID": 0},                                            # big + short slabs only (the taper of rounds 2-3)
    {"P3D_FUSED_NMID": 3, "P3D_FUSED_XT_MID": 3},                     # a few middle slabs of another length
    {"P3D_FUSED_NBIG": 2, "P3D_FUSED_XT": 9, "P3D_FUSED_XT_TAIL": 1},  # nearly everything in middle and one-plane slabs
    {"P3D_FUSED_XT": 16, "P3D_FUSED_XT_MID": 15, "P3D_FUSED_NMID": 2},
    {"P3D_FUSED_BLOCKS": 100000},                                     # one-plane slabs throughout
])
def test_fixed_slab_taper_shapes_agree(gpu, tuning_env, built, knobs):
    """The fixed-slab launch cuts x into long slabs, a middle level and short slabs (csrc/p3d_mc.hip launch_fused); every
    split must give the default's mesh (sorted triangle soup of bit patterns) -- slabs only move the block boundaries."""
    from bench import soup_hashes
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    g = perlin_grid((300, 100, 200), period=20, seed=9, device=gpu)
    up = [float(s) for s in g.shape]
    v0, f0 = capi.extract_fused(g, 0.0, [0.0] * 3, up)
    torch.cuda.synchronize()
    want = soup_hashes(v0, f0)
    for k, val in knobs.items():
        tuning_env(k, val)
    v, f = capi.extract_fused(g, 0.0, [0.0] * 3, up)
    torch.cuda.synchronize()
    got = soup_hashes(v, f)
    assert (v.shape[0], f.shape[0]) == (v0.shape[0], f0.shape[0]) and f.shape[0] > 100000
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
