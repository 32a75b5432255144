"""The multi-GPU orchestration (primitive3d_amd/slab.py: halo send/recv, count all-gather, record
exchange, global vertex ids) exercised with world_size 2 and 3 on the CPU over gloo.  The local
compute backend is a stand-in built on the oracle (tests may use the oracle; the product's
HipBackend is covered on the GPU by tests/test_gpu_slab.py)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import canonical_mesh, oracle_extract  # noqa: E402


class OracleBackend:
    """CPU stand-in for HipBackend with the same four methods."""

    def count_and_vertices(self, grid, thresh, lower, upper, full_res, x_origin, halo):
        g = grid.numpy()
        rx, ry, rz = g.shape
        n = rx - (1 if halo else 0)
        _, faces, keys = oracle_extract(g, thresh)
        lin, ax = keys // 3, keys % 3
        x, y, z = lin // (ry * rz), (lin // rz) % ry, lin % rz
        owned = x < n
        self.keys, self.faces_k = keys, keys[faces.astype(np.int64)]
        self.local_id = np.full(keys.shape, -1, np.int64)
        self.local_id[owned] = np.arange(int(owned.sum()))
        # positions with the reference arithmetic (marching_cubes.cu:105-109, :293-298) in FULL-grid terms
        d0 = g.reshape(-1)[lin]
        step = np.array([ry * rz, rz, 1])[ax]
        d1 = g.reshape(-1)[lin + step]
        dt = (np.float32(thresh) - d0) / (d1 - d0)
        pos = np.stack([(x + x_origin).astype(np.float32), y.astype(np.float32), z.astype(np.float32)], 1)
        pos[np.arange(len(ax)), ax] += dt
        lo, up = np.float32(lower), np.float32(upper)
        scale = np.array([(up[0] - lo[0]) / np.float32(full_res[0]), (up[2] - lo[1]) / np.float32(full_res[1]),
                          (up[2] - lo[2]) / np.float32(full_res[2])], np.float32)
        v = (pos * scale).astype(np.float32) + lo
        self.n, self.shape = n, g.shape
        # plane-0 records: local id of the in-plane (axis 1/2) edge starting at each voxel of plane 0
        self.plane0 = torch.full((ry, rz, 2), -1, dtype=torch.int64)
        sel = owned & (x == 0) & (ax > 0)
        self.plane0[y[sel], z[sel], ax[sel] - 1] = torch.from_numpy(self.local_id[sel])
        self.halo = torch.full((ry, rz, 2), -1, dtype=torch.int64)
        self._xyz_ax = (x, y, z, ax)
        return int(owned.sum()), len(faces), torch.from_numpy(v[owned])

    def export_first_plane_records(self):
        return self.plane0

    def halo_records_buffer(self):
        return self.halo

    def faces(self, vertex_id_base, halo_vertex_id_base):
        x, y, z, ax = self._xyz_ax
        gid = np.where(self.local_id >= 0, self.local_id + vertex_id_base, -1)
        far = self.local_id < 0
        gid[far] = self.halo.numpy()[y[far], z[far], ax[far] - 1] + halo_vertex_id_base
        assert (gid >= 0).all()
        lut = dict(zip(self.keys.tolist(), gid.tolist()))
        f = np.vectorize(lut.get, otypes=[np.int64])(self.faces_k) if self.faces_k.size else np.zeros((0, 3), np.int64)
        return torch.from_numpy(f.astype(np.int32))


def _worker(rank, world, port, out_dir, shape, thresh, lower, upper):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from primitive3d_amd.fields import perlin_grid
    from primitive3d_amd.slab import SlabExtractor
    ex = SlabExtractor(shape, rank, world, torch.device("cpu"), backend=OracleBackend())
    ex.fill_local(lambda x0, x1: perlin_grid(shape, period=10, seed=2, x0=x0, x1=x1))  # each rank synthesises its own planes
    res = ex.extract(thresh, lower, upper)
    k = ex.backend.keys[ex.backend.local_id >= 0]
    ry, rz = shape[1], shape[2]
    gk = (k // 3 + ex.x0 * ry * rz) * 3 + k % 3
    np.savez(Path(out_dir) / f"r{rank}.npz", v=res.vertices.numpy(), f=res.faces.numpy(), k=gk, base=res.vertex_base)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_orchestration_over_gloo(tmp_path, world, built):
    from primitive3d_amd.fields import perlin_grid
    shape, thresh, lower, upper = (13, 9, 20), 0.03, [0.0, 1.0, -2.0], [2.0, 3.0, 4.0]
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path), shape, thresh, lower, upper), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    bases = [int(p["base"]) for p in parts]
    assert bases == list(np.cumsum([0] + [len(p["v"]) for p in parts[:-1]]))
    hip = (np.concatenate([p["v"] for p in parts]), np.concatenate([p["f"] for p in parts]),
           np.concatenate([p["k"] for p in parts]))
    g = perlin_grid(shape, period=10, seed=2).numpy()
    hk, hv, hf = canonical_mesh(*hip)
    rk, rv, rf = canonical_mesh(*oracle_extract(g, thresh, lower, upper))
    assert np.array_equal(hk, rk) and np.array_equal(hf, rf) and np.array_equal(hv, rv)


def _overflow_worker(rank, world, port, out_dir, shape):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from primitive3d_amd.fields import perlin_grid
    from primitive3d_amd.slab import SlabExtractor
    be = OracleBackend()
    be.id_overflow = rank == 1   # only ONE rank's slab reports ambiguous ids
    ex = SlabExtractor(shape, rank, world, torch.device("cpu"), backend=be)
    ex.fill_local(lambda x0, x1: perlin_grid(shape, period=10, seed=2, x0=x0, x1=x1))
    try:
        ex.extract(0.03)
        outcome = "returned"
    except OverflowError as e:
        outcome = "OverflowError: " + str(e)
    (Path(out_dir) / f"r{rank}.txt").write_text(outcome)
    dist.barrier()   # (every rank gets here: none is stuck inside a collective of extract())
    dist.destroy_process_group()


def test_id_overflow_on_one_rank_raises_on_every_rank(tmp_path, built):
    """One rank's local extraction reports an id-space overflow (include/p3d_mc.h, p3d_mc_read_counts bit 1).  It used
    to raise on that rank alone, before the collectives: the other ranks then waited in the all-gather for ever.  The
    flag now travels with the gathered counts and EVERY rank raises after the collectives."""
    world, shape = 3, (13, 9, 20)
    port = 29500 + (os.getpid() % 2000) + 17
    mp.spawn(_overflow_worker, args=(world, port, str(tmp_path), shape), nprocs=world, join=True)
    outcomes = [(tmp_path / f"r{r}.txt").read_text() for r in range(world)]
    assert all(o.startswith("OverflowError") and "[1]" in o for o in outcomes), outcomes


def test_slab_bounds():
    from primitive3d_amd.slab import slab_bounds
    assert slab_bounds(1024, 8) == [(i * 128, (i + 1) * 128) for i in range(8)]
    b = slab_bounds(13, 3)
    assert b == [(0, 5), (5, 9), (9, 13)]
    with pytest.raises(AssertionError):
        slab_bounds(2, 3)


def test_more_than_64_ranks_are_refused_before_any_collective():
    """The device path gives one lane of a wave to every rank (p3d_mc_slab.rank_counts: ranks 0..63): a larger world is
    refused in the constructor, on every rank alike, not inside an extraction that other ranks are already waiting in."""
    from primitive3d_amd.slab import SlabExtractor
    with pytest.raises(ValueError, match="at most 64 ranks"):
        SlabExtractor((130, 4, 70), 3, 65, torch.device("cpu"))
    SlabExtractor((130, 4, 70), 3, 65, torch.device("cpu"), backend=object())   # (a host-path backend: no such limit)
