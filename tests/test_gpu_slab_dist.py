"""The multi-GPU call itself (SlabExtractor.extract with the HIP backend: collectives enqueued before the finalize
kernels, id bases derived on the device from the all-gathered vertex counts) run by TWO processes that share the one
GPU of the test box (2, 3 and 8 processes), gloo standing in for RCCL as transport.  The merged mesh must equal the oracle's mesh of the whole
grid."""
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

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir, shape, thresh, lower, upper, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    index = rank if backend == "nccl" else 0   # RCCL: one GPU per rank; gloo: all ranks share the one GPU
    torch.cuda.set_device(index)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", index))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    from primitive3d_amd.slab import SlabExtractor
    from tests.ws_keys import vertex_keys_from_workspace
    dev = torch.device("cuda", index)
    ex = SlabExtractor(shape, rank, world, dev)
    # (generated on the CPU like the oracle's input: the device generator may differ in the last bit)
    ex.fill_local(lambda x0, x1: perlin_grid(shape, period=12, seed=5, x0=x0, x1=x1).to(dev))
    ex.trace = True
    for _ in range(2):  # the second call reuses the size hints and the cursor ring
        res = ex.extract(thresh, lower, upper)
    torch.cuda.synchronize()
    res.check_total()  # the deferred int32 guard of the device path
    phases = ex.phase_times_ms()
    assert all(t >= 0 for t in phases.values())
    # every phase mark of extract()'s device path exists: each line of it that CAN run on this box has run
    for name in ("interior planes streamed", "halo plane received (wait)", "last planes streamed + record export",
                 "all-gather of vertex counts", "face count + early vertex copy", "halo records received (wait)",
                 "faces + rest of vertex copy"):
        assert name in phases, (name, sorted(phases))
    if backend == "nccl" and world == 1:
        # RCCL's point-to-point path itself, on the one GPU: the batched isend / irecv pair extract() posts between
        # neighbours, with THIS rank as the peer (RCCL runs a grouped send + recv to self), on the very views it hands over
        # -- a plane of the grid, a plane of vertex-id records inside the workspace -- while a kernel of the caller runs;
        # then the 24-byte all_gather_into_tensor straight out of the workspace header
        plane, recs = ex.grid[0], ex.backend._plane_view(0)
        got_plane, got_recs = torch.zeros_like(plane), torch.zeros_like(recs)
        works = dist.batch_isend_irecv([dist.P2POp(dist.isend, plane, 0), dist.P2POp(dist.irecv, got_plane, 0),
                                        dist.P2POp(dist.isend, recs, 0), dist.P2POp(dist.irecv, got_recs, 0)])
        busy = torch.randn(1024, 1024, device=dev) @ torch.randn(1024, 1024, device=dev)
        for w in works:
            w.wait()
        gathered = torch.empty((world, 3), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(gathered.view(-1), ex.backend.header_words())
        torch.cuda.synchronize()
        assert torch.equal(got_plane, plane) and torch.equal(got_recs, recs) and busy.isfinite().all()
        assert int(gathered[0, 0]) == res.vertices.shape[0]
    lshape = tuple(ex.grid.shape)
    k = vertex_keys_from_workspace(ex.backend._ws.cpu().numpy(), lshape, res.vertices.shape[0],
                                   capi.debug_layout(*lshape), halo_last_plane=ex.has_halo)
    ry, rz = shape[1], shape[2]
    gk = (k // 3 + ex.x0 * ry * rz) * 3 + k % 3
    np.savez(Path(out_dir) / f"r{rank}.npz", v=res.vertices.cpu().numpy(), f=res.faces.cpu().numpy(), k=gk,
             base=res.vertex_base)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,shape", [(2, (61, 21, 150)), (3, (61, 21, 150)), (8, (61, 21, 150)),
                                         (8, (208, 21, 150))])   # 8 ranks: thin slabs, and slabs streamed in two parts
def test_distributed_extract_on_one_gpu(tmp_path, gpu, world, shape):
    _run_and_compare(tmp_path, world, shape, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (this box has one)")
@pytest.mark.parametrize("shape", [(61, 21, 150), (208, 21, 150)])
def test_distributed_extract_over_rccl(tmp_path, gpu, shape):
    """The same check with the real transport: backend "nccl" (= RCCL), one GPU per rank, as many ranks as the node has
    GPUs (up to 8).  Skipped on the single-GPU test box; runs wherever a multi-GPU node executes `pytest -m gpu`."""
    _run_and_compare(tmp_path, min(8, torch.cuda.device_count()), shape, "nccl")


@pytest.mark.parametrize("shape", [(61, 21, 150), (208, 40, 150)])
def test_device_path_over_rccl_with_one_rank(tmp_path, gpu, shape):
    """RCCL itself, executed on the one GPU of the test box: a `world_size = 1` process group of the "nccl" backend in a
    spawned child, SlabExtractor.extract() through the DEVICE path -- `all_gather_into_tensor` of the three header words
    straight out of the workspace, the side-stream int32 / overflow guard with its `record_stream`, the face launch
    taking its id bases from the gathered tensor -- twice (the second call takes the size hints), result == the oracle's
    mesh of the whole grid; every phase mark of extract() must exist; then RCCL's batched isend / irecv pair with the rank
    itself as the peer on a grid plane and on a record plane of the workspace (the views extract() posts between
    neighbours).  (Two ranks need two GPUs: the test above.)"""
    _run_and_compare(tmp_path, 1, shape, "nccl")


def _run_and_compare(tmp_path, world, shape, backend):
    from primitive3d_amd.fields import perlin_grid
    thresh, lower, upper = 0.02, [0.5, -1.0, 2.0], [3.0, 4.0, 9.0]
    port = 29600 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path), shape, thresh, lower, upper, backend), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert [int(p["base"]) for p in parts] == list(np.cumsum([0] + [len(p["v"]) for p in parts[:-1]]))
    hip = (np.concatenate([p["v"] for p in parts]), np.concatenate([p["f"] for p in parts]),
           np.concatenate([p["k"] for p in parts]))
    g = perlin_grid(shape, period=12, seed=5).numpy()
    hk, hv, hf = canonical_mesh(*hip)
    rk, rv, rf = canonical_mesh(*oracle_extract(g, thresh, lower, upper))
    assert hk.shape == rk.shape and hf.shape == rf.shape, (hk.shape, rk.shape, hf.shape, rf.shape)
    assert np.array_equal(hk, rk), "vertex edge keys differ"
    assert np.array_equal(hv, rv), "vertex positions differ"
    assert np.array_equal(hf, rf), "faces differ"
