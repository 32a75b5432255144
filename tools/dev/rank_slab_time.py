"""dev: bench.py's c4_rank_slab workload alone (one rank's share of the 8-GPU 1024^3 run, transport stubbed); the launch
knobs come from the environment (P3D_FUSED_XT, P3D_FUSED_BLOCKS)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
r = bench.rank_slab_workload(capi, perlin_grid, torch.device("cuda", 0), steps=int(os.environ.get("STEPS", "20")), rank=int(os.environ.get("RANK_", "3")), hold=int(os.environ.get("HOLD", "2")))
print("XT=%s BLOCKS=%s  %.4f ms  phases %s" % (os.environ.get("P3D_FUSED_XT", "-"), os.environ.get("P3D_FUSED_BLOCKS", "-"), r["ms_per_step"], r["phases_ms_last_step"]))
# per-stage hipEvent times of one more extraction (the stubs are gone: drive the phases by hand as bench does)
if os.environ.get("STAGES"):
    import torch.distributed as dist
    from primitive3d_amd.slab import SlabExtractor
    import time
    world, rank, shape = 8, 3, (1024, 1024, 1024)
    ex = SlabExtractor(shape, rank, world, torch.device("cuda", 0))
    ex.fill_local(lambda x0, x1: perlin_grid(shape, period=64, seed=0, device="cuda", x0=x0, x1=x1))
    lower, upper = [0.0] * 3, [1024.0] * 3
    be = ex.backend
    rc = torch.zeros((world, 3), dtype=torch.int64, device="cuda")
    capi.profile_enable(2)
    for it in range(4):
        ex.phase_interior(0.0, lower, upper)
        be.stream_rest(ex.grid, 0.0, lower, upper, ex.shape, ex.x0, ex.has_halo)
        sb = ex.records_send_buffer()
        rc[rank].copy_(be.header_words())
        be.launch_finalize()
        nv, nf, v, f = be.finish_on_device(rc, rank)
        torch.cuda.synchronize()
        st = capi.profile_read()
    print({k: round(t * 1e3, 1) for k, t in st.items()}, "us; V", nv, "F", nf)
