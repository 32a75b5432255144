"""dev: who is wrong when the chunk-prefix path and the P3D_NO_CHUNK_PRE path disagree?  Stack of 20 x (130,256,256) fp16
(1360 chunks): every combination of {prefix on/off} x {face kernel v1/v2}, 6 runs each; the soups are compared with each
other and item 3 / item 17 with the CPU oracle."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
import primitive3d_amd as p3d
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
from oracle import oracle_extract
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
from test_gpu_configs import soup_hashes

dev = torch.device("cuda")
grids = torch.stack([perlin_grid((130, 256, 256), period=32, seed=s, device=dev).half() for s in range(20)])
ref = {}
for b in (3, 17):
    rv, rf, _ = oracle_extract(grids[b].float().cpu().numpy(), 0.0, threads=0, want_keys=False)
    ref[b] = soup_hashes(torch.from_numpy(rv).to(dev), torch.from_numpy(rf).to(dev))


def setenv(k, v):
    if v is None:
        os.environ.pop(k, None)
    else:
        os.environ[k] = v
    capi.reload_tuning()


first = None
for nopre in ("1", None):
    for fv in ("1", "2"):
        setenv("P3D_NO_CHUNK_PRE", nopre)
        setenv("P3D_FACES_V", fv)
        bad_items, bad_whole = 0, 0
        for it in range(6):
            v, f, vo, fo = p3d.marching_cubes_batched(grids, 0.0)
            torch.cuda.synchronize()
            h = soup_hashes(v, f)
            if first is None:
                first = h
            if not torch.equal(h, first):
                bad_whole += 1
            for b in (3, 17):
                hb = soup_hashes(v[vo[b]:vo[b + 1]], f[fo[b]:fo[b + 1]])
                if hb.shape != ref[b].shape or not torch.equal(hb, ref[b]):
                    bad_items += 1
                    # how many faces of the item are wrong, and where
                    fb = f[fo[b]:fo[b + 1]].long()
                    nvb = int(vo[b + 1] - vo[b])
                    oob = int(((fb < 0) | (fb >= nvb)).any(1).sum())
                    print(f"   item {b} run {it}: faces {fb.shape[0]} out-of-range rows {oob}")
        print(f"nopre={nopre} faces_v={fv}: whole-soup mismatches vs first run {bad_whole}/6, item-vs-oracle mismatches {bad_items}/12", flush=True)
