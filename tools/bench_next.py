"""Measurement of the two SURVEY.md section-8(f4) rows next to the hot path: marching tetrahedra and the BVH ray caster.
One JSON line per workload (the headline bench stays bench.py).

  tetra    n^3 jittered cubes of 5 tetrahedra each, in lattice order (a structured tet grid) and shuffled (no locality at
           all: every corner gather misses), SDF = a wavy sphere.  Timed: the whole wrapper call
           `marching_tetrahedras(points, tets, sdf)` on device-resident tensors (median of K, hipEvents on the current
           stream), beside the reference's own formulation -- a chain of ATen ops (oracle/mt_torch.py) -- on the SAME GPU,
           and both results compared (faces equal, positions bit-identical).
  raycast  mesh = marching cubes of the Perlin field (scaled into the unit cube), rays = a W x H pinhole camera outside
           the cube.  Timed: BVH build (host) and `RayCaster.invoke` (median of K); a sample of rays is compared with
           the brute-force oracle.

usage: python tools/bench_next.py [--what tetra|raycast|all] [--n 128] [--grid 256] [--rays 1024] [--steps 10]
       [--order lattice|shuffled|both]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch


def median_ms(fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


def tet_lattice(n, seed, dev, shuffle=True):
    """The lattice of tests/test_gpu_tetra.py::_grid_tets, generated on the device."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    ax = torch.arange(n + 1, device=dev)
    P = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3).float()
    P += (torch.rand(P.shape, generator=g) * 0.4 - 0.2).to(dev)
    c = torch.stack(torch.meshgrid(*(torch.arange(n, device=dev),) * 3, indexing="ij"), -1).reshape(-1, 3)
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    v = [((x + dx) * (n + 1) + y + dy) * (n + 1) + z + dz for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)]
    par = ((x + y + z) % 2 == 0)[:, None]
    A = [(0, 3, 5, 6), (0, 1, 3, 5), (0, 2, 3, 6), (0, 4, 5, 6), (3, 5, 6, 7)]
    B = [(1, 2, 4, 7), (0, 1, 2, 4), (1, 2, 3, 7), (1, 4, 5, 7), (2, 4, 6, 7)]
    T = torch.cat([torch.where(par, torch.stack([v[k] for k in ta], 1), torch.stack([v[k] for k in tb], 1))
                   for ta, tb in zip(A, B)])
    if shuffle:
        T = T[torch.randperm(T.shape[0], generator=g).to(dev)]
    else:   # cube by cube, the order a structured tet grid file has
        T = T.reshape(5, -1, 4).permute(1, 0, 2).reshape(-1, 4)
    T = T.contiguous()
    sdf = ((P - n / 2).norm(dim=1) - n / 3 + 0.3 * torch.sin(P[:, 0])).contiguous()
    return P.contiguous(), T, sdf


def bench_tetra(args, shuffle):
    import primitive3d_amd as p3d
    from oracle.mt_torch import mt_torch   # baseline leg only
    dev = torch.device("cuda", 0)
    P, T, sdf = tet_lattice(args.n, 0, dev, shuffle)
    v, f = p3d.marching_tetrahedras(P, T, sdf)          # (also fixes the orientation of T in place, once)
    med, best = median_ms(lambda: p3d.marching_tetrahedras(P, T, sdf), args.steps)
    rv, rf = mt_torch(P, T, sdf)
    same = bool(torch.equal(rf, f) and torch.equal(rv, v))
    tmed, tbest = median_ms(lambda: mt_torch(P, T, sdf), max(3, args.steps // 2), warmup=1)
    nt = T.shape[0]
    # algorithmic bytes: every tet read once (32 B) + its four sdf signs; the outputs
    alg = nt * 32 + P.shape[0] * 4 + v.numel() * 4 + f.numel() * 8
    print(json.dumps({
        "metric": "marching_tetrahedras_Mtets_per_s", "value": round(nt / med / 1e3, 1), "unit": "Mtets/s",
        "ms_per_call": round(med, 4), "ms_best": round(best, 4), "dtype": "f32/i64", "data": "synthetic",
        "config": {"workload": f"{args.n}^3 jittered 5-tet cubes, {'shuffled' if shuffle else 'in lattice order'}; wavy-sphere SDF", "points": P.shape[0],
                   "tets": nt, "vertices": v.shape[0], "faces": f.shape[0]},
        "roofline": {"bound": "hbm", "achieved": round(alg / med / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(alg / med / 1e6 / 8000.0, 4), "traffic": None,
                     "note": "whole call incl. two host synchronisations; algorithmic = tets + sdf read once + outputs"},
        "aten_chain_same_gpu": {"ms_per_call": round(tmed, 3), "ms_best": round(tbest, 3),
                                "speedup": round(tmed / med, 1), "results_identical": same,
                                "what": "oracle/mt_torch.py: the reference's tensor-op formulation on this device"},
    }), flush=True)
    assert same, "HIP marching tetrahedra differ from the tensor-op chain"


def bench_raycast(args):
    import primitive3d_amd as p3d
    from primitive3d_amd.fields import perlin_grid
    from oracle.rc_oracle import raycast_oracle   # checker only
    dev = torch.device("cuda", 0)
    n = args.grid
    grid = perlin_grid((n, n, n), device=dev)
    v, f = p3d.marching_cubes(grid, 0.0, 1.0)             # mesh inside the unit cube
    t0 = time.perf_counter()
    rc = p3d.create_raycaster(v, f)
    build_s = time.perf_counter() - t0
    W = H = args.rays
    eye = torch.tensor([0.5, 0.5, -1.5], device=dev)
    u = (torch.arange(W, device=dev).float() + 0.5) / W - 0.5
    px = torch.stack(torch.meshgrid(u, u, indexing="ij"), -1).reshape(-1, 2)
    d = torch.cat([px * 0.9, torch.ones(W * H, 1, device=dev)], 1)
    d = (d / d.norm(dim=1, keepdim=True)).contiguous()
    o = eye.expand(W * H, 3).contiguous()
    depth = torch.empty(W * H, device=dev)
    nrm = torch.empty(W * H, 3, device=dev)
    ids = torch.empty(W * H, dtype=torch.int32, device=dev)
    med, best = median_ms(lambda: rc.invoke(o, d, depth, nrm, ids), args.steps)
    # sample check against brute force (bounded: sample x faces ray/triangle tests on the host)
    ns = max(16, min(512, int(4e8 // max(1, f.shape[0]))))
    sel = torch.randperm(W * H, generator=torch.Generator().manual_seed(0))[:ns].to(dev)
    od, on, oi, second = raycast_oracle(v.cpu().numpy(), f.cpu().numpy(), o[sel].cpu().numpy(), d[sel].cpu().numpy(), chunk=16)
    gd, gi = depth[sel].cpu().numpy(), ids[sel].cpu().numpy()
    clear = (second - od) > 1e-4
    ok = bool(np.allclose(gd, od, rtol=0, atol=1e-5) and np.array_equal(gi[clear], oi[clear]))
    hits = int((ids >= 0).sum())
    print(json.dumps({
        "metric": "raycast_Mrays_per_s", "value": round(W * H / med / 1e3, 1), "unit": "Mrays/s",
        "ms_per_call": round(med, 4), "ms_best": round(best, 4), "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{W}x{H} pinhole rays onto the marching-cubes mesh of the {n}^3 Perlin field",
                   "triangles": f.shape[0], "vertices": v.shape[0], "rays": W * H, "hit_fraction": round(hits / (W * H), 3)},
        "bvh_build_host_s": round(build_s, 3),
        "oracle_check": {"rays": ns, "ok": ok, "what": "brute force over all triangles (oracle/rc_oracle.py)"},
    }), flush=True)
    assert ok, "ray caster differs from the brute-force oracle on the sample"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="all", choices=["tetra", "raycast", "all"])
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--order", default="both", choices=["lattice", "shuffled", "both"])
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("tools/bench_next.py needs a GPU")
    if args.what in ("tetra", "all"):
        if args.order in ("lattice", "both"):
            bench_tetra(args, False)
        if args.order in ("shuffled", "both"):
            bench_tetra(args, True)
    if args.what in ("raycast", "all"):
        bench_raycast(args)


if __name__ == "__main__":
    main()
