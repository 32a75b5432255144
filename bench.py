#!/usr/bin/env python3
"""bench.py -- headline benchmark of the marching-cubes hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU; started
                                                         WITHOUT a rank environment it starts those ranks itself as a child
                                                         `python -m torch.distributed.run ... bench.py` and relays its result)

A "step" is one whole call of the drop-in boundary (`libPrim3D.marching_cubes`: classify+count ->
host read of V,F -> allocation -> vertex + face emission incl. the scale/offset epilogue) on a
device-resident fp32 grid.  Workload per rank is fixed at 2^27 voxels (weak scaling):
  N=1  512^3           single-octave Perlin SDF, period 64, iso 0  (BASELINE.json configs[2])
  N=2  1024x512x512    axis-0 slabs, one halo plane per boundary over RCCL
  N=4  1024x1024x512
  N=8  1024^3          (BASELINE.json configs[3])
Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant
kernel (hipEvent-timed inside the library, on the launch stream) and `cpu_baseline` (the CPU oracle
restatement of the reference kernels on all host cores and on one thread, an `import mcubes` attempt, and the
whole-mesh comparison with the GPU's result; N=1 only).  `--config c2|c4|c5` selects the other single-GPU workloads.
The N>1 line also carries what RCCL saw (`config.rccl`: backend, world size, every rank's device) and, measured by rank 0
in the same run outside the timed region, the same whole volume through the plain single-GPU call (`full_volume_1gpu`,
`speedup_vs_1gpu`; at N=8 also `c4_1gpu_ms` / `speedup_vs_1gpu_1024`: the north-star's ">= 6x at 8 GPUs on 1024^3").
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak

SHAPES = {1: (512, 512, 512), 2: (1024, 512, 512), 4: (1024, 1024, 512), 8: (1024, 1024, 1024)}


def soup_hashes(v, f):
    """Order-free fingerprint of a mesh on the device: one 64-bit hash per triangle of the BIT PATTERNS of its nine
    coordinates in corner order (winding kept), sorted; plus the sorted vertex rows themselves."""
    import torch
    bits = v.contiguous().view(torch.int32)[f.long()].reshape(-1, 9).long() & 0xFFFFFFFF
    h = torch.zeros(bits.shape[0], dtype=torch.int64, device=v.device)
    for k in range(9):
        h = (h ^ bits[:, k]) * -7046029254386353131 + (k + 1)
        h = h ^ (h >> 29)
    vb = v.contiguous().view(torch.int32).long() & 0xFFFFFFFF
    vkey = torch.sort((vb[:, 0] * 0x9E3779B1 + vb[:, 1]) * 0x85EBCA6B + vb[:, 2]).values
    return torch.sort(h).values, vkey


def cpu_baseline(grid, thresh, lower, upper, out_v, out_f, budget=(8.0, 6.0)):
    """The CPU leg (rank 0, N=1): the oracle restatement of the reference kernels (`kind: "port"`) timed on the host
    cores next to the GPU number -- one thread and all cores (OpenMP over axis-0 planes) -- and, when PyMCubes is
    importable, the reference's own CPU path `mcubes.marching_cubes` (marching_cubes.py:74; `kind: "reference"`).
    The GPU mesh of the last timed call is compared with the oracle's as a WHOLE (sorted triangle soups, positions
    bit for bit), not only by its counts.  A baseline, not a target."""
    import numpy as np
    from oracle import oracle_count, oracle_extract
    rx, ry, rz = grid.shape
    nvox = rx * ry * rz
    g = grid.cpu().numpy()
    counts = oracle_count(g, thresh)
    # bounded sample: whole extractions of the same grid, ~8 s of single-thread work, then ~6 s on all cores (default)
    reps1, c0 = 0, time.perf_counter()
    while True:
        ov, of, _ = oracle_extract(g, thresh, lower, upper, counts=counts, want_keys=False)
        reps1 += 1
        c1 = time.perf_counter()
        if c1 - c0 >= budget[0] or reps1 >= 8:
            break
    t1 = (c1 - c0) / reps1
    repsn, c0 = 0, time.perf_counter()
    while True:
        mv, mf, _ = oracle_extract(g, thresh, lower, upper, threads=0, counts=counts, want_keys=False)
        repsn += 1
        c1 = time.perf_counter()
        if c1 - c0 >= budget[1] or repsn >= 32:
            break
    tn = (c1 - c0) / repsn
    nthreads = int(getattr(oracle_extract, "last_threads", 1))
    import torch
    gv, gf = out_v, out_f
    assert tuple(ov.shape) == tuple(gv.shape) and tuple(of.shape) == tuple(gf.shape), \
        ("GPU/CPU count mismatch", ov.shape, of.shape, gv.shape, gf.shape)
    assert np.array_equal(mv, ov) and np.array_equal(mf, of), "OpenMP oracle differs from the serial oracle"
    hg, kg = soup_hashes(gv, gf)
    ho, ko = soup_hashes(torch.from_numpy(ov).to(gv.device), torch.from_numpy(of).to(gv.device))
    assert torch.equal(kg, ko), "GPU vertex positions differ from the CPU oracle's"
    assert torch.equal(hg, ho), "GPU mesh differs from the CPU oracle's (triangle soup)"
    res = {"value": round(nvox / tn / 1e6, 2), "unit": "Mvoxels/s", "cores": nthreads, "kind": "port",
           "sample": f"{repsn} full extractions of the same {rx}x{ry}x{rz} grid by oracle/mc_oracle.c with OpenMP on "
                     f"{nthreads} threads ({tn:.2f} s each); single thread: {reps1} extractions, {t1:.2f} s each = "
                     f"{nvox / t1 / 1e6:.1f} Mvoxels/s; host has {os.cpu_count()} cores; whole GPU mesh "
                     f"(V={gv.shape[0]}, F={gf.shape[0]}) compared bit for bit with the oracle's as a sorted triangle soup",
           "single_thread_value": round(nvox / t1 / 1e6, 2)}
    try:  # the reference's own CPU branch is third-party PyMCubes (single-threaded): timed as examples/sphere.py:22-23
        import mcubes
        c0 = time.perf_counter()
        pv, pf = mcubes.marching_cubes(g, thresh)
        tm = time.perf_counter() - c0
        res["pymcubes"] = {"value": round(nvox / tm / 1e6, 2), "unit": "Mvoxels/s", "cores": 1, "kind": "reference",
                           "vertices": int(pv.shape[0]), "faces": int(pf.shape[0])}
    except Exception as e:  # not installed in this image
        res["pymcubes"] = f"unavailable ({type(e).__name__})"
    return res


def other_configs(p3d, capi, perlin_grid, dev):
    """The other single-GPU workloads of BASELINE.json (configs[1], [3] on one GPU, [4]) for a few steps each, AFTER the
    headline measurement and outside its timed region: same call pattern (back-to-back calls between two
    synchronisations), `k_fused` timed on the steps that follow by dispatch-attached hipEvents.  No CPU leg here (the
    parity of these configurations at full size is tests/test_gpu_configs.py).  A few seconds in all."""
    import numpy as np
    import torch
    out = {}

    def measure(name, workload, step, nvox, sizeof, steps, warmup=3):
        for _ in range(warmup):
            res = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        capi.profile_enable(1)
        dom = []
        for _ in range(3):
            res = step()
            st = capi.profile_read()
            dom.append(st.get("k_fused", float("nan")))
        torch.cuda.synchronize()
        capi.profile_enable(0)
        kms = sum(dom) / len(dom)
        alg = nvox * sizeof
        out[name] = {"workload": workload, "steps": steps, "ms_per_step": round(ms, 4),
                     "value": round(nvox / (ms * 1e-3) / 1e6, 1), "unit": "Mvoxels/s",
                     "dtype": "f16" if sizeof == 2 else "f32", "vertices": int(res[0].shape[0]), "faces": int(res[1].shape[0]),
                     "k_fused_ms": round(kms, 4), "k_fused_frac": round(alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "whole_call_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    b66 = torch.from_numpy(np.load(ROOT / "tests" / "golden" / "bunny66.npy"))
    g2 = torch.nn.functional.interpolate(b66[None, None], size=(256,) * 3, mode="trilinear", align_corners=True)[0, 0]
    g2 = g2.contiguous().to(dev)
    measure("c2", "256x256x256 fp32 bunny SDF (examples/data/bunny.npy trilinearly resampled from 66^3), iso 0",
            lambda: p3d.libPrim3D.marching_cubes(g2, 0.0, [0.0] * 3, [256.0] * 3), 256 ** 3, 4, steps=40, warmup=5)
    del g2
    # the reference's OWN example inputs at their own sizes: examples/sphere.py:8-9 verbatim (200^3,
    # V = 11766 / F = 23528: marching_cubes.cu on this field) and examples/data/bunny.npy as it is
    # (66^3, 13282 / 26560).  Tiny: a call is launch latency and ramp, not bandwidth
    from primitive3d_amd.fields import sphere_grid
    gs = torch.tensor(sphere_grid(200)).to(dev).to(torch.float32).contiguous()
    measure("sphere200", "200x200x200 sphere field of examples/sphere.py:8-9 (centre 50, radius 25; int64 in the example, float32 "
                         "on the device as its wrapper makes it, marching_cubes.py:87), iso 0",
            lambda: p3d.libPrim3D.marching_cubes(gs, 0.0, [0.0] * 3, [200.0] * 3), 200 ** 3, 4, steps=40, warmup=5)
    assert (out["sphere200"]["vertices"], out["sphere200"]["faces"]) == (11766, 23528), out["sphere200"]
    del gs
    gb = b66.to(dev).to(torch.float32).contiguous()
    measure("bunny66", "66x66x66 fp32 bunny SDF (examples/data/bunny.npy as shipped, examples/bunny_sdf.py), iso 0",
            lambda: p3d.libPrim3D.marching_cubes(gb, 0.0, [0.0] * 3, [66.0] * 3), 66 ** 3, 4, steps=40, warmup=5)
    assert (out["bunny66"]["vertices"], out["bunny66"]["faces"]) == (13282, 26560), out["bunny66"]
    del gb
    g5 = torch.stack([perlin_grid((256,) * 3, period=64, seed=s, device=dev).half() for s in range(32)])
    measure("c5", "batch of 32 x 256x256x256 fp16 Perlin SDF grids (period 64, seeds 0..31), iso 0, one "
                  "marching_cubes_batched call per step",
            lambda: p3d.marching_cubes_batched(g5, 0.0)[:2], 32 * 256 ** 3, 2, steps=10)
    del g5
    g4 = perlin_grid((1024,) * 3, period=64, seed=0, device=dev)
    measure("c4_1gpu", "1024x1024x1024 fp32 single-octave Perlin SDF (period 64, seed 0), iso 0, on ONE GPU (the "
                       "baseline of the 8-GPU target)",
            lambda: p3d.libPrim3D.marching_cubes(g4, 0.0, [0.0] * 3, [1024.0] * 3), 1024 ** 3, 4, steps=6)
    del g4
    try:
        out["c4_rank_slab"] = rank_slab_workload(capi, perlin_grid, dev)
        # (the conservative figure: the receives cost a local copy each; and the step's own GPU work alone)
        out["c4_rank_slab"]["predicted_speedup_8gpu_no_transport"] = round(
            out["c4_1gpu"]["ms_per_step"] / out["c4_rank_slab"]["ms_per_step"], 2)
        out["c4_rank_slab"]["predicted_speedup_8gpu_without_standin_copies"] = round(
            out["c4_1gpu"]["ms_per_step"] / out["c4_rank_slab"]["ms_per_step_without_standin_copies"], 2)
    except Exception as e:   # (the headline must not depend on it)
        out["c4_rank_slab"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    # SURVEY.md 8d, C3's secondary workload: four octaves (period 64 -> 8, persistence 0.5), about 6.5 % active cells
    g6 = perlin_grid((512,) * 3, period=64, seed=0, octaves=4, persistence=0.5, device=dev)
    measure("c3_4oct", "512x512x512 fp32 four-octave Perlin SDF (periods 64..8, persistence 0.5, seed 0), iso 0",
            lambda: p3d.libPrim3D.marching_cubes(g6, 0.0, [0.0] * 3, [512.0] * 3), 512 ** 3, 4, steps=12)
    del g6
    # the reference's own input class at bench size: the examples/sphere.py:8-9 recipe (centre n/4, radius n/8; an object's
    # SDF in a box -- most of the grid is far from the surface) at n = 512
    ax = torch.arange(512, device=dev, dtype=torch.float32)
    g7 = ((ax[:, None, None] - 128) ** 2 + (ax[None, :, None] - 128) ** 2 + (ax[None, None, :] - 128) ** 2 - 64.0 ** 2).contiguous()
    measure("sphere512", "512x512x512 fp32 sphere field of examples/sphere.py:8-9 at n = 512 (centre 128, radius 64), iso 0",
            lambda: p3d.libPrim3D.marching_cubes(g7, 0.0, [0.0] * 3, [512.0] * 3), 512 ** 3, 4, steps=20)
    del g7
    torch.cuda.empty_cache()
    return out


def rank_slab_workload(capi, perlin_grid, dev, world=8, rank=3, shape=(1024, 1024, 1024), steps=12, warmup=4, hold=2):
    """What ONE rank of the 8-GPU run of BASELINE.json configs[3] does per step, on this GPU: rank 3's slab of the 1024^3
    volume -- 128 planes + the halo plane -- through SlabExtractor.extract()'s real sequence (interior planes streamed
    while the halo plane would travel: part 1; the last planes + header: part 3; export of the first plane's records for the
    previous rank; face count + early vertex copy: part 4; faces with the id bases taken ON THE DEVICE from the gathered
    counts + the rest of the copy: part 5).  The transport is stubbed: the halo plane and the imported records are local
    device copies of the right size, the all-gather a device copy of this rank's header words into every row -- so this is the
    per-rank cost WITHOUT xGMI latency, and c4_1gpu / this = the speed-up the 8-GPU run can reach at most.  Reported twice:
    `ms_per_step` with the two receives as local copies of the same size (what `predicted_speedup_8gpu_no_transport` uses: the
    conservative figure) and `ms_per_step_without_standin_copies` with nothing in their place (the step's own GPU work)."""
    import torch
    import torch.distributed as dist
    from primitive3d_amd.slab import SlabExtractor
    ex = SlabExtractor(shape, rank, world, dev, hold_planes=hold)
    ex.fill_local(lambda x0, x1: perlin_grid(shape, period=64, seed=0, device=dev, x0=x0, x1=x1))
    halo_src = perlin_grid(shape, period=64, seed=0, device=dev, x0=ex.x1, x1=ex.x1 + 1)[0].contiguous()
    rec_src = {}

    class _Op:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor = op, tensor

    standin = [True]   # the receives as local copies of the same size (False: nothing -- the halo plane stays as it was filled)

    def _batch(ops):   # a receive = a local copy of as many bytes; a send = nothing (the peer's receive is its copy)
        for o in ops:
            if o.op == "recv" and standin[0]:
                if o.tensor.shape == halo_src.shape and o.tensor.dtype == halo_src.dtype:
                    o.tensor.copy_(halo_src)
                else:
                    if o.tensor.numel() not in rec_src:   # (made once: a fill kernel per step would be the stub's cost, not the path's)
                        rec_src[o.tensor.numel()] = torch.zeros(o.tensor.numel(), dtype=o.tensor.dtype, device=dev)
                    o.tensor.copy_(rec_src[o.tensor.numel()])
        return []

    def _all_gather(out_t, inp, async_op=False):
        # (ONE device copy, like the collective it stands in for: every row receives this rank's header words -- the other
        #  ranks' counts would arrive here; similar slabs have similar counts, so the id bases are realistic)
        out_t.view(world, -1).copy_(inp.view(1, -1).expand(world, -1))
        return None   # (the asynchronous form's work object: nothing to wait for, the copy is in stream order)

    saved = {k: getattr(dist, k, None) for k in ("get_backend", "P2POp", "batch_isend_irecv", "all_gather_into_tensor", "isend", "irecv")}
    try:
        dist.get_backend = lambda *a, **k: "nccl"
        dist.P2POp, dist.batch_isend_irecv, dist.all_gather_into_tensor = _Op, _batch, _all_gather
        dist.isend, dist.irecv = "send", "recv"
        lower, upper = [0.0] * 3, [float(n) for n in shape]
        # (the other ranks' counts: zeros would do for timing; a made-up base keeps the id arithmetic realistic)
        for _ in range(warmup):
            res = ex.extract(0.0, lower, upper)
        res = None

        def group():   # K back-to-back extractions between two synchronisations
            nonlocal res
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                res = ex.extract(0.0, lower, upper)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3

        # (the median of three groups: one slow group -- a hiccup of the box -- once put 0.316 ms next to 0.242)
        ms = sorted(group() for _ in range(3))[1]
        # the same steps with NO stand-in for the two receives (the halo plane is in place from the steps above): the GPU work
        # of the step itself, without anything that takes the transport's place
        standin[0] = False
        for _ in range(2):
            res = ex.extract(0.0, lower, upper)
        ms_bare = sorted(group() for _ in range(3))[1]
        standin[0] = True
        ex.trace = True
        res = ex.extract(0.0, lower, upper)
        phases = {k: round(v, 4) for k, v in ex.phase_times_ms().items()}
    finally:
        for k, v in saved.items():
            setattr(dist, k, v)
    nvox = ex.n * shape[1] * shape[2]
    out = {"workload": f"rank {rank} of {world} of the 1024^3 run: a {ex.n}(+1 halo)x{shape[1]}x{shape[2]} fp32 slab through "
                       "SlabExtractor.extract() (parts 1/3/4/5, record export, device-side id bases), transport stubbed by "
                       "local copies of the same size",
           "steps": steps, "timing": "median of three groups of `steps` back-to-back extractions",
           "ms_per_step": round(ms, 4), "ms_per_step_without_standin_copies": round(ms_bare, 4),
           "value": round(nvox / (ms * 1e-3) / 1e6, 1), "unit": "Mvoxels/s",
           "dtype": "f32", "vertices": int(res.vertices.shape[0]), "faces": int(res.faces.shape[0]),
           "whole_call_frac": round(nvox * 4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "phases_ms_last_step": phases}
    del ex, res
    return out


def child_exact_mode(steps, warmup):
    """`--child exact` (run by measure_modes in a fresh process with P3D_MC_MODE=exact): the adapter in the reference's own
    call order -- count, host read of (V, F), exact allocation, emit (marching_cubes.cu:242-287).  Since round 4 the field
    is streamed ONCE for it when the guess for the internal vertex scratch holds (parts 3 / 4 / 6 of the C ABI); the
    two-pass route (count-only pass, then an exactly sized pass) remains behind it."""
    import torch
    import primitive3d_amd as p3d
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    dev = torch.device("cuda", 0)
    g = perlin_grid(SHAPES[1], period=64, seed=0, device=dev)
    up = [float(s) for s in SHAPES[1]]
    for _ in range(warmup):
        out = p3d.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, up)
    torch.cuda.synchronize()
    p0 = capi.debug_counters()["streaming_passes"]
    t0 = time.perf_counter()
    for _ in range(steps):
        out = p3d.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, up)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({"ms_per_step": round(ms, 4), "steps": steps,
                      "streaming_passes_per_call": (capi.debug_counters()["streaming_passes"] - p0) / steps,
                      "vertices": int(out[0].shape[0]), "faces": int(out[1].shape[0])}))


def child_fresh(steps, warmup):
    """`--child fresh` (run under rocprofv3 --kernel-trace --stats by tools/profile_round.sh): modes.fresh_grid's call stream
    alone -- four distinct 512^3 grids taken in turn -- so that the profiler's average for k_fused can be set beside the
    headline's (the headline re-extracts one resident grid)."""
    import torch
    import primitive3d_amd as p3d
    from primitive3d_amd.fields import perlin_grid
    dev = torch.device("cuda", 0)
    grids = [perlin_grid(SHAPES[1], period=64, seed=sd, device=dev) for sd in range(4)]
    up = [float(s) for s in SHAPES[1]]
    for i in range(warmup + steps):
        p3d.libPrim3D.marching_cubes(grids[i % 4], 0.0, [0.0] * 3, up)
    torch.cuda.synchronize()


def child_stream(steps, warmup):
    """`--child stream` (run under rocprofv3 --pmc by measure_traffic_live): the headline's call, a few times, nothing else."""
    import torch
    import primitive3d_amd as p3d
    from primitive3d_amd.fields import perlin_grid
    g = perlin_grid(SHAPES[1], period=64, seed=0, device=torch.device("cuda", 0))
    up = [float(s) for s in SHAPES[1]]
    for _ in range(warmup + steps):
        p3d.libPrim3D.marching_cubes(g, 0.0, [0.0] * 3, up)
    torch.cuda.synchronize()


def run_child(cmd, timeout, capture=False, **kw):
    """A child process in its OWN session: on a time-out the whole process group goes (a launcher such as rocprofv3 may
    have spawned the program instead of exec'ing it; killing only the launcher would leave it on the GPU while the
    sections that follow are being timed).  stdout is returned when asked for, everything else goes to /dev/null."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE if capture else subprocess.DEVNULL, stderr=subprocess.DEVNULL, text=True,
                         start_new_session=True, **kw)
    try:
        out, _ = p.communicate(timeout=timeout)
        return out
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        p.wait()
        raise


CALL_KERNELS = ("k_fused<", "k_face_count_walk<", "k_faces<")   # the three launches of a whole-grid call


def measure_traffic_live():
    """The bytes `k_fused` moves over the fabric per launch, measured in THIS run: two `rocprofv3 --pmc` passes (counters
    only, with --kernel-trace: the combination the GPU pool allows) over a child process that makes the headline's call
    seven times -- reads as L2 -> fabric read requests BY REQUEST SIZE (32 / 64 / 128 B: no correction factor; FETCH_SIZE
    tallies a 128-byte request at 64 bytes on gfx950, MI355X_MICROARCH.md "HBM"), writes as WRITE_SIZE (KiB, exact for
    streaming stores, same section).  Mean of the last three launches.  None when rocprofv3 is not there or a pass fails
    (the committed profiles/traffic.json is reported instead, labelled)."""
    import csv
    import glob
    import shutil
    import tempfile
    if not shutil.which("rocprofv3") or any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None   # (not there, or this process is itself being profiled)
    passes = {"rd": ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
              "wr": ["WRITE_SIZE"]}
    got = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            for tag, ctrs in passes.items():
                cmd = ["rocprofv3", "--pmc", *ctrs, "--kernel-trace", "--output-format", "csv", "-d", f"{td}/{tag}", "--",
                       sys.executable, str(ROOT / "bench.py"), "--child", "stream", "--steps", "5", "--warmup", "2"]
                run_child(cmd, timeout=120, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
                acc = {}
                for f in glob.glob(f"{td}/{tag}/**/*counter_collection.csv", recursive=True):
                    for r in csv.DictReader(open(f)):
                        for kern in CALL_KERNELS:
                            if kern in r["Kernel_Name"]:
                                acc.setdefault((kern, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
                for c, v in acc.items():
                    got[c] = sum(v[-3:]) / len(v[-3:])

        def read_of(kern):
            return (32 * got[(kern, "TCC_EA0_RDREQ_32B_sum")] + 64 * got[(kern, "TCC_EA0_RDREQ_64B_sum")]
                    + 128 * got[(kern, "TCC_EA0_RDREQ_128B_sum")])

        rd, wr = read_of("k_fused<"), 1024 * got[("k_fused<", "WRITE_SIZE")]
        if rd <= 0 or wr <= 0:
            return None
        out = {"read": int(rd), "write": int(wr)}
        try:   # the whole call: the streaming kernel, the triangle count, the faces (the last three calls of the child)
            out["call_read"] = int(sum(read_of(k) for k in CALL_KERNELS))
            out["call_write"] = int(sum(1024 * got[(k, "WRITE_SIZE")] for k in CALL_KERNELS))
        except KeyError:
            pass
        return out
    except Exception:
        return None


def measure_modes(p3d, capi, grid, lower, upper, perlin_grid):
    """What a call costs OUTSIDE the steady state the headline measures (same grid, same boundary function; a few steps
    each, after the headline's timed region):
      exact            P3D_MC_MODE=exact in a fresh child process: count -> read (V, F) on the host -> exact allocation -> emit,
                       the reference's own order (marching_cubes.cu:242-287): freshly allocated tensors of exactly V / F
                       rows, two host round trips inside the call, one pass over the field
      sparse_dense     the field alternating with an all-outside grid of the same shape (per-frame extraction of a changing
                       field): the adapter sizes its buffers for the largest of the last four calls, so every call is one pass
      hint_miss        one dense call after four sparse ones (the dense size has been forgotten): the output buffers are too
                       small, the vertex scratch (sized for at least a vertex per 16 voxels) is not -- faces and compaction run
                       a second time into larger buffers (`emissions_per_call` 2), the field is streamed once
      fresh_grid       FOUR distinct 512^3 Perlin grids (seeds 0..3: 2 GiB, eight times the 256 MB memory-side cache) taken in
                       turn through the same call: the headline re-extracts ONE resident grid, and part of that grid is
                       still in the memory-side cache when the next call starts (profiles/r04/dyn_ranges.txt section 13:
                       ~10 us of k_fused); a caller streaming new grids sees this number.  `k_fused_ms` by the same
                       dispatch-attached events as `roofline` -> `roofline.frac_fresh`
      two_phase        the literal pair of INTEGRATION.md section 2: p3d_mc_count -> p3d_mc_read_counts -> allocate ->
                       p3d_mc_emit (the reference's structure, marching_cubes.cu:242-287).  Since ABI v10 both are the
                       one-pass kernels: the streaming kernel in count-only form + the face count, then a second streaming
                       pass that stores every output region at its final rows + the face launch (no scratch, no copy)"""
    import torch
    out = {}
    try:
        env = dict(os.environ, P3D_MC_MODE="exact")
        r = run_child([sys.executable, str(ROOT / "bench.py"), "--child", "exact", "--steps", "8", "--warmup", "3"],
                      timeout=600, capture=True, env=env)
        out["exact"] = json.loads(r.strip().splitlines()[-1])
    except Exception as e:   # the headline must not depend on it
        out["exact"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    sparse = torch.ones_like(grid)

    extra = [0]   # emissions without a streaming pass of their own, in the last timed() call

    def timed(g):
        torch.cuda.synchronize()
        c0, t0 = capi.debug_counters(), time.perf_counter()
        p3d.libPrim3D.marching_cubes(g, 0.0, lower, upper)
        torch.cuda.synchronize()
        t1, c1 = time.perf_counter(), capi.debug_counters()
        extra[0] = c1["emissions_without_a_pass"] - c0["emissions_without_a_pass"]
        return (t1 - t0) * 1e3, c1["streaming_passes"] - c0["streaming_passes"]

    for _ in range(2):
        timed(sparse), timed(grid)
    dense_t, sparse_t, passes = [], [], 0
    for _ in range(6):
        t, n = timed(sparse)
        sparse_t.append(t)
        passes += n
        t, n = timed(grid)
        dense_t.append(t)
        passes += n
    out["sparse_dense"] = {"dense_call_ms": round(sorted(dense_t)[len(dense_t) // 2], 4),
                           "sparse_call_ms": round(sorted(sparse_t)[len(sparse_t) // 2], 4),
                           "streaming_passes_per_call": passes / 12, "timing": "synchronised single calls (median of 6)"}
    miss, emis = [], 0
    for _ in range(3):
        for _ in range(4):
            timed(sparse)
        miss.append(timed(grid))
        emis += 1 + extra[0]
    out["hint_miss"] = {"dense_call_ms": round(sorted(t for t, _ in miss)[1], 4),
                        "streaming_passes_per_call": sum(n for _, n in miss) / 3, "emissions_per_call": emis / 3,
                        "timing": "synchronised single calls (median of 3)"}
    timed(grid)   # (leave the hints as the headline left them)
    try:
        grids = [grid] + [perlin_grid(tuple(grid.shape), period=64, seed=sd, device=grid.device) for sd in (1, 2, 3)]
        for i in range(8):
            res = p3d.libPrim3D.marching_cubes(grids[i % 4], 0.0, lower, upper)
        torch.cuda.synchronize()
        p0, nst = capi.debug_counters()["streaming_passes"], 16
        t0 = time.perf_counter()
        for i in range(nst):
            res = p3d.libPrim3D.marching_cubes(grids[i % 4], 0.0, lower, upper)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / nst * 1e3
        passes = (capi.debug_counters()["streaming_passes"] - p0) / nst
        capi.profile_enable(1)
        dom = []
        for i in range(8):
            res = p3d.libPrim3D.marching_cubes(grids[i % 4], 0.0, lower, upper)
            dom.append(capi.profile_read().get("k_fused", float("nan")))
        torch.cuda.synchronize()
        capi.profile_enable(0)
        kms = sum(dom) / len(dom)
        alg = grid.numel() * grid.element_size()
        out["fresh_grid"] = {"grids": 4, "bytes_rotated": 4 * alg, "steps": nst, "ms_per_step": round(ms, 4),
                             "k_fused_ms": round(kms, 4), "k_fused_frac": round(alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "whole_call_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "streaming_passes_per_call": passes, "timing": "back-to-back calls between two synchronisations"}
        del grids, res
    except Exception as e:
        out["fresh_grid"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    try:
        for _ in range(3):
            v, f = capi.extract(grid, 0.0, lower, upper)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            v, f = capi.extract(grid, 0.0, lower, upper)
        torch.cuda.synchronize()
        out["two_phase"] = {"ms_per_step": round((time.perf_counter() - t0) / 8 * 1e3, 4), "steps": 8,
                            "vertices": int(v.shape[0]), "faces": int(f.shape[0]),
                            "binding": "p3d_mc_count -> p3d_mc_read_counts -> torch.empty x2 -> p3d_mc_emit through ctypes (capi.extract): "
                                       "two streaming passes, no scratch (ABI v10)",
                            "timing": "back-to-back calls between two synchronisations"}
        del v, f
    except Exception as e:
        out["two_phase"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    timed(grid)
    return out


def self_launch(gpus):
    """`python3 bench.py --gpus N` with no rank environment: this process -- which has made no GPU call and makes none --
    starts the N ranks the way the driver would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py <the same arguments>`) as a CHILD process, lets rank 0's JSON line and
    the ranks' stderr through (inherited descriptors) and exits with the launcher's return code.  Never os.exec*: a
    process that touched the GPU must not be replaced by another, and this one must stay around to report the code."""
    import socket
    import subprocess
    with socket.socket() as sk:   # (a free port of the loopback interface; the container's host name may not resolve)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), *sys.argv[1:]]
    sys.stderr.write("bench.py: no WORLD_SIZE in the environment -- starting the ranks: " + " ".join(cmd) + "\n")
    sys.stderr.flush()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if os.environ.get("P3D_BENCH_LAUNCH_DRY") == "1":   # (tests without a GPU: say what would be started, start nothing)
        return 0
    return subprocess.call(cmd, env=env, cwd=str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=0, help="override: cubic grid of this size (N=1 only)")
    ap.add_argument("--config", default="c3", choices=["c2", "c3", "c4", "c5"],
                    help="N=1 workload (BASELINE.json configs): c2 = 256^3 bunny SDF fp32, c3 = 512^3 Perlin fp32 (the "
                         "headline; default), c4 = the 1024^3 volume on one GPU, c5 = batch of 32 x 256^3 fp16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short c2 / c5 / c4-on-one-GPU measurements that follow the headline run")
    ap.add_argument("--stages", action="store_true", help="also print per-stage hipEvent times to stderr")
    ap.add_argument("--no-modes", action="store_true",
                    help="skip the short measurements of the call outside its steady state (`modes`) that follow the headline run")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure `roofline.traffic` in this run (two rocprofv3 --pmc child passes, ~20 s): report the "
                         "committed profiles/traffic.json instead")
    ap.add_argument("--child", default="", help=argparse.SUPPRESS)   # (internal: measure_modes' / measure_traffic_live's fresh process)
    args = ap.parse_args()
    if args.child == "exact":
        return child_exact_mode(args.steps, args.warmup)
    if args.child == "stream":
        return child_stream(args.steps, args.warmup)
    if args.child == "fresh":
        return child_fresh(args.steps, args.warmup)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus))   # (before anything touches the GPU: torch is not even imported yet)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: they must agree")
    # dev-only overrides to dry-run the N>1 code path on a 1-GPU box: all ranks on cuda:0, gloo transport
    share = os.environ.get("P3D_BENCH_SHARE_DEVICE") == "1"
    backend = os.environ.get("P3D_BENCH_BACKEND", "nccl")
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.load_build_module().build_all()  # no-op when the in-tree libraries are up to date
    if world > 1:
        dist.barrier()
    import primitive3d_amd as p3d
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid

    batch = 1
    if world == 1:
        shape = (args.size,) * 3 if args.size else {"c2": (256,) * 3, "c3": SHAPES[1], "c4": (1024,) * 3,
                                                     "c5": (256,) * 3}[args.config]
        if args.config == "c5":
            batch = 32 if not args.size else 4
    else:
        if world not in SHAPES:
            sys.exit(f"unsupported world size {world}")
        shape = SHAPES[world]
        if args.size:  # dev: smaller dry-run volume, cubic per rank
            shape = (args.size * world, args.size, args.size)
    rx, ry, rz = shape
    thresh = 0.0
    lower, upper = [0.0, 0.0, 0.0], [float(rx), float(ry), float(rz)]

    workload = f"{rx}x{ry}x{rz} fp32 single-octave Perlin SDF (period 64, seed 0), iso 0"
    sizeof = 4
    if world == 1 and args.config == "c5":
        # BASELINE.json configs[4]: per-frame density grids, fp16 in memory (compared in fp32: the up-cast is exact)
        grid = torch.stack([perlin_grid(shape, period=64, seed=s, device=dev).half() for s in range(batch)])
        sizeof = 2
        workload = (f"batch of {batch} x {rx}x{ry}x{rz} fp16 Perlin SDF grids (period 64, seeds 0..{batch - 1}), iso 0, "
                    f"one marching_cubes_batched call per step")

        def step():
            return p3d.marching_cubes_batched(grid, thresh)[:2]
    elif world == 1 and args.config == "c2" and not args.size:
        # BASELINE.json configs[1] as SURVEY.md 8d defines it: the reference's 66^3 bunny SDF resampled to 256^3
        import numpy as np
        b66 = torch.from_numpy(np.load(ROOT / "tests" / "golden" / "bunny66.npy"))
        grid = torch.nn.functional.interpolate(b66[None, None], size=shape, mode="trilinear", align_corners=True)[0, 0]
        grid = grid.contiguous().to(dev)
        workload = "256x256x256 fp32 bunny SDF (examples/data/bunny.npy trilinearly resampled from 66^3), iso 0"

        def step():
            return p3d.libPrim3D.marching_cubes(grid, thresh, lower, upper)
    elif world == 1:
        grid = perlin_grid(shape, period=64, seed=0, device=dev)

        def step():
            return p3d.libPrim3D.marching_cubes(grid, thresh, lower, upper)
    else:
        from primitive3d_amd.slab import SlabExtractor
        ex = SlabExtractor(shape, rank, world, dev)
        ex.trace = args.stages
        ex.fill_local(lambda x0, x1: perlin_grid(shape, period=64, seed=0, device=dev, x0=x0, x1=x1))

        def step():
            return ex.extract(thresh, lower, upper)

    def barrier():
        if world > 1:
            dist.barrier()

    # cold call: the first call on this shape has no size hint (density guess, maybe a second streaming pass) and pays
    # the library's one-time setup; reported once, never part of `value`
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    out = step()
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - c0) * 1e3
    for _ in range(max(0, args.warmup - 1)):
        out = step()
    torch.cuda.synchronize()

    # ---- the timed region: exactly K steps, nothing else on the stream ----
    barrier()
    torch.cuda.synchronize()
    lay0 = capi.debug_counters()["layout_passes"]   # (a host-side counter of the library: no GPU work)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    laid_out = capi.debug_counters()["layout_passes"] - lay0

    # ---- the dominant kernel's duration, live, by hipEvents that ride on its own dispatch packet, on the steps that
    # follow immediately (same inputs, same stream, same back-to-back call pattern).  Not inside the timed region: an
    # event-carrying dispatch costs the stream ~14 us per call (kernel trace: 7 us of idle before the next call and
    # slower neighbours; tools/dev/plain_loop.py), and switching the events on and off between steps costs more still
    # (the runtime re-configures queue profiling), so either way `value` would be taxed by ~5 % by its own measurement.
    # The kernel's own duration is the same with and without the events (and agrees with rocprofv3, profiles/).
    n_inst = max(3, min(args.steps, 10))
    capi.profile_enable(2 if args.stages else 1)
    dom_ms = []
    stage_acc = {}
    for _ in range(n_inst):
        out = step()
        st = capi.profile_read()  # events of kernels that already finished (the host read V,F after them)
        dom_ms.append(st.get("k_fused", st.get("k_classify", float("nan"))) + st.get("k_fused(interior part)", 0.0))
        for k, v in st.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    torch.cuda.synchronize()
    capi.profile_enable(0)
    barrier()

    # SURVEY.md 8d also asks for the per-call median: >= 10 calls, each bracketed by events on the call's stream
    # (after the timed region: event packets between the calls perturb the back-to-back stream slightly)
    call_ms = []
    if world == 1:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
        for e0, e1 in evs:
            e0.record()
            out = step()
            e1.record()
        torch.cuda.synchronize()
        call_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)

    # N > 1: what RCCL saw (ranks, their devices), and -- on rank 0, after and outside the timed region, while the other
    # ranks wait at the barrier -- the SAME whole volume through the plain single-GPU call: the denominator of the
    # north-star's ">= 6x at 8 GPUs on 1024^3" measured in the same run, on the same box
    rccl_seen = full_1gpu = None
    if world > 1:
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(),
                                      "name": torch.cuda.get_device_name(dev)})
        rccl_seen = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                     "devices": [d["device"] for d in sorted(seen, key=lambda d: d["rank"])],
                     "device_names": sorted(set(d["name"] for d in seen))}
        if rank == 0:
            try:
                gfull = perlin_grid(shape, period=64, seed=0, device=dev)
                n1 = 4
                for _ in range(2):
                    fv, ff = p3d.libPrim3D.marching_cubes(gfull, thresh, lower, upper)
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                for _ in range(n1):
                    fv, ff = p3d.libPrim3D.marching_cubes(gfull, thresh, lower, upper)
                torch.cuda.synchronize()
                full_1gpu = {"ms_per_step": (time.perf_counter() - c0) / n1 * 1e3, "steps": n1,
                             "vertices": int(fv.shape[0]), "faces": int(ff.shape[0])}
                del gfull, fv, ff
                torch.cuda.empty_cache()
            except Exception as e:   # (the line must not depend on it: e.g. no room for the whole volume)
                full_1gpu = {"error": f"{type(e).__name__}: {e}"[:300]}
        barrier()

    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    nvox_total = rx * ry * rz * batch
    ms_per_step = elapsed / args.steps * 1e3
    value = nvox_total * args.steps / elapsed / 1e6

    out_v, out_f = out
    nv, nf = int(out_v.shape[0]), int(out_f.shape[0])
    if world > 1:
        cnt = torch.tensor([nv, nf], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt)
        nv, nf = int(cnt[0]), int(cnt[1])

    if rank == 0:
        # roofline of the dominant kernel: algorithmic bytes = the field read once
        local_vox = nvox_total // world
        alg_bytes = local_vox * sizeof
        avg_ms = sum(dom_ms) / len(dom_ms)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = traffic_build = None
        tfile = ROOT / "profiles" / "traffic.json"
        headline = world == 1 and shape == SHAPES[1] and batch == 1 and args.config == "c3"   # this workload only
        live = None   # (measured LAST, behind every timed section: it starts child processes -- see below)
        if tfile.exists() and headline:
            try:
                tj = json.loads(tfile.read_text())
                traffic = tj.get("k_fused_hbm_bytes_per_launch")
                traffic_build = tj.get("build")   # the commit / library stamp the PMC passes were taken on
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "k_fused", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "avg_kernel_ms": round(avg_ms, 4), "launches_timed": len(dom_ms), "timed_on": "the steps right after the timed region (dispatch-attached hipEvents perturb the call stream by ~14 us per call)", "alg_bytes_per_launch": alg_bytes,
                    "whole_call_frac": round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        def note_traffic():   # the same launch time against the bytes the kernel really moves (halo planes and rows, outputs)
            if not roofline["traffic"]:
                return
            roofline["traffic_frac"] = round(roofline["traffic"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if live:
                roofline["traffic_read"], roofline["traffic_write"] = live["read"], live["write"]
                if "call_read" in live:
                    # the WHOLE call over the fabric (all three kernels), and the rate the call reaches on those bytes
                    roofline["call_traffic_read"], roofline["call_traffic_write"] = live["call_read"], live["call_write"]
                    roofline["call_traffic"] = live["call_read"] + live["call_write"]
                    roofline["call_traffic_frac"] = round(roofline["call_traffic"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                roofline["traffic_source"] = ("this run: rocprofv3 --pmc over a child making the same call (fabric read requests by "
                                              "request size; WRITE_SIZE), k_fused, mean of the last 3 of 7 launches")
                roofline.pop("traffic_build", None)
            else:
                roofline["traffic_source"] = "profiles/traffic.json (a committed PMC result of the build named in traffic_build, not of this run)"
                roofline["traffic_build"] = traffic_build
        if call_ms:
            roofline["call_median_ms_hipevents"] = round(call_ms[len(call_ms) // 2], 4)
        roofline["cold_first_call_ms"] = round(cold_ms, 3)
        if args.stages:
            print("stage ms/step:", {k: round(v / len(dom_ms), 4) for k, v in stage_acc.items()}, file=sys.stderr)

        line = {
            "metric": "Mvoxels/s on 512^3 fp32 SDF (whole marching_cubes call, device-resident grid)"
                      if (rx, ry, rz, batch) == (512, 512, 512, 1) else
                      f"Mvoxels/s on {'%d x ' % batch if batch > 1 else ''}{rx}x{ry}x{rz} "
                      f"{'fp16' if sizeof == 2 else 'fp32'} SDF (whole marching_cubes call, device-resident grid)",
            "value": round(value, 1), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16" if sizeof == 2 else "f32", "data": "synthetic",
            "config": {"workload": workload,
                       # which allocation policy of the pybind adapter the headline ran: "hinted"
                       # (default) or "exact" (P3D_MC_MODE=exact; under `modes` when it is not the headline's)
                       "adapter_mode": (os.environ.get("P3D_MC_MODE") or "hinted") if world == 1 and batch == 1 else None,
                       # how many of the K timed calls stored their vertices where they stay (the predicted region layout,
                       # no scratch tensor, no copy of the vertex rows) -- K in the steady state of `hinted`
                       "calls_with_region_layout": laid_out,
                       "voxels_per_gpu": local_vox, "vertices": nv, "faces": nf,
                       "partition": "none" if world == 1 else f"axis-0 slabs x{world}, 1-plane RCCL halo"},
            "roofline": roofline,
        }
        if world > 1:
            # "RCCL saw N ranks": the process group's own view (backend, world size, the device of every rank)
            line["config"]["rccl"] = rccl_seen
            line["config"]["parallelism"] = f"slab{world}"
            if full_1gpu and "error" not in full_1gpu:
                one = full_1gpu["ms_per_step"]
                line["full_volume_1gpu"] = {**full_1gpu, "ms_per_step": round(one, 4),
                                            "workload": f"the same {rx}x{ry}x{rz} volume through the plain single-GPU call on "
                                                        "rank 0's GPU, same run, after the timed region (the other ranks wait)",
                                            "meshes_agree": (full_1gpu["vertices"], full_1gpu["faces"]) == (nv, nf)}
                line["speedup_vs_1gpu"] = float(f"{one / ms_per_step:.4g}")
                if (rx, ry, rz) == (1024, 1024, 1024):   # BASELINE.json configs[3]: the >= 6x target is this one field
                    line["c4_1gpu_ms"] = round(one, 4)
                    line["speedup_vs_1gpu_1024"] = float(f"{one / ms_per_step:.4g}")
            elif full_1gpu:
                line["full_volume_1gpu"] = full_1gpu
        if world == 1 and args.config == "c3" and not args.size and not args.no_modes:
            line["modes"] = measure_modes(p3d, capi, grid, lower, upper, perlin_grid)
            fg = line["modes"].get("fresh_grid", {})
            if "k_fused_frac" in fg:
                roofline["frac_fresh"] = fg["k_fused_frac"]
                roofline["frac_note"] = ("frac: the headline re-extracts ONE resident grid (part of it is still in the 256 MB "
                                         "memory-side cache when the next call starts); frac_fresh: four distinct grids in turn "
                                         "(modes.fresh_grid)")
        if world == 1 and args.config == "c3" and not args.size and not args.no_other_configs:
            line["other_configs"] = other_configs(p3d, capi, perlin_grid, dev)
        if world == 1 and not args.no_cpu_baseline:
            if batch > 1:   # bounded sample: the first items of the batch, each compared with the GPU's mesh of that item
                v_all, f_all, vo, fo = p3d.marching_cubes_batched(grid, thresh)
                torch.cuda.synchronize()
                vo, fo = vo.cpu(), fo.cpu()
                nsample = min(batch, 4)
                res = None
                for b in range(nsample):
                    r = cpu_baseline(grid[b].float(), thresh, lower, upper, v_all[vo[b]:vo[b + 1]], f_all[fo[b]:fo[b + 1]],
                                     budget=(2.0, 1.5))
                    res = r if res is None else res
                res["sample"] = (f"items 0..{nsample - 1} of the batch, each extracted repeatedly by oracle/mc_oracle.c "
                                 f"(first item's rates reported) and compared with the GPU's mesh of that item; "
                                 + res["sample"])
                line["cpu_baseline"] = res
            else:
                line["cpu_baseline"] = cpu_baseline(grid, thresh, lower, upper, out_v, out_f)
        # the bytes k_fused really moves, measured in this run -- LAST: two rocprofv3 child processes, after every timed section
        if headline and not args.no_live_traffic:
            live = measure_traffic_live()
            if live:
                roofline["traffic"] = live["read"] + live["write"]
        note_traffic()
        print(json.dumps(line))
    if world > 1 and args.stages:
        # per-phase GPU time of the LAST step on every rank (halo wait, all-gather, record exchange, ...): what the first
        # real multi-GPU run needs to be diagnosable
        # (ONE write per rank: the ranks share stderr, and a line printed in pieces interleaves with the others')
        sys.stderr.write(f"rank {rank} phases (ms): " + str({k: round(v, 4) for k, v in ex.phase_times_ms().items()}) + "\n")
        sys.stderr.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
