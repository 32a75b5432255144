"""Dry run of bench.py's N > 1 branch (SlabExtractor over torch.distributed, max-over-ranks timing, gathered counts, the
per-rank phase lines of --stages) the way the driver launches it -- `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...` -- but on ONE GPU: all ranks share cuda:0 (P3D_BENCH_SHARE_DEVICE=1) and talk over gloo
(P3D_BENCH_BACKEND=gloo; RCCL refuses two ranks on one device).  The first real `--gpus 8` run is the driver's; it must
not also be the first run of this code path.  BASELINE.json configs[3]: the volume as axis-0 slabs, one halo
plane per boundary."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,size,launcher", [(2, 128, "driver"), (8, 64, "driver"), (2, 96, "self")])
def test_bench_gpus_n_dry_run(gpu, built, world, size, launcher):
    """launcher "driver": the driver's command line (torch.distributed.run around bench.py); "self": plain
    `python bench.py --gpus N` with no rank environment -- bench.py starts the ranks itself as a child process and relays
    rank 0's line and the return code (the first 8-GPU run must not die at argument parsing)."""
    from primitive3d_amd import capi
    from primitive3d_amd.fields import perlin_grid
    env = dict(os.environ, P3D_BENCH_SHARE_DEVICE="1", P3D_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    tail = [str(ROOT / "bench.py"), "--gpus", str(world), "--size", str(size), "--steps", "3", "--warmup", "2", "--stages"]
    if launcher == "driver":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), *tail]
    else:
        cmd = [sys.executable, *tail]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]   # rank 0 alone prints the line
    d = json.loads(lines[0])
    shape = (size * world, size, size)
    assert d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["partition"] == f"axis-0 slabs x{world}, 1-plane RCCL halo"
    assert d["config"]["voxels_per_gpu"] == size ** 3 and f"{shape[0]}x{shape[1]}x{shape[2]}" in d["config"]["workload"]
    assert abs(d["value"] - shape[0] * shape[1] * shape[2] / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.01
    assert "cpu_baseline" not in d and "other_configs" not in d and "modes" not in d
    # what the process group saw, and the same whole volume on ONE GPU measured by rank 0 in the same run (item 2b, 2d)
    rc = d["config"]["rccl"]
    assert rc["world_size"] == world and rc["backend"] == "gloo" and rc["devices"] == [0] * world, rc
    one = d["full_volume_1gpu"]
    assert "error" not in one and one["meshes_agree"] is True and one["ms_per_step"] > 0, one
    assert d["speedup_vs_1gpu"] == pytest.approx(one["ms_per_step"] / d["ms_per_step"], rel=0.01)
    assert d["roofline"]["kernel"] == "k_fused" and d["roofline"]["alg_bytes_per_launch"] == size ** 3 * 4
    # the slabs' meshes add up to the plain call's on the whole field
    g = perlin_grid(shape, period=64, seed=0, device=gpu)
    v, f = capi.extract_fused(g, 0.0, [0.0] * 3, [float(s) for s in shape])
    torch.cuda.synchronize()
    assert (d["config"]["vertices"], d["config"]["faces"]) == (v.shape[0], f.shape[0])
    # --stages: every rank reports the GPU time between the phase marks of its last extraction
    import re
    assert sorted(int(r) for r in re.findall(r"rank (\d+) phases \(ms\): \{", out.stderr)) == list(range(world)), out.stderr[-2000:]
    assert any("stage ms/step" in ln for ln in out.stderr.splitlines())
