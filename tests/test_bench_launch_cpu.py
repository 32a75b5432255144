"""bench.py --gpus N without a rank environment starts its ranks itself: the parent makes no GPU
call (torch is not imported on that path), spawns `python -m torch.distributed.run ... bench.py <same arguments>` as a
CHILD process on 127.0.0.1 and exits with its return code.  No GPU needed: the launcher is replaced by a recorder."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_self_launch_builds_the_drivers_command(monkeypatch):
    sys.path.insert(0, str(ROOT))
    import bench
    seen = {}

    def fake_call(cmd, env=None, cwd=None):
        seen.update(cmd=cmd, env=env, cwd=cwd)
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    assert bench.self_launch(8) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(str(ROOT / "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_gpus_n_without_rank_environment_takes_the_launcher_and_relays_its_code():
    """The whole program, in a child: `python bench.py --gpus 2` with WORLD_SIZE / RANK unset must reach the launcher
    (before importing torch: no GPU in this container) and exit with the launcher's code -- here the ranks fail (no GPU),
    so the code is non-zero, and the parent says what it started."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["P3D_BENCH_LAUNCH_DRY"] = "1"
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert "starting the ranks" in out.stderr and "torch.distributed.run" in out.stderr
    assert out.returncode == 0 and "--nproc-per-node=2" in out.stderr
