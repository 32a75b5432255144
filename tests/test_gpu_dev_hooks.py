"""Knobs and hooks.  The default library (primitive3d_amd/libp3dmc.so) reads six documented knobs from the environment
and nothing else; the developer sweeps' launch knobs and the test hooks (P3D_TEST_ID_LIMIT, P3D_TEST_INDEX_LIMIT,
P3D_NO_CHUNK_PRE, P3D_TEST_FAIL_AFTER_LEASE, ...) exist only in the -DP3D_DEV_HOOKS=1 variant (primitive3d_amd/dev/).
The tests marked `dev_hooks` all over tests/ run HERE, in one child pytest process whose pybind module and ctypes binding
both load that variant."""
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
SUPPORTED = {"P3D_FUSED_BLOCKS", "P3D_FUSED_XT", "P3D_COMPACT_BLOCKS", "P3D_COMPACT_EARLY", "P3D_FACES_SPARSE",
             "P3D_NO_MAILBOX"}


def _env_names(path):
    return set(re.findall(rb"P3D_[A-Z0-9_]+", Path(path).read_bytes()))


def test_the_default_library_holds_only_the_supported_knobs(built):
    """strings(libp3dmc.so): every P3D_* name in the default library is a documented knob -- at most 8 -- and
    every one of them is listed in include/p3d_mc.h and INTEGRATION.md; the dev variant holds the hooks."""
    from primitive3d_amd import _build
    names = {n.decode() for n in _env_names(_build.capi_path())}
    assert names == SUPPORTED, names ^ SUPPORTED
    assert len(names) <= 8
    header = (ROOT / "include" / "p3d_mc.h").read_text()
    integ = (ROOT / "INTEGRATION.md").read_text()
    for n in sorted(names | {"P3D_MC_MODE"}):
        assert n in header, f"{n} is not documented in include/p3d_mc.h"
        assert n in integ, f"{n} is not documented in INTEGRATION.md"
    dev = {n.decode() for n in _env_names(_build.capi_dev_path())}
    assert {"P3D_TEST_ID_LIMIT", "P3D_TEST_INDEX_LIMIT", "P3D_NO_CHUNK_PRE", "P3D_TEST_FAIL_AFTER_LEASE"} <= dev
    # the pybind adapter reads exactly one variable
    assert {n.decode() for n in _env_names(_build.pybind_path()) if n.startswith(b"P3D_MC_") and b"ABI" not in n} == {"P3D_MC_MODE"}


@pytest.mark.gpu
def test_the_default_library_ignores_test_hooks(gpu, built):
    """P3D_TEST_ID_LIMIT=16 in the environment of a process that loads the DEFAULT library: no id-overflow flag, one streaming
    pass, the oracle's counts -- where the dev variant reports the overflow (test_region_id_space_overflow_... in the child run
    below)."""
    code = (
        "import numpy as np, torch, json\n"
        "from primitive3d_amd import capi\n"
        "from primitive3d_amd.fields import perlin_grid\n"
        "assert capi.lib().p3d_mc_dev_hooks() == 0\n"
        "g = perlin_grid(64, period=16, seed=3).cuda()\n"
        "ws = torch.empty(capi.workspace_bytes(64, 64, 64), dtype=torch.uint8, device='cuda')\n"
        "v = torch.empty((1 << 18, 3), dtype=torch.float32, device='cuda')\n"
        "f = torch.empty((1 << 19, 3), dtype=torch.int32, device='cuda')\n"
        "before = capi.debug_counters()['streaming_passes']\n"
        "capi.extract_fused_raw(g, 0.0, [0, 0, 0], [64, 64, 64], ws, v, f)\n"
        "nv, nf, flags = capi.read_counts(ws, with_flags=True)\n"
        "print(json.dumps({'nv': nv, 'nf': nf, 'flags': int(flags), 'passes': capi.debug_counters()['streaming_passes'] - before}))\n")
    env = {**os.environ, "P3D_TEST_ID_LIMIT": "16", "P3D_TEST_INDEX_LIMIT": "5", "P3D_NO_CHUNK_PRE": "1"}
    env.pop("P3D_CAPI_LIB", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    r = json.loads(out.stdout.strip().splitlines()[-1])
    from oracle import oracle_count
    from primitive3d_amd.fields import perlin_grid
    assert (r["nv"], r["nf"]) == oracle_count(perlin_grid(64, period=16, seed=3).numpy(), 0.0)
    assert r["flags"] == 0 and r["passes"] == 1


@pytest.mark.gpu
def test_hook_cases_on_the_dev_variant(gpu, built):
    """Every test marked dev_hooks, in a child pytest on primitive3d_amd/dev/libp3dmc.so (LD_LIBRARY_PATH makes the pybind
    module load it, P3D_CAPI_LIB the ctypes binding: one library, one set of globals)."""
    from primitive3d_amd import _build
    env = _build.dev_env()
    env["P3D_DEV_VARIANT"] = "1"
    out = subprocess.run([sys.executable, "-m", "pytest", "tests", "-q", "-x", "-m", "gpu and dev_hooks", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, cwd=ROOT, timeout=3000)
    tail = out.stdout[-3000:] + out.stderr[-2000:]
    assert out.returncode == 0, tail
    m = re.search(r"(\d+) passed", out.stdout)
    assert m and int(m.group(1)) >= 6, tail   # (nothing silently deselected)
    assert "skipped" not in out.stdout.splitlines()[-1], tail


@pytest.mark.gpu
@pytest.mark.dev_hooks
def test_the_child_really_runs_on_the_dev_variant(gpu, built):
    from primitive3d_amd import capi
    assert capi.lib().p3d_mc_dev_hooks() == 1
    import ctypes
    # the pybind module resolved the same library (one set of globals): its calls count in the same counters
    before = capi.debug_counters()["streaming_passes"]
    built.marching_cubes(torch.zeros(8, 8, 8, device=gpu), 0.5)
    assert capi.debug_counters()["streaming_passes"] == before + 1
