import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "dev_hooks: needs the -DP3D_DEV_HOOKS=1 variant of the C-ABI library (a test hook such as "
                            "P3D_TEST_ID_LIMIT): runs only in the child process tests/test_gpu_dev_hooks.py starts on that variant")
    # test modules import the package at collection time and the package refuses to import without its native
    # artefacts: (re)build them first -- a no-op when they are up to date (they travel with the repo snapshot)
    import __graft_entry__
    try:
        __graft_entry__.load_build_module().build_all()
    except FileNotFoundError as e:
        # no hipcc on this machine: the oracle / gloo tests still run; everything that needs the libraries fails at
        # import or in the `built` fixture (there is no fallback to hide behind)
        print(f"conftest: native build skipped ({e})", file=sys.stderr)


def pytest_collection_modifyitems(config, items):
    """Tests that need a test hook run on the dev variant of the library only (P3D_DEV_VARIANT=1: the child process of
    tests/test_gpu_dev_hooks.py); everywhere else they are deselected -- the default library has no hooks to drive."""
    if os.environ.get("P3D_DEV_VARIANT") == "1":
        return
    hooked = [it for it in items if it.get_closest_marker("dev_hooks")]
    if hooked:
        items[:] = [it for it in items if not it.get_closest_marker("dev_hooks")]
        config.hook.pytest_deselected(items=hooked)


SUPPORTED_KNOBS = {"P3D_FUSED_BLOCKS", "P3D_FUSED_XT", "P3D_COMPACT_BLOCKS", "P3D_COMPACT_EARLY", "P3D_FACES_SPARSE"}


@pytest.fixture(scope="session")
def built():
    """Build (if stale) and import the native artefacts; every test that touches them depends on this."""
    import __graft_entry__
    __graft_entry__.load_build_module().build_all()
    import primitive3d_amd
    return primitive3d_amd


@pytest.fixture(scope="session")
def gpu(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible (torch.cuda.is_available() is False)")
    return torch.device("cuda:0")


@pytest.fixture
def tuning_env(built):
    """Set P3D_* knobs of the native library for one test.  The library reads them once per process
    (include/p3d_mc.h, p3d_mc_reload_tuning): set(name, value) changes the environment and makes the library re-read
    it; the old environment is restored, and re-read, when the test ends."""
    from primitive3d_amd import capi
    saved = {}

    def set_(name, value):
        if name not in SUPPORTED_KNOBS and not capi.lib().p3d_mc_dev_hooks():
            pytest.fail(f"{name} is a test hook: this test must be marked dev_hooks (it runs on dev/libp3dmc.so)")
        saved.setdefault(name, os.environ.get(name))
        if value is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = str(value)
        capi.reload_tuning()

    yield set_
    for name, old in saved.items():
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old
    capi.reload_tuning()
