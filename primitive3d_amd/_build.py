"""In-tree builds of the native artefacts (all land next to this file so they travel with the
repo snapshot):

  libp3dmc.so        HIP kernels + C ABI (include/p3d_mc.h), built with hipcc for gfx950
  dev/libp3dmc.so    the same sources with -DP3D_DEV_HOOKS=1: the developer sweeps' launch knobs and the test hooks
                     (P3D_TEST_*, P3D_NO_CHUNK_PRE, ...) are compiled in; only tests that need a hook load it
                     (tests/test_gpu_dev_hooks.py, in a child process with LD_LIBRARY_PATH + P3D_CAPI_LIB)
  libPrim3D*.so      pybind adapter (csrc/bindings.cpp) over that C ABI, built with the host compiler
                     against libtorch; module name kept from the reference (src/pybind/CMakeLists.txt:13-14)
"""
import os
import subprocess
import sys
import sysconfig
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
ROOT = PKG.parent
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def _digest(deps, extra="") -> str:
    import hashlib
    h = hashlib.sha256(extra.encode())
    for d in deps:
        h.update(Path(d).name.encode())
        h.update(Path(d).read_bytes())
    return h.hexdigest()


def _stale(out: Path, deps, extra="") -> bool:
    """An artefact is current when the stamp next to it holds the digest of its sources (mtimes mean nothing after a
    checkout; the stamp travels with the .so in the gpurun snapshot)."""
    stamp = out.with_name(out.name + ".stamp")
    if not out.exists() or not stamp.exists():
        return True
    return stamp.read_text().strip() != _digest(deps, extra)


def _write_stamp(out: Path, deps, extra=""):
    out.with_name(out.name + ".stamp").write_text(_digest(deps, extra) + "\n")


# k_fused's plane claims are `if (lane == 0) old = atomic add`, consumed one plane later: LLVM's atomic optimizer would
# rewrite that into a wave reduction that reads the result (and waits for it) right behind the atomic
CAPI_EXTRA_FLAGS = ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def capi_path() -> Path:
    override = os.environ.get("P3D_CAPI_LIB")  # dev: try an alternative build of the C-ABI library
    return Path(override) if override else PKG / "libp3dmc.so"


def capi_dev_path() -> Path:
    """The -DP3D_DEV_HOOKS=1 variant.  Same file name in its own directory, so that a child process started with
    LD_LIBRARY_PATH=<that directory> has the pybind module (which links `libp3dmc.so` by name) load it too."""
    return PKG / "dev" / "libp3dmc.so"


def dev_env(base=None) -> dict:
    """Environment of a child process that runs on the dev-hooks variant (pybind module and ctypes alike)."""
    env = dict(os.environ if base is None else base)
    d = str(capi_dev_path().parent)
    env["LD_LIBRARY_PATH"] = d + (":" + env["LD_LIBRARY_PATH"] if env.get("LD_LIBRARY_PATH") else "")
    env["P3D_CAPI_LIB"] = str(capi_dev_path())
    return env


def pybind_path() -> Path:
    return PKG / ("libPrim3D" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_capi(force: bool = False, verbose: bool = False, dev: bool = False) -> Path:
    import time
    if not dev and os.environ.get("P3D_CAPI_LIB"):
        return capi_path()   # someone else's build, bound for an A/B run: never rebuilt over (it has no stamp of ours)
    out = capi_dev_path() if dev else capi_path()
    flags = CAPI_EXTRA_FLAGS + (["-DP3D_DEV_HOOKS=1"] if dev else [])
    deps = [CSRC / "p3d_mc.hip", *sorted(CSRC.glob("*.inc")), *sorted(CSRC.glob("*.h")), ROOT / "include" / "p3d_mc.h"]
    if force or _stale(out, deps, " ".join(flags)):
        out.parent.mkdir(exist_ok=True)
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
               # the reference epilogue is mul-then-add, never an FMA (marching_cubes.cu:298)
               "-ffp-contract=off", "-Wall", "-Wextra", *flags,
               str(CSRC / "p3d_mc.hip"), "-o", str(out)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        t0 = time.time()
        subprocess.check_call(cmd)
        dt = time.time() - t0
        out.with_name(out.name + ".buildtime").write_text(f"{dt:.1f} s  ({' '.join(cmd[:2])} ... p3d_mc.hip)\n")
        if verbose:
            print(f"compiled p3d_mc.hip in {dt:.1f} s", file=sys.stderr)
        _write_stamp(out, deps, " ".join(flags))
    return out


def build_pybind(force: bool = False, verbose: bool = False) -> Path:
    out = pybind_path()
    deps = [CSRC / "bindings.cpp", ROOT / "include" / "p3d_mc.h", ROOT / "include" / "p3d_rc.h"]
    if force or _stale(out, deps):
        import torch
        from torch.utils import cpp_extension as ce
        inc = []
        for p in ce.include_paths():
            inc += ["-isystem", p]
        inc += ["-isystem", sysconfig.get_paths()["include"], "-isystem", "/opt/rocm/include"]
        tlib = Path(torch.__file__).parent / "lib"
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared",
               "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=libPrim3D",
               "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
               *inc, str(CSRC / "bindings.cpp"), "-o", str(out),
               f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", "-ltorch_hip", "-lc10_hip",
               f"-L{PKG}", "-lp3dmc", "-lp3drc", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}", "-Wno-deprecated-declarations"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        _write_stamp(out, deps)
    return out


def mt_path() -> Path:
    return PKG / "libp3dmt.so"


def build_mt(force: bool = False, verbose: bool = False) -> Path:
    """libp3dmt.so: marching tetrahedra (include/p3d_mt.h), HIP kernels around rocPRIM's sort and scans."""
    out = mt_path()
    deps = [CSRC / "p3d_mt.hip", ROOT / "include" / "p3d_mt.h"]
    if force or _stale(out, deps):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
               # the reference's interpolation is two separately rounded products and a sum (marching_tetrahedras.py:188)
               "-ffp-contract=off", "-Wall", "-Wextra", "-Wno-unused-parameter",
               str(CSRC / "p3d_mt.hip"), "-o", str(out)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        _write_stamp(out, deps)
    return out


def rc_path() -> Path:
    return PKG / "libp3drc.so"


def build_rc(force: bool = False, verbose: bool = False) -> Path:
    """libp3drc.so: ray caster (include/p3d_rc.h): BVH4 build on the host, HIP traversal kernel."""
    out = rc_path()
    deps = [CSRC / "p3d_rc.hip", ROOT / "include" / "p3d_rc.h"]
    if force or _stale(out, deps):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
               "-Wextra", str(CSRC / "p3d_rc.hip"), "-o", str(out)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        _write_stamp(out, deps)
    return out


def build_all(force: bool = False, verbose: bool = False):
    a = build_capi(force, verbose)
    build_capi(force, verbose, dev=True)
    d = build_rc(force, verbose)     # (the pybind adapter links it)
    b = build_pybind(force, verbose)
    c = build_mt(force, verbose)
    return a, b, c, d


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
