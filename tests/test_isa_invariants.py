"""Properties of the COMPILED streaming kernel that its correctness and speed lean on and that the source cannot promise
(hipcc cross-compiles without a GPU, so this runs in the CPU suite):

  * the cursor atomic of k_fused is issued by hand (inline asm) and its result awaited with a counted
    `s_waitcnt vmcnt(NU)`: that is only right if the NU plane loads of the prefetch sit BETWEEN the atomic and the wait in
    the instruction stream (fused_stream.inc, `process`);
  * the variants the configurations of BASELINE.json run (8 x 3 and 4 x 6 tiles, fp32 and fp16) use no scratch memory.
"""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not Path(HIPCC).exists() and not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "p3d_mc.s"
    cmd = [HIPCC if Path(HIPCC).exists() else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
           "--cuda-device-only", str(ROOT / "primitive3d_amd" / "csrc" / "p3d_mc.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return out.read_text()


def _kernels(asm, name):
    """{mangled name: body} of the kernels whose mangled name contains `name`."""
    out = {}
    for m in re.finditer(r"^(_Z\w*%s\w*):\s*;.*?$" % name, asm, re.M):
        start = m.end()
        end = asm.index(".amdhsa_kernel", start)
        out[m.group(1)] = asm[start:end]
    return out


def test_counted_wait_of_the_cursor_atomic(device_asm):
    kernels = _kernels(device_asm, "k_fused")
    assert len(kernels) >= 8   # 4 tile geometries x 2 sample types
    for name, body in kernels.items():
        lines = body.splitlines()
        atomics = [i for i, ln in enumerate(lines) if "global_atomic_add " in ln]
        assert len(atomics) == 2, (name, len(atomics))   # one per half of the unrolled plane loop
        for a in atomics:
            # the hand-written wait: "s_cmp_eq_u32 .. / s_cbranch_scc1 1f / s_waitcnt vmcnt(NU)"
            w = next(i for i in range(a, len(lines)) if "s_cbranch_scc1 1f" in lines[i])
            m = re.search(r"s_waitcnt vmcnt\((\d+)\)", lines[w + 1])
            assert m, (name, lines[w:w + 3])
            nu = int(m.group(1))
            loads = sum(1 for ln in lines[a:w] if re.search(r"\bbuffer_load_(dword|ushort|short)", ln))
            assert loads >= nu, f"{name}: {loads} plane loads between the atomic and its vmcnt({nu}) wait"
            assert "vmcnt(0)" in lines[w + 4], (name, lines[w:w + 6])   # the no-prefetch branch of the same statement


def test_no_scratch_in_the_benchmarked_variants(device_asm):
    # kernel descriptors: .amdhsa_kernel <name> ... .amdhsa_private_segment_fixed_size N
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", device_asm, re.S):
        name, desc = m.group(1), m.group(2)
        if "k_fused" not in name and "k_faces" not in name and "k_face_count_walk" not in name:
            continue
        if "Li2ELi15" in name:   # the short-row tile spills a few scalar registers (rows of at most 128 voxels)
            continue
        size = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc).group(1))
        assert size == 0, (name, size)
