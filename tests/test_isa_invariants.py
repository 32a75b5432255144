"""Properties of the COMPILED kernels that their speed leans on and that the source cannot promise (hipcc cross-compiles
without a GPU, so this runs in the CPU suite):

  * the cursor atomic of k_fused is ONE scalar-memory atomic per wave-plane (`s_atomic_add ... glc`, counted by lgkmcnt):
    a vector atomic in its place would queue its result behind the wave's vector stores and plane loads again;
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


def test_cursor_atomic_is_scalar(device_asm):
    kernels = _kernels(device_asm, "k_fused")
    assert len(kernels) >= 8   # 4 tile geometries x 2 sample types
    for name, body in kernels.items():
        assert body.count("s_atomic_add ") == 2, name        # one per half of the unrolled plane loop
        assert "global_atomic" not in body and "buffer_atomic" not in body, name
        # its result is awaited by the hand-written scalar-counter wait, right in front of the slot computation
        assert len(re.findall(r"#ASMSTART\s*\n\s*s_waitcnt lgkmcnt\(0\)\s*\n\s*;;#ASMEND", body)) >= 2, name


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
