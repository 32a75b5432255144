"""Properties of the COMPILED kernels that their speed leans on and that the source cannot promise (hipcc cross-compiles
without a GPU, so this runs in the CPU suite):

  * the cursor atomic of k_fused is ONE scalar-memory atomic per wave-plane (`s_atomic_add ... glc`, counted by lgkmcnt):
    a vector atomic in its place would queue its result behind the wave's vector stores and plane loads again;
  * its destination register is written ASYNCHRONOUSLY (the compiler does not know): nothing may read, copy or spill it
    between the atomic and the hand-written `s_waitcnt lgkmcnt(0)` that hands it over;
  * no other atomic lives in that kernel;
  * no k_fused / k_faces / k_face_count_walk variant uses scratch memory.
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
    import importlib.util
    spec = importlib.util.spec_from_file_location("p3d_build", ROOT / "primitive3d_amd" / "_build.py")
    build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(build)
    cmd = [HIPCC if Path(HIPCC).exists() else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
           *build.CAPI_EXTRA_FLAGS, "-S",
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


_WAIT = re.compile(r"#ASMSTART\s*\n\s*s_waitcnt lgkmcnt\(0\)\s*\n\s*;;#ASMEND")


def _sregs(text):
    """Scalar registers a line of assembly mentions: s7 -> {7}, s[16:17] -> {16, 17}."""
    regs = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(r) for r in re.findall(r"\bs(\d+)\b", text))
    return regs


def test_cursor_atomic_is_scalar(device_asm):
    kernels = _kernels(device_asm, "k_fused")
    assert len(kernels) == 8   # 4 tile geometries x 2 sample types (the round-4 DYN variants left in round 5)
    for name, body in kernels.items():
        # one cursor atomic per half of the unrolled plane loop, and behind each the spill area's (region-layout mode, the rare
        # wave-plane that does not fit: awaited inside its own asm statement)
        assert body.count("s_atomic_add ") == 4, name
        assert len(re.findall(r"s_atomic_add s\d+, s\[\d+:\d+\], 0x0 glc\n\s*s_waitcnt lgkmcnt\(0\)", body)) == 2, name
        assert "s_atomic_add_x2" not in body and "buffer_atomic" not in body, name
        assert "global_atomic" not in body, name
        # the results are awaited by the hand-written scalar-counter wait, right in front of the slot computation
        assert len(_WAIT.findall(body)) >= 2, name


def test_async_atomic_results_are_left_alone(device_asm):
    """`got` is an SGPR written by a scalar atomic that is still in flight when its asm statement ends.  Until the
    hand-written lgkmcnt(0) wait that hands it over nothing may read, copy or spill it -- a copy there reads the stale value
    (round 4: a second scalar atomic made the compiler move the operand to another register pair in front of the wait;
    round 6: a second hand-written wait on the same variable made it copy the register right in front of the real one).
    Checked on the text, block by block (block placement is the compiler's, and the register is reused for other values in
    blocks that run BEFORE the atomic but stand behind it in the file, so a scan in file order says nothing):
      A. from the atomic to the end of its basic block no instruction mentions its destination;
      B. in the basic block of every hand-written wait, between the block's label and the wait, no instruction READS a
         register an atomic of the kernel writes.
    What this cannot see -- a copy in a block in between -- shows as wrong vertex ids: tests/test_gpu_parity.py compares
    them on every path (the counting pass's records alone included, since the round-6 case was masked by a later pass)."""
    for name, body in _kernels(device_asm, "k_fused").items():
        lines = body.split("\n")
        atomics = [(i, _sregs(m.group(1))) for i, ln in enumerate(lines) for m in [re.match(r"\s*s_atomic_add (s\d+),", ln)] if m]
        assert len(atomics) >= 2, name
        dests = set().union(*(d for _, d in atomics))
        for i, dest in atomics:
            j = i + 1
            if "s_waitcnt lgkmcnt(0)" in lines[j]:
                continue   # (awaited inside its own asm statement: the spill area's atomic)
            while j < len(lines) and not re.match(r"\s*\.LBB\w+:", lines[j]):
                if "s_waitcnt lgkmcnt(0)" in lines[j] and "ASMSTART" in lines[j - 1]:
                    break
                assert not (dest & _sregs(lines[j].split(";")[0])), (name, "A", i, lines[i].strip(), j, lines[j].strip())
                j += 1
        waits = [j for j, ln in enumerate(lines) if "s_waitcnt lgkmcnt(0)" in ln and "ASMSTART" in lines[j - 1]]
        assert len(waits) >= 2, name
        for w in waits:
            j = w - 2   # (w - 1 is the ASMSTART line)
            while j >= 0 and not re.match(r"\s*\.LBB\w+:", lines[j]):
                ins = lines[j].split(";")[0].strip()
                m = re.match(r"(\S+)\s+([^,]+),(.*)", ins)
                if m and not ins.startswith(("s_cbranch", "s_branch")):
                    sources = _sregs(m.group(3))   # everything behind the first operand: what the instruction reads
                    assert not (dests & sources), (name, "B", w, j, lines[j].strip())
                j -= 1


def test_no_scratch_in_the_benchmarked_variants(device_asm):
    # kernel descriptors: .amdhsa_kernel <name> ... .amdhsa_private_segment_fixed_size N
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", device_asm, re.S):
        name, desc = m.group(1), m.group(2)
        if "k_fused" not in name and "k_faces" not in name and "k_face_count_walk" not in name:
            continue
        size = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc).group(1))
        assert size == 0, (name, size)
        seen += 1
    assert seen >= 12
