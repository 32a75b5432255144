"""CPU tests of the oracle itself: pinned against the reference's known answers (tests/golden/
known_answers.json <- SURVEY.md section 4), an independent numpy count, mesh invariants, and the
committed canonical meshes."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import canonical_mesh, oracle_count, oracle_extract, oracle_tri_table
from oracle.np_counts import np_count, tri_counts
from primitive3d_amd.fields import sphere_grid
from tests.cases import small_cases

GOLD = Path(__file__).resolve().parent / "golden"
KNOWN = json.loads((GOLD / "known_answers.json").read_text())


def _edges(faces):
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    return e


def _mesh_stats(v, f):
    e = _edges(f.astype(np.int64))
    und = np.sort(e, axis=1)
    uniq, cnt = np.unique(und, axis=0, return_counts=True)
    duniq, dcnt = np.unique(e, axis=0, return_counts=True)
    a, b, c = (v[f[:, i]].astype(np.float64) for i in range(3))
    vol = float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0)
    area2 = np.linalg.norm(np.cross(b - a, c - a), axis=1)
    return {"E": len(uniq), "edge_use_ok": bool((cnt == 2).all()), "dir_edge_once": bool((dcnt == 1).all()),
            "volume": vol, "zero_area": int((area2 == 0).sum()),
            "dup_pos": int(len(v) - len(np.unique(v, axis=0)))}


def test_case_table_bytes_match_reference_sha():
    t = oracle_tri_table()
    assert hashlib.sha256(t.tobytes()).hexdigest() == KNOWN["tri_table_sha256"]
    assert (t[:, 15] == -1).all()
    hist = np.bincount(tri_counts(), minlength=6)
    assert {str(i): int(n) for i, n in enumerate(hist)} == KNOWN["tri_count_histogram"]


def test_sphere200_known_answers_and_manifold():
    k = KNOWN["cases"]["sphere200"]
    g = sphere_grid(200)  # examples/sphere.py:8-9, int64; the wrapper casts to f32
    assert oracle_count(g, k["thresh"]) == (k["V"], k["F"])
    v, f, _ = oracle_extract(g, k["thresh"])
    st = _mesh_stats(v, f)
    assert st["E"] == k["E"] and st["edge_use_ok"] and st["dir_edge_once"]
    assert k["V"] - st["E"] + k["F"] == k["euler"]
    assert st["zero_area"] == k["zero_area_triangles"] and st["dup_pos"] == k["duplicate_positions"]
    assert round(st["volume"]) == k["signed_volume"]
    assert v.min(0).tolist() == k["bbox_min"] and v.max(0).tolist() == k["bbox_max"]
    assert np_count(g, 0)[2] == k["active_cells"]


def test_bunny66_known_answers():
    k = KNOWN["cases"]["bunny66"]
    g = np.load(GOLD / "bunny66.npy")
    assert g.shape == (66, 66, 66) and g.dtype == np.float32
    assert oracle_count(g, 0) == (k["V"], k["F"])
    v, f, _ = oracle_extract(g, 0)
    st = _mesh_stats(v, f)
    assert st["E"] == k["E"] and st["edge_use_ok"] and st["dir_edge_once"] and k["V"] - st["E"] + k["F"] == 2
    assert st["zero_area"] == 0 and st["dup_pos"] == 0
    assert np.allclose(v.min(0), k["vmin"], rtol=0, atol=1e-6) and np.allclose(v.max(0), k["vmax"], rtol=0, atol=1e-5)
    assert v.min(0).astype(np.float32).tolist() == np.float32(k["vmin"]).tolist()
    assert np_count(g, 0)[2] == k["active_cells"]


def test_sphere64_known_answers():
    k = KNOWN["cases"]["sphere64"]
    assert oracle_count(sphere_grid(64), 0) == (k["V"], k["F"])


@pytest.mark.parametrize("name", sorted(small_cases().keys()))
def test_numpy_count_agrees_with_c_oracle(name):
    g, thresh, _, _ = small_cases()[name]
    v, f, _ = np_count(g, thresh)
    assert oracle_count(g, thresh) == (v, f)


def test_committed_canonical_meshes_reproduce():
    z = np.load(GOLD / "small_meshes.npz")
    for name, (g, thresh, lower, upper) in small_cases().items():
        k, v, f = canonical_mesh(*oracle_extract(g, thresh, lower, upper))
        assert np.array_equal(k, z[name + "__keys"]), name
        assert np.array_equal(f, z[name + "__faces"]), name
        assert np.array_equal(v, z[name + "__verts"], equal_nan=True), name


def test_epilogue_quirk_and_separate_rounding():
    """marching_cubes.cu:293-298: y scale uses upper[2]-lower[1]; multiply then add, each rounded to fp32."""
    g, thresh, lower, upper = small_cases()["noise_5x7x9_box"]
    v, _, keys = oracle_extract(g, thresh, lower, upper)
    # grid-unit positions: x,y from a call whose x/y scales are exactly 1 (upper[2]=7 feeds the y scale),
    # z from a call whose z scale is exactly 1
    ra, _, keys0 = oracle_extract(g, thresh, [0, 0, 0], [5, 0, 7])
    rb, _, _ = oracle_extract(g, thresh, [0, 0, 0], [5, 0, 9])
    raw = np.stack([ra[:, 0], ra[:, 1], rb[:, 2]], axis=1)
    assert np.array_equal(keys, keys0)
    lo, up = np.float32(lower), np.float32(upper)
    scale = np.array([(up[0] - lo[0]) / np.float32(5), (up[2] - lo[1]) / np.float32(7), (up[2] - lo[2]) / np.float32(9)],
                     dtype=np.float32)
    expect = (raw * scale).astype(np.float32) + lo
    assert np.array_equal(v, expect.astype(np.float32))


def test_strict_greater_and_nan_are_outside():
    g = np.zeros((2, 2, 2), np.float32)  # every sample == thresh: nothing is inside
    assert oracle_count(g, 0.0) == (0, 0)
    g[0, 0, 0] = np.nan
    assert oracle_count(g, -1.0)[0] == 3  # NaN > t is false: the NaN corner is the only outside sample


def test_bitsliced_count_network_is_current(tmp_path, monkeypatch):
    """csrc/tri_count_bitsliced.inc is what tools/gen_tri_count_bitsliced.py generates from the committed case table
    (the generator also verifies the network against the table for all 256 masks)."""
    import importlib.util
    root = Path(__file__).resolve().parents[1]
    spec = importlib.util.spec_from_file_location("gen_bs", root / "tools" / "gen_tri_count_bitsliced.py")
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    out = tmp_path / "tri_count_bitsliced.inc"
    monkeypatch.setattr(gen, "OUT", out)
    assert gen.main() == 0
    assert out.read_text() == (root / "primitive3d_amd" / "csrc" / "tri_count_bitsliced.inc").read_text()


@pytest.mark.parametrize("threads", [0, 3])
def test_openmp_oracle_is_identical_to_the_serial_one(threads):
    """p3d_oracle_extract_mt (bench.py's all-host-cores CPU baseline) must give the serial oracle's output, values
    and order, on every seeded case and on the reference's bunny input."""
    from pathlib import Path
    cases = dict(small_cases())
    cases["bunny66"] = (np.load(Path(__file__).parent / "golden" / "bunny66.npy"), 0.0, None, None)
    for name, (g, thresh, lower, upper) in cases.items():
        a = oracle_extract(g, thresh, lower, upper)
        b = oracle_extract(g, thresh, lower, upper, threads=threads)
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True), name


def test_fastdiv_is_exact(tmp_path):
    """primitive3d_amd/csrc/fastdiv.h (the face kernel's divisions by run-time constants as multiply-high + shifts, magic
    numbers made on the host) against the hardware division: every d < 3000 with every n < 70000 plus random and extreme n,
    and two million random (n, d) pairs -- compiled here with the host compiler."""
    import subprocess
    root = Path(__file__).resolve().parents[1]
    src = tmp_path / "fd.cpp"
    src.write_text('''
#include <cstdio>
#include <cstdlib>
#include "fastdiv.h"
int main() {
    unsigned long long bad = 0;
    srand(1);
    for (uint32_t d = 1; d < 3000; ++d) {
        const FastDiv f = make_fastdiv(d);
        for (uint32_t n = 0; n < 70000; ++n) bad += fd_div(n, f) != n / d;
        for (int k = 0; k < 500; ++k) { uint32_t n = ((uint32_t)rand() << 16) ^ (uint32_t)rand(); bad += fd_div(n, f) != n / d; }
        const uint32_t edge[] = {0xffffffffu, 0x7fffffffu, 0x80000000u, d * 7u - 1u, d * 1000003u};
        for (uint32_t n : edge) bad += fd_div(n, f) != n / d;
    }
    for (int k = 0; k < 2000000; ++k) {
        uint32_t d = (((uint32_t)rand() << 16) ^ (uint32_t)rand()) | 1u;
        if (k & 1) d >>= (rand() % 31);
        if (!d) d = 1;
        const FastDiv f = make_fastdiv(d);
        const uint32_t n = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
        bad += fd_div(n, f) != n / d;
    }
    printf("%llu\\n", bad);
    return bad != 0;
}
''')
    exe = tmp_path / "fd"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", str(root / "primitive3d_amd" / "csrc"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "0", out.stdout + out.stderr


def test_half_round_down_is_exact(tmp_path):
    """primitive3d_amd/csrc/half_round.h (the threshold of the 16-bit compares on fp16 grids): for every one of the 65536
    fp16 bit patterns v and a few thousand thresholds t -- representable or not, subnormal, beyond the range, infinite,
    NaN -- `v > half_round_down(t)` == `float(v) > t`.  Compiled with the ROCm clang (the host compiler of hipcc)."""
    import shutil
    import subprocess
    clang = shutil.which("amdclang++") or "/opt/rocm/lib/llvm/bin/clang++"
    if not Path(clang).exists():
        pytest.skip("no clang with _Float16 on this machine")
    root = Path(__file__).resolve().parents[1]
    src = tmp_path / "hr.cpp"
    src.write_text('''
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <limits>
#include "half_round.h"
int main() {
    float ts[4000];
    int n = 0;
    const float fixed[] = {0.1f, -0.1f, 1e-10f, -1e-10f, 5.9604645e-8f, 6e-8f, -6e-8f, 0.0f, -0.0f, 65504.0f, 65519.9f, 65520.0f,
                           70000.0f, -65504.0f, -65520.0f, -70000.0f, INFINITY, -INFINITY, NAN, 0.333251953125f,
                           1.0009765625f, 6.1e-5f, -6.1e-5f, std::numeric_limits<float>::max(), -std::numeric_limits<float>::max()};
    for (float f : fixed) ts[n++] = f;
    srand(3);
    const float scales[] = {1e-8f, 1e-5f, 1e-3f, 1.0f, 100.0f, 1e5f};
    while (n < 4000) ts[n++] = ((float)rand() / RAND_MAX - 0.5f) * 2.0f * scales[rand() % 6];
    unsigned long long bad = 0;
    for (int i = 0; i < n; ++i) {
        const _Float16 t16 = __builtin_bit_cast(_Float16, (unsigned short)half_round_down(ts[i]));
        for (unsigned b = 0; b < 65536; ++b) {
            const _Float16 v = __builtin_bit_cast(_Float16, (unsigned short)b);
            bad += ((float)v > ts[i]) != (v > t16);
        }
    }
    printf("%llu\\n", bad);
    return bad != 0;
}
''')
    exe = tmp_path / "hr"
    subprocess.check_call([clang, "-O1", "-std=c++17", "-I", str(root / "primitive3d_amd" / "csrc"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "0", out.stdout + out.stderr
