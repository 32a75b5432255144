"""bench.py prints ONE JSON line with the fields the driver and the judge read (task statement, section 4)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_bench_line_schema(gpu):
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "2", "--size", "128"],
                         capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mvoxels/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["config"]["adapter_mode"] == "hinted"   # the headline names the adapter's allocation policy
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["achieved"] > 0
    assert r["traffic"] is None  # measured for the 512^3 workload only
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "Mvoxels/s" and c["value"] > 0 and c["sample"]
    assert c["single_thread_value"] > 0 and "pymcubes" in c
    assert r["call_median_ms_hipevents"] > 0 and r["cold_first_call_ms"] > 0
    assert d["value"] > 0 and abs(d["value"] - 128 ** 3 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.01


def test_default_run_reports_the_other_configs(gpu):
    """The default run (the one the driver records) also times BASELINE.json's other single-GPU workloads for a few steps
    each, after and outside the headline measurement, under `other_configs`, and the call outside its steady state under
    `modes`; the headline fields stay what they were."""
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert d["metric"].startswith("Mvoxels/s on 512^3 fp32 SDF") and d["dtype"] == "f32" and d["steps"] == 3
    assert abs(d["value"] - 512 ** 3 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.01
    # the steady state of the default adapter mode: every timed call stored its vertices where they stay (the two calls of the
    # warm-up gave the region totals the layouts are made from)
    assert d["config"]["adapter_mode"] == "hinted" and d["config"]["calls_with_region_layout"] == 3
    # the headline workload carries the measured HBM traffic of its dominant kernel and the fraction of the peak that the
    # kernel reaches on THOSE bytes (>= the fraction on the algorithmic bytes)
    r = d["roofline"]
    assert r["traffic"] > r["alg_bytes_per_launch"] and r["frac"] <= r["traffic_frac"] < 1.0
    oc = d["other_configs"]
    assert sorted(oc) == ["bunny66", "c2", "c3_4oct", "c4_1gpu", "c4_rank_slab", "c5", "sphere200", "sphere512"]
    # the reference's own example inputs at their own sizes, with the counts its CUDA kernels give
    assert (oc["sphere200"]["vertices"], oc["sphere200"]["faces"]) == (11766, 23528)
    assert (oc["bunny66"]["vertices"], oc["bunny66"]["faces"]) == (13282, 26560)
    nvox = {"sphere200": 200 ** 3, "bunny66": 66 ** 3, "c2": 256 ** 3, "c5": 32 * 256 ** 3, "c4_1gpu": 1024 ** 3, "c3_4oct": 512 ** 3, "sphere512": 512 ** 3,
            "c4_rank_slab": 128 * 1024 ** 2}
    for k, c in oc.items():
        assert "error" not in c, (k, c)
        assert c["unit"] == "Mvoxels/s" and c["ms_per_step"] > 0 and c["vertices"] > 0 and c["faces"] > 0, (k, c)
        assert abs(c["value"] - nvox[k] / (c["ms_per_step"] * 1e-3) / 1e6) / c["value"] < 0.01
        if k != "c4_rank_slab":
            # (both present and sane; NOT compared with each other: on the small grids the kernel's event time of a few
            #  instrumented steps can exceed the call time of the timed ones when the box hiccups -- it did once in ~35 runs)
            assert 0 < c["whole_call_frac"] < 1.0 and 0 < c["k_fused_frac"] < 1.0, (k, c)
    # one rank's share of the 8-GPU run of the 1024^3 volume: the slab of rank 3 through the real
    # SlabExtractor.extract() with the transport stubbed, and what the 8-GPU run can reach at most without it
    rs = oc["c4_rank_slab"]
    assert rs["predicted_speedup_8gpu_no_transport"] == pytest.approx(oc["c4_1gpu"]["ms_per_step"] / rs["ms_per_step"], abs=0.01)
    assert 0 < rs["ms_per_step_without_standin_copies"] and rs["predicted_speedup_8gpu_without_standin_copies"] > 0
    assert 0.09 * oc["c4_1gpu"]["vertices"] < rs["vertices"] < 0.16 * oc["c4_1gpu"]["vertices"]   # an eighth of the surface
    assert "faces + rest of vertex copy" in rs["phases_ms_last_step"]
    assert oc["c5"]["dtype"] == "f16" and oc["c2"]["dtype"] == "f32"
    assert (oc["c2"]["vertices"], oc["c2"]["faces"]) == (204670, 409336)   # (the resampled bunny: a closed surface, V - F/2 = 2)
    # the four-octave field has about twice the surface of the single-octave one; the sphere of the reference's
    # own example at n = 512 is a closed surface of radius 64 (V - F/2 = 2)
    assert oc["c3_4oct"]["vertices"] > 1.5 * d["config"]["vertices"]
    assert oc["sphere512"]["vertices"] - oc["sphere512"]["faces"] // 2 == 2 and oc["sphere512"]["faces"] % 2 == 0
    # what a call costs outside the steady state
    m = d["modes"]
    assert sorted(m) == ["exact", "fresh_grid", "hint_miss", "sparse_dense", "two_phase"]
    for k, c in m.items():
        assert "error" not in c, (k, c)
    # Structural facts only: which path every mode took and that it produced the headline's mesh.  Timings are
    # reported, not compared -- medians of a handful of synchronised calls on a shared box prove nothing about the code.
    # (exact mode: the reference's count -> read -> allocate -> emit order; one pass over the field once its scratch guess holds)
    assert m["exact"]["streaming_passes_per_call"] == 1 and m["exact"]["ms_per_step"] > 0
    assert (m["exact"]["vertices"], m["exact"]["faces"]) == (d["config"]["vertices"], d["config"]["faces"])
    assert m["sparse_dense"]["streaming_passes_per_call"] == 1 and m["sparse_dense"]["dense_call_ms"] > 0 and m["sparse_dense"]["sparse_call_ms"] > 0
    # (a too-small guess for the OUTPUT buffers: one pass over the field, faces and compaction twice)
    assert m["hint_miss"]["streaming_passes_per_call"] == 1 and m["hint_miss"]["emissions_per_call"] == 2
    # (four distinct grids in turn: one pass each; the kernel-level fraction next to the headline's)
    fg = m["fresh_grid"]
    assert fg["grids"] == 4 and fg["bytes_rotated"] == 4 * 512 ** 3 * 4 and fg["streaming_passes_per_call"] == 1
    assert 0 < fg["whole_call_frac"] < 1.0 and 0 < fg["k_fused_frac"] < 1.0 and r["frac_fresh"] == fg["k_fused_frac"]
    # (the literal count -> read -> allocate -> emit binding of INTEGRATION.md: same mesh, its cost on record)
    assert (m["two_phase"]["vertices"], m["two_phase"]["faces"]) == (d["config"]["vertices"], d["config"]["faces"])
    assert m["two_phase"]["ms_per_step"] > 0
    # the traffic is measured in the run itself (two rocprofv3 --pmc child passes) -- or, where the profiler is not there, the
    # committed figure is reported with the build it was taken on
    assert r["traffic_source"]
    if r["traffic_source"].startswith("this run"):
        assert r["traffic_read"] + r["traffic_write"] == r["traffic"]
        # (generous bounds -- the counters are hardware- and SKU-specific: the field is read at least nearly once and not
        #  twice; the vertex rows are written at least once)
        assert 0.9 * r["alg_bytes_per_launch"] <= r["traffic_read"] < 2.0 * r["alg_bytes_per_launch"]
        assert 12 * d["config"]["vertices"] <= r["traffic_write"] < 4 * 12 * d["config"]["vertices"]
        # the whole call over the fabric (all three kernels): at least what it must read and write -- the field, V rows, F rows
        # -- and (every vertex row stored once in the steady state) well under the 1.06 GB it took through the scratch
        must = r["alg_bytes_per_launch"] + 12 * d["config"]["vertices"] + 12 * d["config"]["faces"]
        assert r["call_traffic"] == r["call_traffic_read"] + r["call_traffic_write"] and must <= r["call_traffic"] < 1.0e9
        assert r["traffic"] < r["call_traffic"] and 0 < r["call_traffic_frac"] < 1.0
    else:
        assert r["traffic_build"]
