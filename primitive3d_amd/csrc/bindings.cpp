// bindings.cpp -- pybind module `libPrim3D`: the same Python-visible surface as the reference's
// src/pybind/bindings.cpp:13-32 (marching_cubes, save_mesh_as_ply, enable_optix, test, and the two
// ray-casting names the package imports at module load), implemented as a thin adapter over the C ABI
// in include/p3d_mc.h.  No kernels live here: torch is used for device memory and the current stream.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/p3d_mc.h"
#include "../../include/p3d_rc.h"

namespace py = pybind11;
using torch::Tensor;

namespace {

// output-size hints of the one-pass path, per (device, grid shape): the counts of the last kHintCalls calls on that shape.
// The buffers of a call are sized for the LARGEST of them, so a sparse frame between two dense ones (per-frame extraction
// of a changing field) does not make the next dense frame stream the field twice; after kHintCalls sparse calls in a row
// the dense size is forgotten again.
using CapKey = std::tuple<int, int64_t, int64_t, int64_t>;
constexpr int kHintCalls = 4;
struct CapHint {
    int64_t nv[kHintCalls] = {0, 0, 0, 0}, nf[kHintCalls] = {0, 0, 0, 0};
    int n = 0;        // calls recorded (the ring's next slot is n % kHintCalls)
    int64_t peak_v = 0;   // the most vertices any call on this shape produced: bounds the vertex SCRATCH (below)
    int slack_q = 5;  // headroom of every scratch region in quarters (5 = 1.25x); doubles after a region overflow ...
    int clean_calls = 0;   // ... and halves again after kSlackDecayCalls calls in a row without one (it used to stay for good)
    // the predicted region layout (p3d_mc_slab.region_first_rows): the 32 region totals of the last TWO calls on the shape (a
    // layout is made from the last call's when they moved by less than 1/kLayoutDrift of the vertices between the two: a field
    // that stands still or changes slowly), their vertex counts, and calls left to sit out after a spill area overflowed
    int64_t regions[32] = {0}, prev_regions[32] = {0};
    int64_t last_v = -1, prev_v = -1;
    int layout_cooldown = 0;
    uint64_t last_use = 0;
    int64_t max_v() const { return *std::max_element(nv, nv + kHintCalls); }
    int64_t max_f() const { return *std::max_element(nf, nf + kHintCalls); }
    void record(int64_t v, int64_t f) {
        nv[n % kHintCalls] = v;
        nf[n % kHintCalls] = f;
        peak_v = std::max(peak_v, v);
        ++n;
    }
    // the region totals of a pass (null: the pass left none that a layout could be made from)
    void record_regions(const int64_t* r, int64_t v) {
        std::copy(regions, regions + 32, prev_regions);
        prev_v = last_v;
        if (r) std::copy(r, r + 32, regions);
        last_v = r ? v : -1;
    }
    // rows that changed region between the last two calls, as a measure of how far off a layout made from the last call will be
    bool regions_stand_still() const {
        // 1/64 of the vertices: an iso level creeping by 1 % of the field's range per call moves 0.5-0.7 %, a different random field
        // 1.5-4 % -- and from about 2.5 % on the scratch is the faster way (profiles/r06/layout_ab.txt)
        constexpr int64_t kLayoutDrift = 64;
        if (last_v <= 0 || prev_v <= 0) return false;
        int64_t moved = 0;
        for (int r = 0; r < 32; ++r) moved += std::abs(regions[r] - prev_regions[r]);
        return moved * kLayoutDrift <= last_v;
    }
};
std::map<CapKey, CapHint> g_cap_hint;   // at most kMaxHints shapes; when full, the least recently used one goes
std::mutex g_cap_mu;
uint64_t g_cap_clock = 0;
constexpr size_t kMaxHints = 256;

void check_rc(int rc, const char* what) {
    TORCH_CHECK(rc == P3D_OK, what, " failed (", rc, "): ", p3d_last_error());
}

// Replaces prim3d::marching_cubes, src/prim3d/Utility/marching_cubes.cu:212-305.
// Same argument meaning and error behaviour: CUDA(HIP)+contiguous checks raise c10::Error with the
// reference's messages (Core/common.h:63-68), ndimension()==3, float32 data (data_ptr<float>()
// would throw in the reference, :243).  lower/upper lengths are TORCH_CHECKed (the reference only
// asserts, :221-222, compiled out in Release).
std::vector<Tensor> marching_cubes(const Tensor& density_grid, const float thresh, const std::vector<float> lower,
                                   const std::vector<float> upper) {
    TORCH_CHECK(density_grid.is_cuda(), "density_grid must be a CUDA tensor");
    TORCH_CHECK(density_grid.is_contiguous(), "density_grid must be contiguous");
    TORCH_CHECK(density_grid.ndimension() == 3);
    TORCH_CHECK(lower.size() == 3 && upper.size() == 3, "lower and upper must have 3 elements");
    // float32 as in the reference (data_ptr<float>() throws there on anything else, :243) -- and float16, which the reference's
    // wrapper would have up-cast before it got here (marching_cubes.py:87): the library reads it as it is (half the bytes,
    // no copy) and compares exactly like the up-cast (`float(v) > thresh`), so the mesh is the one the up-cast gives
    const bool half_grid = density_grid.scalar_type() == torch::kHalf;
    TORCH_CHECK(half_grid || density_grid.scalar_type() == torch::kFloat, "expected scalar type Float but found ",
                density_grid.scalar_type());
    const int dtype = half_grid ? P3D_F16 : P3D_F32;

    const c10::hip::HIPGuardMasqueradingAsCUDA guard(density_grid.device());
    void* stream = (void*)c10::hip::getCurrentHIPStream(density_grid.device().index()).stream();
    const int64_t rx = density_grid.size(0), ry = density_grid.size(1), rz = density_grid.size(2);
    const auto dev = density_grid.device();

    size_t ws_bytes = 0;
    check_rc(p3d_mc_workspace_bytes(rx, ry, rz, &ws_bytes), "p3d_mc_workspace_bytes");
    Tensor ws = torch::empty({(int64_t)ws_bytes}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
    const auto vopt = torch::TensorOptions().dtype(torch::kFloat).device(dev);
    const auto fopt = torch::TensorOptions().dtype(torch::kInt).device(dev);
    const void* grid = density_grid.data_ptr();
    Tensor vertices, faces;
    int64_t nv = 0, nf = 0;

    // P3D_MC_MODE (the one environment variable this module reads; INTEGRATION.md section 2 says why the default is what it is):
    //   hinted (default)  rows [0, V) / [0, F) of buffers sized from the last calls on this grid shape, which may be up to
    //                     1/8 + 4096 rows longer; no host round trip inside the call.  On a field that stands still or changes
    //                     slowly the vertices are stored where they stay (the predicted region layout below), otherwise they
    //                     go through a scratch tensor of this call
    //   scratch           hinted, always through the scratch tensor (what `hinted` was until round 6; for A/B measurements)
    //   exact             freshly allocated tensors that own exactly V / F rows, like the reference's torch::zeros({V,3})
    //                     (marching_cubes.cu:260-263), in the reference's order count -> read -> allocate -> emit (below);
    //                     no state is carried from call to call except the vertex-scratch sizing
    static const bool exact_mode = [] {
        const char* m = std::getenv("P3D_MC_MODE");
        TORCH_CHECK(!m || !*m || std::string(m) == "exact" || std::string(m) == "hinted" || std::string(m) == "scratch",
                    "P3D_MC_MODE must be 'hinted', 'scratch' or 'exact', got '", m, "'");
        return m && std::string(m) == "exact";
    }();
    static const bool layout_off = [] { const char* m = std::getenv("P3D_MC_MODE"); return m && std::string(m) == "scratch"; }();

    // One streaming pass into buffers of the given capacities (0,0 = count only).  The vertex scratch is cut into
    // 32 independently filled regions: `slack` is the headroom per region, and every region can hold 8192 rows
    // because a small output may come from very few wave-planes.  Returns true when everything fitted.
    bool region_overflow = false, id_overflow = false, scratch_overflow = false;
    int64_t last_regions[32] = {0};   // the streaming kernel's 32 region totals of the last pass (what a layout is made from)
    bool have_last_regions = false;
    // The scratch is sized for the LARGER of the expectation and a guess: it is memory of this call only, and a field that
    // turns out denser than the last few calls on its shape -- or than the guess of a first call -- then only outgrows the
    // OUTPUT buffers: the faces and the compaction run again into larger ones (below), the field is not streamed twice.
    // The guess: a vertex per 16 voxels.  Where that is more than 128 MiB of scratch (grids beyond ~560^3) and the shape has a
    // history, it follows the history instead: twice the most any call on the shape has produced, at least 128 MiB, never
    // more than a vertex per 16 voxels -- an object's SDF in a 1024^3 box costs 128 MiB of scratch per call, not 0.8 GB,
    // and a dense frame after a run of sparse ones still finds room on every grid up to that size.
    Tensor scratch;
    int64_t scratch_rows = 0;
    int64_t scratch_guess = rx * ry * rz / 16;
    auto size_scratch = [&](int64_t capv, int64_t slack_num, int64_t slack_den, bool generous) {
        const int64_t expect = generous ? std::max<int64_t>(capv, scratch_guess) : capv;
        const int64_t per_region = (expect + 31) / 32;
        scratch_rows = 32 * std::max<int64_t>(per_region * slack_num / slack_den + 256, std::min<int64_t>(expect, 8192));
        scratch = torch::empty({scratch_rows, 3}, vopt);
    };
    auto run_pass = [&](int64_t capv, int64_t capf, int64_t slack_num, int64_t slack_den, bool generous) {
        scratch_rows = 0;
        if (capv > 0) {
            size_scratch(capv, slack_num, slack_den, generous);
            vertices = torch::empty({capv, 3}, vopt);
        }
        if (capf > 0) faces = torch::empty({capf, 3}, fopt);
        check_rc(p3d_mc_extract_fused(grid, dtype, rx, ry, rz, thresh, lower.data(), upper.data(), nullptr, nullptr,
                                      ws.data_ptr(), capv ? vertices.data_ptr<float>() : nullptr, capv,
                                      capv ? scratch.data_ptr<float>() : nullptr, scratch_rows,
                                      capf ? faces.data_ptr<int32_t>() : nullptr, capf, stream),
                 "p3d_mc_extract_fused");
        int32_t overflow = 0;
        check_rc(p3d_mc_read_counts_ex(ws.data_ptr(), &nv, &nf, &overflow, last_regions, stream), "p3d_mc_read_counts_ex");
        have_last_regions = true;
        scratch_overflow = (overflow & 1) != 0;
        region_overflow = scratch_overflow && nv <= capv;  // the total fitted, the split over the regions did not
        id_overflow = (overflow & 2) != 0;                     // a region outgrew its id space: renumber (below)
        return nv <= capv && nf <= capf && !overflow;
    };

    // Output sizes: the reference counts first (marching_cubes.cu:242-252).  Here the default is to guess them from
    // the last calls on a grid of this shape (first call: a density guess) and to stream the field ONCE; a guess that was
    // too small costs a second face launch and compaction, or -- when the scratch was too small as well -- a second pass
    // over the field into exactly sized buffers.  P3D_MC_MODE=exact keeps the reference's order (below).
    const CapKey key{dev.index(), rx, ry, rz};
    int64_t capv = 0, capf = 0;
    int slack_q = 5;
    {
        std::lock_guard<std::mutex> g(g_cap_mu);
        auto it = g_cap_hint.find(key);
        if (it != g_cap_hint.end()) {
            capv = it->second.max_v() + it->second.max_v() / 8 + 4096;
            capf = it->second.max_f() + it->second.max_f() / 8 + 4096;
            slack_q = it->second.slack_q;
            constexpr int64_t kFreeRows = (int64_t(128) << 20) / 12;
            if (scratch_guess > kFreeRows)
                scratch_guess = std::min<int64_t>(scratch_guess, std::max<int64_t>(2 * it->second.peak_v + 4096, kFreeRows));
        } else {
            capv = std::max<int64_t>(4096, rx * ry * rz / 16);
            capf = 2 * capv;
        }
    }
    // Steady state of the default mode: the streaming kernel stores every vertex where it STAYS.  The 32 output regions are laid
    // out inside the vertex tensor itself from the region totals of the last call on this shape (exactly: a field extracted twice
    // in a row moves nothing; a changed field lets the regions that grew spill into eight small areas behind them, and the few
    // rows that end up beyond V are moved down by blocks riding in the counting launch).  No scratch tensor, no copy of the
    // vertex rows.  Tried when fewer than 1/64 of the vertices changed region between the last two calls on the shape (beyond that
    // the regions that grew send every wave-plane to a second cursor and the scratch is the faster way); a spill area overflowing
    // (flag 4) costs a second pass over the field into exactly sized tensors and two calls in the scratch mode below.
    int64_t lay_regions[32];
    bool use_layout = false;
    {
        std::lock_guard<std::mutex> g(g_cap_mu);
        auto it = g_cap_hint.find(key);
        if (!exact_mode && !layout_off && it != g_cap_hint.end()) {
            CapHint& h = it->second;
            if (h.layout_cooldown > 0) {
                --h.layout_cooldown;
            } else if (h.regions_stand_still()) {
                use_layout = true;
                std::copy(h.regions, h.regions + 32, lay_regions);
            }
        }
    }
    if (use_layout) {
        int64_t expect = 0;
        for (int r = 0; r < 32; ++r) expect += lay_regions[r];
        const int64_t spill = expect / 8 + 4096;   // the eight areas together
        use_layout = expect + spill < (int64_t)0x7fff0000;   // (rows are 32-bit in the table; beyond that: the scratch route)
        for (int r = 0; r < 32; ++r) use_layout = use_layout && lay_regions[r] <= (int64_t(1) << 28);   // (the library's limit per region)
        use_layout = use_layout && spill / 8 < (int64_t(1) << 28);
    }
    if (use_layout) {
        uint32_t first[41];
        int64_t row = 0;
        for (int r = 0; r < 32; ++r) {
            first[r] = (uint32_t)row;
            row += lay_regions[r];
        }
        const int64_t spill = row / 8 + 4096;
        for (int a = 0; a < 8; ++a) {
            first[32 + a] = (uint32_t)row;
            row += (spill + 7 - a) / 8;
        }
        first[40] = (uint32_t)row;
        {
            const int64_t capv_l = row;
            vertices = torch::empty({capv_l, 3}, vopt);
            if (capf > 0) faces = torch::empty({capf, 3}, fopt);
            p3d_mc_slab lay{};
            lay.region_first_rows = first;
            check_rc(p3d_mc_extract_fused(grid, dtype, rx, ry, rz, thresh, lower.data(), upper.data(), nullptr, &lay, ws.data_ptr(),
                                          vertices.data_ptr<float>(), capv_l, nullptr, 0, capf ? faces.data_ptr<int32_t>() : nullptr,
                                          capf, stream),
                     "p3d_mc_extract_fused");
            int32_t overflow = 0;
            int64_t regs[32];
            check_rc(p3d_mc_read_counts_ex(ws.data_ptr(), &nv, &nf, &overflow, regs, stream), "p3d_mc_read_counts_ex");
            const bool fitted = !overflow && nf <= capf;
            {
                std::lock_guard<std::mutex> g(g_cap_mu);
                CapHint& h = g_cap_hint[key];
                h.record(nv, nf);
                h.record_regions(overflow & 2 ? nullptr : regs, nv);
                if (overflow & 4) h.layout_cooldown = 2;
                h.last_use = ++g_cap_clock;
            }
            if (!fitted) {   // (a spill area overflowed, or the faces outgrew their tensor: exactly sized tensors, a second pass)
                vertices = torch::empty({nv, 3}, vopt);
                faces = torch::empty({nf, 3}, fopt);
                if (nv > 0 && nf > 0)
                    check_rc(p3d_mc_emit(grid, dtype, rx, ry, rz, thresh, lower.data(), upper.data(), nullptr, nullptr, ws.data_ptr(),
                                         vertices.data_ptr<float>(), nv, faces.data_ptr<int32_t>(), nf, nullptr, stream),
                             "p3d_mc_emit");
                return {vertices, faces};
            }
            auto fit_l = [&](Tensor& t, int64_t n, int64_t cap) {
                if (n != cap) t = 2 * n >= cap ? t.narrow(0, 0, n) : t.narrow(0, 0, n).clone();
            };
            fit_l(vertices, nv, capv_l);
            if (capf > 0) fit_l(faces, nf, capf);
            else faces = torch::empty({0, 3}, fopt);
            return {vertices, faces};
        }
    }
    // P3D_MC_MODE=exact: the reference's order -- count, read (V, F), allocate exactly, emit (marching_cubes.cu:242-287) --
    // with ONE pass over the field when the guess holds for the vertex SCRATCH (the only thing sized by it here): the field
    // is streamed into the scratch regions (part 3), the faces are counted (part 4), the host reads the totals, allocates
    // the two tensors and the face launch writes into them (part 6).  A scratch region that overflowed, or ids beyond a
    // region's range: the two-pass route below (count-only pass, then an exactly sized pass).
    bool exact_done = false;
    if (exact_mode) {
        size_scratch(capv, slack_q, 4, true);
        p3d_mc_slab parts{};
        parts.part = 3;
        auto call = [&](float* v, int64_t cv, int32_t* f, int64_t cf) {
            check_rc(p3d_mc_extract_fused(grid, dtype, rx, ry, rz, thresh, lower.data(), upper.data(), nullptr, &parts,
                                          ws.data_ptr(), v, cv, scratch.data_ptr<float>(), scratch_rows, f, cf, stream),
                     "p3d_mc_extract_fused");
        };
        call(nullptr, 0, nullptr, 0);
        parts.part = 4;
        call(nullptr, 0, nullptr, 0);
        int32_t overflow = 0;
        check_rc(p3d_mc_read_counts(ws.data_ptr(), &nv, &nf, &overflow, stream), "p3d_mc_read_counts");
        scratch_overflow = region_overflow = (overflow & 1) != 0;
        id_overflow = (overflow & 2) != 0;
        if (!overflow) {
            vertices = torch::empty({nv, 3}, vopt);
            faces = torch::empty({nf, 3}, fopt);
            if (nv > 0) {
                parts.part = 6;
                call(vertices.data_ptr<float>(), nv, nf ? faces.data_ptr<int32_t>() : nullptr, nf);
            }
            exact_done = true;
        }
    }
    // (exact mode whose scratch guess did not hold: the counts are right all the same -- straight to the exactly sized pass)
    bool ok = exact_done || (!exact_mode && run_pass(capv, capf, slack_q, 4, true));
    {
        // a field whose vertices are spread unevenly over the 32 regions gets more headroom per region next time
        std::lock_guard<std::mutex> g(g_cap_mu);
        if (g_cap_hint.size() >= kMaxHints && g_cap_hint.find(key) == g_cap_hint.end()) {
            auto oldest = g_cap_hint.begin();
            for (auto it = g_cap_hint.begin(); it != g_cap_hint.end(); ++it)
                if (it->second.last_use < oldest->second.last_use) oldest = it;
            g_cap_hint.erase(oldest);
        }
        CapHint& h = g_cap_hint[key];
        h.record(nv, nf);
        h.record_regions(have_last_regions && !id_overflow ? last_regions : nullptr, nv);
        constexpr int kSlackDecayCalls = 64;
        if (region_overflow) {
            h.slack_q = std::min(2 * slack_q, 32);
            h.clean_calls = 0;
        } else if (++h.clean_calls >= kSlackDecayCalls && h.slack_q > 5) {
            h.slack_q = std::max(5, h.slack_q / 2);
            h.clean_calls = 0;
        }
        h.last_use = ++g_cap_clock;
    }
    if (exact_done) return {vertices, faces};
    if (!ok && !exact_mode && !scratch_overflow && !id_overflow && nv > 0 && scratch_rows > 0) {
        // only the output buffers were too small: every vertex is in the scratch, the workspace is complete -- the face
        // launch and the compaction again, into exactly sized tensors (p3d_mc_slab.part = 6 behind a whole call)
        vertices = torch::empty({nv, 3}, vopt);
        faces = torch::empty({nf, 3}, fopt);
        p3d_mc_slab again{};
        again.part = 6;
        check_rc(p3d_mc_extract_fused(grid, dtype, rx, ry, rz, thresh, lower.data(), upper.data(), nullptr, &again,
                                      ws.data_ptr(), vertices.data_ptr<float>(), nv, scratch.data_ptr<float>(), scratch_rows,
                                      nf ? faces.data_ptr<int32_t>() : nullptr, nf, stream),
                 "p3d_mc_extract_fused");
        return {vertices, faces};
    }
    if (!ok || exact_mode) {
        const int64_t ev = nv, ef = nf;  // exact sizes are known now
        ok = !id_overflow && run_pass(ev, ef, std::max(8, 2 * slack_q), 4, false);
        if (!ok) {
            // a region numbered more than 2^26 vertices: the one-pass ids are ambiguous -> dense ids by the
            // scan-numbered counting call (include/p3d_mc.h: p3d_mc_read_counts, bit 1; p3d_mc_count_scan)
            if (id_overflow) {
                check_rc(p3d_mc_count_scan(grid, dtype, rx, ry, rz, thresh, nullptr, ws.data_ptr(), stream), "p3d_mc_count_scan");
                check_rc(p3d_mc_read_counts(ws.data_ptr(), &nv, &nf, nullptr, stream), "p3d_mc_read_counts");
            }
            // pathological region imbalance: the gather emitter writes by vertex id and cannot overflow
            vertices = torch::empty({nv, 3}, vopt);
            faces = torch::empty({nf, 3}, fopt);
            check_rc(p3d_mc_emit(grid, dtype, rx, ry, rz, thresh, lower.data(), upper.data(), nullptr, nullptr,
                                 ws.data_ptr(), nv ? vertices.data_ptr<float>() : nullptr, nv,
                                 nf ? faces.data_ptr<int32_t>() : nullptr, nf, nullptr, stream),
                     "p3d_mc_emit");
        }
        if (nv == 0) vertices = torch::empty({0, 3}, vopt);
        if (nf == 0) faces = torch::empty({0, 3}, fopt);
        return {vertices, faces};
    }
    // exact-size results: the first V / F rows of the buffers; a copy when the guess was generous (the first call on a
    // shape).  (Exact allocations -- P3D_MC_MODE=exact -- returned above.)
    auto fit = [&](Tensor& t, int64_t n, int64_t cap) {
        if (n == cap) return;
        t = 2 * n >= cap ? t.narrow(0, 0, n) : t.narrow(0, 0, n).clone();
    };
    fit(vertices, nv, capv);
    fit(faces, nf, capf);
    return {vertices, faces};
}

// Replaces prim3d::save_mesh_as_ply, marching_cubes.cu:307-352: same header text and binary layout
// (per vertex xyz f32 + rgb u8; per face int32 count 3 + 3 int32 indices), written from ONE packed
// host buffer per section instead of a write call per vertex.
void save_mesh_as_ply(const std::string filename, Tensor vertices, Tensor faces, Tensor colors) {
    TORCH_CHECK(vertices.is_contiguous(), "vertices must be contiguous");
    TORCH_CHECK(faces.is_contiguous(), "faces must be contiguous");
    TORCH_CHECK(colors.is_contiguous(), "colors must be contiguous");
    // the reference reads the three buffers with data_ptr<float>() / data_ptr<int32_t>() / data_ptr<uint8_t>()
    // (marching_cubes.cu:334-347), which throws on any other dtype: no silent conversion here either
    TORCH_CHECK(vertices.scalar_type() == torch::kFloat, "expected scalar type Float but found ", vertices.scalar_type());
    TORCH_CHECK(faces.scalar_type() == torch::kInt, "expected scalar type Int but found ", faces.scalar_type());
    TORCH_CHECK(colors.scalar_type() == torch::kByte, "expected scalar type Byte but found ", colors.scalar_type());
    vertices = vertices.to(torch::kCPU);
    faces = faces.to(torch::kCPU);
    colors = colors.to(torch::kCPU);
    const int64_t nv = vertices.size(0), nf = faces.size(0);

    std::ofstream ply(filename, std::ios::out | std::ios::binary);
    ply << "ply\n";
    ply << "format binary_little_endian 1.0\n";
    ply << "element vertex " << nv << std::endl;
    ply << "property float x\n";
    ply << "property float y\n";
    ply << "property float z\n";
    ply << "property uchar red\n";
    ply << "property uchar green\n";
    ply << "property uchar blue\n";
    ply << "element face " << nf << std::endl;
    ply << "property list int int vertex_index\n";
    ply << "end_header\n";

    std::vector<char> vbuf((size_t)nv * 15);
    const float* vp = vertices.data_ptr<float>();
    const uint8_t* cp = colors.data_ptr<uint8_t>();
    for (int64_t i = 0; i < nv; ++i) {
        std::memcpy(&vbuf[(size_t)i * 15], vp + i * 3, 12);
        std::memcpy(&vbuf[(size_t)i * 15 + 12], cp + i * 3, 3);
    }
    ply.write(vbuf.data(), (std::streamsize)vbuf.size());

    std::vector<int32_t> fbuf((size_t)nf * 4);
    const int32_t* fp = faces.data_ptr<int32_t>();
    for (int64_t i = 0; i < nf; ++i) {
        fbuf[(size_t)i * 4] = 3;
        fbuf[(size_t)i * 4 + 1] = fp[i * 3];
        fbuf[(size_t)i * 4 + 2] = fp[i * 3 + 1];
        fbuf[(size_t)i * 4 + 3] = fp[i * 3 + 2];
    }
    ply.write((const char*)fbuf.data(), (std::streamsize)(fbuf.size() * sizeof(int32_t)));
    ply.close();
}

void test() { std::cout << "hello world!" << std::endl; }  // Core/utils.cpp:10-12

// Replaces the non-OptiX RayCaster of the reference (src/prim3d/Utility/ray_cast.cu:340-450): `create_raycaster`
// builds a BVH over the mesh from HOST tensors (CHECK_CPU_INPUT, :346-347) and `invoke` writes depth, unit normal and
// face id of the nearest hit of every ray into caller-allocated DEVICE tensors (CHECK_INPUT, :393-397).  A thin
// adapter over the C ABI include/p3d_rc.h (libp3drc.so); same messages as the reference's macros (Core/common.h:63-71).
struct RayCaster {
    p3d_rc_caster* impl = nullptr;
    int device = -1;
    RayCaster() = default;
    RayCaster(const RayCaster&) = delete;
    RayCaster& operator=(const RayCaster&) = delete;
    ~RayCaster() { p3d_rc_destroy(impl); }

    void invoke(const Tensor& origins, const Tensor& directions, Tensor& depths, Tensor& normals, Tensor& primitives_ids) {
        TORCH_CHECK(origins.is_cuda(), "origins must be a CUDA tensor");
        TORCH_CHECK(origins.is_contiguous(), "origins must be contiguous");
        TORCH_CHECK(directions.is_cuda(), "directions must be a CUDA tensor");
        TORCH_CHECK(directions.is_contiguous(), "directions must be contiguous");
        TORCH_CHECK(depths.is_cuda(), "depths must be a CUDA tensor");
        TORCH_CHECK(depths.is_contiguous(), "depths must be contiguous");
        TORCH_CHECK(normals.is_cuda(), "normals must be a CUDA tensor");
        TORCH_CHECK(normals.is_contiguous(), "normals must be contiguous");
        TORCH_CHECK(primitives_ids.is_cuda(), "primitives_ids must be a CUDA tensor");
        TORCH_CHECK(primitives_ids.is_contiguous(), "primitives_ids must be contiguous");
        // (the reference reads them with data_ptr<float>() / data_ptr<int32_t>(), :409-418: other dtypes throw)
        for (const Tensor* t : std::initializer_list<const Tensor*>{&origins, &directions, &depths, &normals})
            TORCH_CHECK(t->scalar_type() == torch::kFloat, "expected scalar type Float but found ", t->scalar_type());
        TORCH_CHECK(primitives_ids.scalar_type() == torch::kInt, "expected scalar type Int but found ",
                    primitives_ids.scalar_type());
        const int64_t n = origins.size(0);
        TORCH_CHECK(origins.numel() == n * 3 && directions.numel() == n * 3 && depths.numel() >= n &&
                        normals.numel() >= n * 3 && primitives_ids.numel() >= n,
                    "ray tensors must hold num_rays entries (origins / directions / normals x3)");
        TORCH_CHECK(origins.device().index() == device, "the ray caster lives on cuda:", device);
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(origins.device());
        void* stream = (void*)c10::hip::getCurrentHIPStream(origins.device().index()).stream();
        const int rc = p3d_rc_invoke(impl, origins.data_ptr<float>(), directions.data_ptr<float>(), n,
                                     depths.data_ptr<float>(), normals.data_ptr<float>(),
                                     primitives_ids.data_ptr<int32_t>(), stream);
        TORCH_CHECK(rc == P3D_RC_OK, "p3d_rc_invoke failed (", rc, "): ", p3d_rc_last_error());
    }
};

RayCaster* create_raycaster(const Tensor& vertices, const Tensor& faces) {
    TORCH_CHECK(!vertices.is_cuda(), "vertices must be a CPU tensor");
    TORCH_CHECK(vertices.is_contiguous(), "vertices must be contiguous");
    TORCH_CHECK(!faces.is_cuda(), "faces must be a CPU tensor");
    TORCH_CHECK(faces.is_contiguous(), "faces must be contiguous");
    TORCH_CHECK(vertices.scalar_type() == torch::kFloat, "expected scalar type Float but found ", vertices.scalar_type());
    TORCH_CHECK(faces.scalar_type() == torch::kInt, "expected scalar type Int but found ", faces.scalar_type());
    TORCH_CHECK(vertices.dim() == 2 && vertices.size(1) == 3 && faces.dim() == 2 && faces.size(1) == 3,
                "expected vertices [V,3] and faces [F,3]");
    auto* rc = new RayCaster();
    const int code = p3d_rc_create(vertices.data_ptr<float>(), vertices.size(0), faces.data_ptr<int32_t>(), faces.size(0),
                                   &rc->impl);
    if (code != P3D_RC_OK) {
        delete rc;
        TORCH_CHECK(false, "p3d_rc_create failed (", code, "): ", p3d_rc_last_error());
    }
    int64_t nn = 0, nt = 0;
    int32_t depth = 0;
    (void)p3d_rc_stats(rc->impl, &nn, &nt, &depth);
    int dev = 0;
    (void)hipGetDevice(&dev);
    rc->device = dev;
    return rc;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "Primitive3D marching cubes, MI355X-native (HIP/CDNA4) build.";
    m.attr("enable_optix") = false;
    m.def("test", &test);
    py::class_<RayCaster>(m, "RayCaster").def("invoke", &RayCaster::invoke);
    m.def("create_raycaster", &create_raycaster, py::return_value_policy::take_ownership);
    m.def("marching_cubes", &marching_cubes);
    m.def("save_mesh_as_ply", &save_mesh_as_ply);
}
