"""dev: the one-pass call in scratch mode against the predicted region layout (p3d_mc_slab.region_first_rows) on the 512^3
bench grid: wall time per call of a back-to-back stream (raw C ABI + read_counts_ex, the adapter's pattern) and stage events.
GRIDS=4: four distinct grids in turn, each laid out from its predecessor's region totals."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
n = int(os.environ.get("N", "512"))
shape = tuple(int(v) for v in os.environ["SHAPE"].split(",")) if os.environ.get("SHAPE") else (n, n, n)
ngr = int(os.environ.get("GRIDS", "1"))
octaves = int(os.environ.get("OCT", "1"))
grids = [perlin_grid(shape, period=64, seed=s, device="cuda", octaves=octaves, persistence=0.5) for s in range(ngr)]
up = [float(s) for s in shape]
ws = torch.empty(capi.workspace_bytes(*shape), dtype=torch.uint8, device="cuda")
capv0 = shape[0] * shape[1] * shape[2] // 16
v0 = torch.empty((capv0, 3), device="cuda"); f = torch.empty((2 * capv0, 3), dtype=torch.int32, device="cuda")
scratch = torch.empty((capi.scratch_rows_for(capv0), 3), device="cuda")
import random
rng = random.Random(1)
JIT = float(os.environ.get("JIT", "0"))
state = {"regions": None, "i": 0, "over": 0}
def call_scratch():
    g = grids[state["i"] % ngr]; state["i"] += 1
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], up, ws, v0, f, scratch=scratch)
    nv, nf, fl, reg = capi.read_counts_ex(ws)
    state["regions"] = reg
    return nv, nf, fl
def call_layout():
    g = grids[state["i"] % ngr]; state["i"] += 1
    reg = state["regions"]
    if JIT:   # a prediction that is off by up to JIT per region (a slowly changing field)
        reg = [max(0, int(r * (1 + JIT * (2 * rng.random() - 1)))) for r in reg]
    first, rows = capi.region_layout(reg)
    v = torch.empty((rows, 3), device="cuda")
    if os.environ.get("ALT") == "1":   # the caller still holds the last mesh: the allocator hands out two blocks in turn
        state["keep"] = v
    slab = capi.Slab(); slab.region_first_rows = ctypes.cast(first, ctypes.c_void_p)
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], up, ws, v, f, slab=slab)
    nv, nf, fl, reg = capi.read_counts_ex(ws)
    state["regions"] = reg
    state["over"] += 1 if fl & 4 else 0
    return nv, nf, fl
def measure(call, label):
    for _ in range(6): r = call()
    torch.cuda.synchronize()
    walls = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(32): r = call()
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) / 32 * 1e6)
    capi.profile_enable(2)
    acc = {}
    for i in range(8):
        call(); torch.cuda.synchronize()
        st = capi.profile_read()
        if i >= 3:
            for k, t in st.items(): acc[k] = acc.get(k, 0) + t / 5
    capi.profile_enable(0)
    st = {k: round(t * 1e3, 1) for k, t in acc.items()}
    walls.sort()
    print(label, "V", r[0], "F", r[1], "flags", r[2], "call_us median %.1f min %.1f" % (walls[2], walls[0]), st, "layout overflows", state["over"])
for rep in range(int(os.environ.get("REPS", "2"))):
    measure(call_scratch, "scratch")
    for j in [float(x) for x in os.environ.get("JITS", str(JIT)).split(",")]:
        JIT = j
        measure(call_layout, "layout jit %.3f" % j)
