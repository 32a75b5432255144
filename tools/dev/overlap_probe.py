"""dev: does the counting kernel hide behind the streaming kernel when both are resident?  (Decides whether counting
blocks riding in the streaming launch would pay.)  Stream A runs the streaming kernel alone (slab part 3), stream B the
counting kernel alone (slab part 4) on the sign words a previous complete extraction left in the same workspace; timed
alone, back to back on one stream, and concurrently on two streams."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid

N = int(os.environ.get("N", "512"))
g = perlin_grid(N, device="cuda")
if os.environ.get("HALF"): g = g.half()
ws = torch.empty(capi.workspace_bytes(N, N, N), dtype=torch.uint8, device="cuda")
v = torch.empty((N * N * N // 16, 3), device="cuda")
f = torch.empty((N * N * N // 8, 3), dtype=torch.int32, device="cuda")
sc = torch.empty((capi.scratch_rows_for(v.shape[0]), 3), device="cuda")
capi.extract_fused_raw(g, 0.0, [0, 0, 0], [N] * 3, ws, v, f, scratch=sc)
print("counts", capi.read_counts(ws))
torch.cuda.synchronize()

A, B = torch.cuda.Stream(), torch.cuda.Stream()


def stream_part(part):
    s = capi.Slab()
    s.part = part
    s.split_plane = 0
    return s


def run_stream():   # the streaming kernel alone (+ a one-wave header kernel)
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], [N] * 3, ws, v, f, slab=stream_part(3), scratch=sc)


def run_count():    # the counting kernel alone (no vertex copy: no buffers)
    capi.extract_fused_raw(g, 0.0, [0, 0, 0], [N] * 3, ws, None, None, slab=stream_part(4))


def timed(fn_a, fn_b, concurrent):
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    with torch.cuda.stream(A):
        e0.record(A)
        if fn_a: fn_a()
        if not concurrent and fn_b: fn_b()
        ea.record(A)
    if concurrent and fn_b:
        with torch.cuda.stream(B):
            B.wait_event(e0)
            fn_b()
            eb.record(B)
    torch.cuda.synchronize()
    t = e0.elapsed_time(ea)
    if concurrent and fn_b: t = max(t, e0.elapsed_time(eb))
    return t * 1e3


for rep in range(4):
    a = min(timed(run_stream, None, False) for _ in range(5))
    b = min(timed(run_count, None, False) for _ in range(5))
    s = min(timed(run_stream, run_count, False) for _ in range(5))
    c = min(timed(run_stream, run_count, True) for _ in range(5))
    print("stream alone %.1f us   count alone %.1f us   back to back %.1f us   concurrent %.1f us" % (a, b, s, c))
