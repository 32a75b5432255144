import torch, time
torch.zeros(1, device="cuda")
def t(f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): x = f()
    return (time.perf_counter() - t0) / n * 1e6
print("empty 70MB dev   %.1f us" % t(lambda: torch.empty((6000000, 3), device="cuda")))
print("empty 24B dev    %.1f us" % t(lambda: torch.empty((8, 3), dtype=torch.int64, device="cuda")))
print("empty pinned     %.1f us" % t(lambda: torch.empty((8, 3), dtype=torch.int64, pin_memory=True)))
print("Event()          %.1f us" % t(lambda: torch.cuda.Event()))
ev = torch.cuda.Event()
print("ev.record        %.1f us" % t(lambda: ev.record()))
s = torch.cuda.Stream()
print("stream.wait_event %.1f us" % t(lambda: s.wait_event(ev)))
print("current_stream   %.1f us" % t(lambda: torch.cuda.current_stream()))
a = torch.zeros(24, device="cuda"); b = torch.zeros(24, device="cuda")
print("small copy_      %.1f us" % t(lambda: a.copy_(b)))
import contextlib
print("cuda.device ctx  %.1f us" % t(lambda: torch.cuda.device(0).__enter__()))
