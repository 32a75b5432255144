import os, sys, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from primitive3d_amd import capi
from primitive3d_amd.fields import perlin_grid
from bench import soup_hashes
dev = torch.device("cuda", 0)
shapes = [(96, 80, 130), (64, 64, 64), (128, 128, 200), (40, 200, 520)]
grids = [perlin_grid(s, period=24, seed=i, device=dev) for i, s in enumerate(shapes)]
lo = [0.0, 0.0, 0.0]
ref = []
for g in grids:
    v, f = capi.extract_fused(g, 0.0, lo, [float(x) for x in g.shape])
    torch.cuda.synchronize(); ref.append((v.shape[0], f.shape[0], soup_hashes(v, f)[0]))
errors = []
def worker(tid):
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        for it in range(60):
            k = (it * 3 + tid) % len(grids)
            g = grids[k]
            v, f = capi.extract_fused(g, 0.0, lo, [float(x) for x in g.shape])
            if (v.shape[0], f.shape[0]) != ref[k][:2]: errors.append((tid, it, k, v.shape, f.shape)); continue
            if it % 6 == 0:
                st.synchronize()
                if not torch.equal(soup_hashes(v, f)[0], ref[k][2]): errors.append((tid, it, k, "soup"))
    st.synchronize()
ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
[t.start() for t in ths]; [t.join() for t in ths]
print("capi threads:", "ok" if not errors else errors[:5])
