"""dev probe: does RCCL on ONE GPU (world_size 1, backend "nccl") execute a batched isend/irecv pair whose peer is the
rank itself, on views of larger device buffers, while a kernel of the caller runs?  (What SlabExtractor.extract() does
between neighbours, minus the second GPU.)  Prints one line per step; run under `timeout`."""
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29791"), HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print("group up:", dist.get_backend(), dist.get_world_size(), flush=True)
big = torch.arange(3 * 1024 * 1024, dtype=torch.float32, device=dev).view(3, 1024, 1024)
dst = torch.zeros(1024, 1024, dtype=torch.float32, device=dev)
hdr = torch.arange(64, dtype=torch.int64, device=dev)
out = torch.empty((1, 3), dtype=torch.int64, device=dev)
dist.all_gather_into_tensor(out.view(-1), hdr[:3])
torch.cuda.synchronize()
print("all_gather_into_tensor of 24 bytes:", out.tolist(), flush=True)
t0 = time.time()
ops = [dist.P2POp(dist.isend, big[1], 0), dist.P2POp(dist.irecv, dst, 0)]
works = dist.batch_isend_irecv(ops)
x = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev)   # a kernel of the caller meanwhile
for w in works:
    w.wait()
torch.cuda.synchronize()
print("loopback isend/irecv of a 4 MiB plane view: equal =", bool(torch.equal(dst, big[1])), f"{time.time() - t0:.3f} s", flush=True)
dist.destroy_process_group()
print("done", flush=True)
