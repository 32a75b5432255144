import sys, numpy as np, torch
sys.path.insert(0, '.')
from primitive3d_amd import capi
shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 4, 4)
val = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
g = torch.full(shape, val, device='cuda')
v, f, ws = capi.extract_fused(g, 0.0, return_ws=True)
torch.cuda.synchronize()
lay = capi.debug_layout(*shape)
w = ws.cpu().numpy()
U = lay['num_units']
bits = w[lay['off_bits']:lay['off_bits'] + U * 8].view(np.uint64)
print('nv nf', v.shape, f.shape)
print('bits', [hex(int(b)) for b in bits[:32]])
print(v[:8].cpu().numpy())
