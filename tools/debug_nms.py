import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from ampis_amd import ops, _lib
from ampis_amd._lib import lib, check, ptr
from oracle import maskrcnn as O
DEV="cuda:0"
ctx = ops.torch_context(0)
n, ncat, thresh = 64, 1, 0.3
rng = np.random.default_rng(n + 7)
B, cap = 1, 128
boxes = np.zeros((B, cap, 4), np.float32); cats = np.full((B, cap), -1, np.int32)
c = rng.uniform(0, 300, (n, 2)); s = rng.uniform(4, 80, (n, 2))
boxes[0, :n] = np.concatenate([c - s / 2, c + s / 2], 1); cats[0, :n] = 0
W = (cap + 63) // 64
mask = torch.zeros(B * cap * W, dtype=torch.int64, device=DEV)
keep = torch.full((B, 1000), -1, dtype=torch.int32, device=DEV); kc = torch.zeros(B, dtype=torch.int32, device=DEV)
db, dc, dn = torch.from_numpy(boxes).to(DEV), torch.from_numpy(cats).to(DEV), torch.tensor([n], dtype=torch.int32, device=DEV)
check(lib().amp_nms(ctx.handle, B, cap, ptr(db), ptr(dc), ptr(dn), thresh, 1000, ptr(mask), ptr(keep), ptr(kc)))
torch.cuda.synchronize()
M = mask.cpu().numpy().view(np.uint64).reshape(cap, W)
ref = O.nms_sorted(torch.from_numpy(boxes[0, :n]), torch.from_numpy(cats[0, :n].astype(np.int64)), thresh).numpy()
print("gpu kept", int(kc[0]), keep[0, :int(kc[0])].cpu().numpy())
print("ref kept", len(ref), ref)
# reference mask
b = boxes[0, :n]; area = (b[:,2]-b[:,0])*(b[:,3]-b[:,1])
bad = 0
for i in range(n):
    bits = 0
    for j in range(i+1, n):
        w = max(min(b[i,2], b[j,2]) - max(b[i,0], b[j,0]), np.float32(0)); h = max(min(b[i,3], b[j,3]) - max(b[i,1], b[j,1]), np.float32(0))
        inter = np.float32(w*h); iou = inter / (area[i] + area[j] - inter)
        if iou > np.float32(thresh): bits |= (1 << j)
    if bits != int(M[i, 0]):
        bad += 1
        if bad < 5: print("row", i, "ref", hex(bits), "gpu", hex(int(M[i,0])))
print("bad mask rows", bad)
# greedy on GPU mask done on host
alive = (1 << n) - 1; kept = []
for i in range(n):
    if (alive >> i) & 1:
        kept.append(i); alive &= ~int(M[i, 0])
print("host greedy on gpu mask", len(kept), kept)
