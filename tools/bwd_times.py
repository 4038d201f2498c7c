"""dev helper: per-wave phase timestamps of the fused backward (needs the ur_dev_set_timing hook)."""
import math, sys, os, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import native, synth
B, H, K, LEVEL = [int(x) for x in os.environ.get('UR_DEV_SHAPE', '32,128,50,3').split(',')]  # batch, image, K, icosphere level
dev = torch.device("cuda", 0)
sc = synth.make_scene(B, level=LEVEL, tex_size=64, seed=0, device=dev)
s = 1.0 / math.tan(math.radians(15.0))
F = sc["faces"].shape[0]
fv = native.project_faces(sc["verts"], sc["faces"], sc["R"], sc["T"], s, s)
first = torch.arange(B, device=dev, dtype=torch.int64) * F
num = torch.full((B,), F, device=dev, dtype=torch.int64)
sigma = 1e-4; blur = math.log(1.0 / 1e-4 - 1.0) * sigma
fuv = sc["verts_uvs"][sc["faces"]].contiguous()
nblk = B * (H // 16) ** 2
tim = torch.zeros(nblk * 64, dtype=torch.int64, device=dev)
L = native.lib()
L.ur_dev_set_timing.argtypes = [ctypes.c_void_p]; L.ur_dev_set_timing.restype = None
img, layers, ws = native.render_layered_forward(fv, first, num, (H, H), blur, K, True, sc["maps"], fuv, True, sigma, (1, 1, 1), None, 1)
cot = torch.randn(B, 4, H, H, device=dev)
def run():
    return native.render_layered_backward(cot, first, num, B * F, (H, H), blur, K, True, sc["maps"], fuv, True, sigma, (1, 1, 1), None, 1, layers, ws)
for _ in range(3): run()
torch.cuda.synchronize()
L.ur_dev_set_timing(ctypes.c_void_p(tim.data_ptr()))
run(); torch.cuda.synchronize()
L.ur_dev_set_timing(None)
t = tim.view(nblk * 4, 16).cpu().double()
us = lambda a: a / 100.0
q = torch.tensor([.5, .9, .99, 1.0], dtype=torch.double)
def show(name, v):
    print("%-28s p50 %.1f p90 %.1f p99 %.1f max %.1f  sum %.0f" % ((name,) + tuple(us(v).quantile(q).tolist()) + (float(us(v).sum()),)))
t0 = t[:, 0][t[:, 0] > 0].min()
live = t[:, 4] > 0
w = t[live]
print("waves", t.shape[0], "reaching the flush", int(live.sum()), "kernel span us", float(us(t[:, 1:5].max() - t0)))
show("start", w[:, 0] - t0)
show("depth scan", w[:, 1] - w[:, 0])
show("table init + phase A", w[:, 2] - w[:, 1])
show("phase B (items)", w[:, 3] - w[:, 2])
show("barrier + flush", w[:, 4] - w[:, 3])
show("whole wave", w[:, 4] - w[:, 0])
for lo, hi in ((0, 4), (4, 9), (9, 51)):
    sel = (w[:, 6] > lo) & (w[:, 6] <= hi)
    if sel.any():
        print("  waves with max depth in (%d,%d]: %d ; scan %.1f ; phase A %.1f ; phase B %.1f" % (lo, hi, int(sel.sum()), float(us(w[sel, 1] - w[sel, 0]).mean()), float(us(w[sel, 2] - w[sel, 1]).mean()), float(us(w[sel, 3] - w[sel, 2]).mean())))
if w.shape[1] > 9 and float(w[:, 9].sum()) > 0:  # dev counters of the timing build: scatter fallbacks to global atomics
    items = float(w[:, 9].sum())
    print("items %.0f ; face-table fallbacks %.0f (%.1f%% of items) ; texel-table fallbacks %.0f (%.1f%% of 4 x items)" % (
        items, float(w[:, 7].sum()), 100 * float(w[:, 7].sum()) / items, float(w[:, 8].sum()), 100 * float(w[:, 8].sum()) / (4 * items)))
    nb = torch.ceil(w[:, 9] / 64)
    print("phase B us per batch: %.2f" % (float(us(w[:, 3] - w[:, 2]).sum()) / float(nb.sum())))
