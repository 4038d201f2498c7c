"""dev helper: per-workgroup phase timestamps of the fused forward (needs the ur_dev_set_timing hook)."""
import math, sys, os, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import native, synth
B, H, K = 32, 128, 50
dev = torch.device("cuda", 0)
sc = synth.make_scene(B, level=3, tex_size=64, seed=0, device=dev)
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
def run():
    return native.render_layered_forward(fv, first, num, (H, H), blur, K, True, sc["maps"], fuv, True, sigma, (1, 1, 1), None, 1)
for _ in range(3): run()
torch.cuda.synchronize()
L.ur_dev_set_timing(ctypes.c_void_p(tim.data_ptr()))
run(); torch.cuda.synchronize()
L.ur_dev_set_timing(None)
t = tim.view(nblk * 4, 16).cpu().double()
live = t[:, 5] > 0
us = lambda a: a / 100.0
t0 = t[:, 0][t[:, 0] > 0].min()
q = torch.tensor([.5, .9, .99, 1.0], dtype=torch.double)
def show(name, v):
    print("%-28s p50 %.1f p90 %.1f p99 %.1f max %.1f  sum %.0f" % ((name,) + tuple(us(v).quantile(q).tolist()) + (float(us(v).sum()),)))
w = t[live]
print("waves", t.shape[0], "with tail", int(live.sum()), "kernel span us", float(us(w[:, 5].max() - t0)))
show("start", w[:, 0] - t0)
show("scan round 0", w[:, 15] - w[:, 0])
show("fine", w[:, 2] - w[:, 15])
show("tail: compaction", w[:, 3] - w[:, 2])
show("tail: sort", w[:, 4] - w[:, 3])
show("tail: items", w[:, 5] - w[:, 4])
show("whole wave", w[:, 5] - w[:, 0])
items = w[:, 6]; mc = w[:, 7]
nb = torch.ceil(items / 64)
print("item batches: total", float(nb.sum()), "us per batch (items phase sum / batches)", float(us(w[:, 5] - w[:, 4]).sum() / nb.sum()))
print("maxcnt before tail compaction: p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(mc.quantile(q).tolist()))
for lo, hi in ((0, 9), (9, 20), (20, 51)):
    sel = (mc > lo) & (mc <= hi)
    if sel.any():
        print("  waves with maxcnt in (%d,%d]: %d ; compaction us mean %.1f ; sort mean %.1f ; items mean %.1f" % (lo, hi, int(sel.sum()), float(us(w[sel, 3] - w[sel, 2]).mean()), float(us(w[sel, 4] - w[sel, 3]).mean()), float(us(w[sel, 5] - w[sel, 4]).mean())))

long = (w[:, 5] - w[:, 0]) > 2000
mhz = (w[long, 14] - w[long, 1]) / (w[long, 5] - w[long, 0]) * 100.0
print("shader clock seen by waves longer than 20 us: MHz p10 %.0f p50 %.0f p90 %.0f" % tuple(mhz.quantile(torch.tensor([.1, .5, .9], dtype=torch.double)).tolist()))
