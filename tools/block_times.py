"""dev helper: per-workgroup phase timestamps of the fused forward (needs the ur_dev_set_timing hook)."""
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
show("tail (all)", w[:, 5] - w[:, 2])
show("whole wave", w[:, 5] - w[:, 0])
if (w[:, 1] > 0).any():  # tail sub-phases (extra stamps of the dev build: after compaction, short sort, deep sort)
    w14 = tim.view(nblk * 4, 16).cpu()[live][:, 14]
    t_deep_end = (w14 >> 8).double()
    show("tail: hide compaction", w[:, 1] - w[:, 2])
    show("tail: short sort", w[:, 6] - w[:, 1])
    show("tail: deep sort", t_deep_end - w[:, 6])
    show("tail: items + composite", w[:, 5] - t_deep_end)
    nd = (w14 & 255).double()
    print("deep lists per wave: mean %.2f max %d ; deep sort us per list %.2f" % (float(nd.mean()), int(nd.max()), float(us(t_deep_end - w[:, 6]).sum() / max(1.0, float(nd.sum())))))
fine = us(w[:, 2] - w[:, 15])
print("fine phase: batches total %.0f pairs total %.0f (fill %.2f) faces(rects) total %.0f chunks total %.0f" % (float(w[:, 8].sum()), float(w[:, 9].sum()), float(w[:, 9].sum() / (64 * w[:, 8].sum())), float(w[:, 7].sum()), float(w[:, 12].sum())))
print("fine wave-us: total %.0f ; barrier wait+stage write %.0f ; chunk setup (incl. stage load issue) %.0f ; rest (batches) %.0f -> us per batch %.2f" % (float(fine.sum()), float(us(w[:, 10]).sum()), float(us(w[:, 11]).sum()), float((fine - us(w[:, 10]) - us(w[:, 11])).sum()), float((fine - us(w[:, 10]) - us(w[:, 11])).sum() / w[:, 8].sum())))
n_lock = torch.floor(w[:, 12] / 65536.0); d_lock = torch.floor(w[:, 13] / 65536.0)
w[:, 12] = w[:, 12] - n_lock * 65536.0; w[:, 13] = w[:, 13] - d_lock * 65536.0
print("lockstep (lane = pixel) sections: faces total %.0f taking %.0f wave-us -> us per face %.2f" % (float(n_lock.sum()), float(us(d_lock).sum()), float(us(d_lock).sum()) / max(1.0, float(n_lock.sum()))))
hv = torch.argsort(fine, descending=True)[:6]
for i in hv.tolist():
    print("  heavy wave: fine %.1f us, batches %d, pairs %d, rects %d, chunks %d, list_n %d, barrier %.1f, setup %.1f, compactions %d taking %.1f us" % (float(fine[i]), int(w[i, 8]), int(w[i, 9]), int(w[i, 7]), int(w[i, 12]), int(w[i, 13]), float(us(w[i, 10])), float(us(w[i, 11])), int(w[i, 4]), float(us(w[i, 3]))) + " ; lockstep faces %d taking %.1f us" % (int(n_lock[i]), float(us(d_lock[i]))))
print("in-loop compactions: %d taking %.0f wave-us" % (int(w[:, 4].sum()), float(us(w[:, 3]).sum())))
