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
# the slowest waves, and whether a wave's duration follows its own work or its CU's (workgroup i runs on CU i mod 256)
dur = us(w[:, 4] - w[:, 0])
idx = torch.nonzero(live)[:, 0]
hv = torch.argsort(dur, descending=True)[:12]
for i in hv.tolist():
    print("  slow wave: wg %d wave %d start %.1f A %.1f B %.1f flush %.1f whole %.1f ; depth %d items %d face-fb %d texel-fb %d" % (
        int(idx[i]) // 4, int(idx[i]) % 4, float(us(w[i, 0] - t0)), float(us(w[i, 2] - w[i, 1])), float(us(w[i, 3] - w[i, 2])),
        float(us(w[i, 4] - w[i, 3])), float(dur[i]), int(w[i, 6]), int(w[i, 9]) if w.shape[1] > 9 else -1,
        int(w[i, 7]) if w.shape[1] > 9 else -1, int(w[i, 8]) if w.shape[1] > 9 else -1))
if w.shape[1] > 9:
    def corr(a, b):
        a = a - a.mean(); b = b - b.mean()
        return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    print("corr(wave items, phase B) %.3f ; corr(wave items, whole) %.3f ; corr(depth, phase A) %.3f" % (
        corr(w[:, 9], us(w[:, 3] - w[:, 2])), corr(w[:, 9], dur), corr(w[:, 6], us(w[:, 2] - w[:, 1]))))
    # per workgroup: end time and items; per CU (first 1024 workgroups, all resident from the start)
    wg = (idx // 4)
    nwg = nblk
    end_wg = torch.zeros(nwg, dtype=torch.double); items_wg = torch.zeros(nwg, dtype=torch.double)
    end_wg.scatter_reduce_(0, wg, us(w[:, 4] - t0), reduce="amax"); items_wg.scatter_add_(0, wg, w[:, 9])
    first = torch.arange(min(1024, nwg))
    cu = first % 256
    cu_end = torch.zeros(256, dtype=torch.double).scatter_reduce_(0, cu, end_wg[first], reduce="amax")
    cu_items = torch.zeros(256, dtype=torch.double).scatter_add_(0, cu, items_wg[first])
    print("per workgroup: items min %.0f mean %.0f max %.0f ; corr(items, end) %.3f" % (float(items_wg[first].min()), float(items_wg[first].mean()), float(items_wg[first].max()), corr(items_wg[first], end_wg[first])))
    print("per CU: end min %.1f mean %.1f max %.1f ; items min %.0f mean %.0f max %.0f ; corr(CU items, CU end) %.3f" % (
        float(cu_end.min()), float(cu_end.mean()), float(cu_end.max()), float(cu_items.min()), float(cu_items.mean()), float(cu_items.max()), corr(cu_items, cu_end)))
if w.shape[1] > 10 and float(w[:, 10].sum()) > 0:  # physical placement (HW_ID: cu 11:8, sh 12, se 15:13; XCC_ID)
    raw = tim.view(nblk * 4, 16).cpu()[live][:, 10]
    hw = raw & 0xffffffff; xcc = raw >> 32
    phys = (xcc * 8 + ((hw >> 13) & 7)) * 32 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15)
    uniq, inv = torch.unique(phys, return_inverse=True)
    n_cu = uniq.numel()
    end = torch.zeros(n_cu, dtype=torch.double).scatter_reduce_(0, inv, us(w[:, 4] - t0), reduce="amax")
    itm = torch.zeros(n_cu, dtype=torch.double).scatter_add_(0, inv, w[:, 9])
    nwv = torch.zeros(n_cu, dtype=torch.double).scatter_add_(0, inv, torch.ones_like(dur))
    print("physical CUs seen %d ; waves per CU min %d max %d ; items per CU min %.0f mean %.0f max %.0f ; end min %.1f mean %.1f max %.1f ; corr(items, end) %.3f ; corr(waves, end) %.3f" % (
        n_cu, int(nwv.min()), int(nwv.max()), float(itm.min()), float(itm.mean()), float(itm.max()), float(end.min()), float(end.mean()), float(end.max()), corr(itm, end), corr(nwv, end)))
    same = 0
    for c in range(256):
        sel = (idx // 4) % 256 == c
        sel &= (idx // 4) < 1024
        if sel.any() and torch.unique(phys[sel]).numel() == 1: same += 1
    print("launch columns (wg mod 256, first 1024 wgs) that sit on ONE physical CU: %d of 256" % same)
    top = torch.argsort(end, descending=True)[:8]
    for c in top.tolist():
        print("  late CU: xcc %d se %d sh %d cu %d : end %.1f items %.0f waves %d ; wgs %s" % (int(uniq[c]) // 256, (int(uniq[c]) // 32) % 8, (int(uniq[c]) // 16) % 2, int(uniq[c]) % 16, float(end[c]), float(itm[c]), int(nwv[c]), sorted(set((idx[inv == c] // 4).tolist()))[:8]))
