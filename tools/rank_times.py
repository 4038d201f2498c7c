import math, sys, os, ctypes
import torch
sys.path.insert(0, os.getcwd())
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
t = tim.view(nblk, 4, 16).cpu().double() / 100.0
t0 = t[:, :, 0][t[:, :, 0] > 0].min()
dur = (t[:, :, 5] - t[:, :, 0]).max(1).values      # per workgroup: longest wave
start = (t[:, :, 0] - t0).min(1).values
end = (t[:, :, 5] - t0).max(1).values
batches = t[:, :, 8].sum(1) * 100; listn = t[:, 0, 13] * 100
fine = (t[:, :, 2] - t[:, :, 15]).max(1).values; tail = (t[:, :, 5] - t[:, :, 2]).max(1).values
print("rank-range : start  dur(mean/min/max)  fine  tail  end(max)  list_n  batches")
for lo in list(range(0, 1024, 128)) + list(range(1024, 2048, 256)):
    hi = lo + (128 if lo < 1024 else 256)
    sl = slice(lo, hi)
    print("%4d-%4d : %5.1f  %5.1f / %5.1f / %5.1f  %5.1f %5.1f  %5.1f  %5.0f  %5.0f" % (lo, hi, start[sl].mean(), dur[sl].mean(), dur[sl].min(), dur[sl].max(), fine[sl].mean(), tail[sl].mean(), end[sl].max(), listn[sl].mean(), batches[sl].mean()))
# how well does the schedule's rank predict the duration?  spearman-ish: correlation of rank with duration among the first 1024
r = torch.arange(1024, dtype=torch.double)
d = dur[:1024]
print("corr(rank, duration) first 1024: %.3f" % float(torch.corrcoef(torch.stack([r, d]))[0, 1]))
print("corr(batches, duration): %.3f ; corr(list_n, duration): %.3f" % (float(torch.corrcoef(torch.stack([batches[:1024], d]))[0, 1]), float(torch.corrcoef(torch.stack([listn[:1024], d]))[0, 1])))
top = torch.argsort(dur, descending=True)[:12]
print("longest workgroups: rank, dur, start, list_n, batches:", [(int(i), round(float(dur[i]), 1), round(float(start[i]), 1), int(listn[i]), int(batches[i])) for i in top])
# CU-level: at t = 0 workgroups c, c+256, c+512, c+768 share a CU (tools/micro/placement.hip)
cu_end = torch.stack([end[c::256][:4].max() for c in range(256)])
cu_b = torch.stack([batches[c::256][:4].sum() for c in range(256)])
items = t[:, :, 6].sum(1) * 100 if False else None
cu_l = torch.stack([listn[c::256][:4].sum() for c in range(256)])
cu_tail = torch.stack([tail[c::256][:4].sum() for c in range(256)])
cu_fine = torch.stack([fine[c::256][:4].sum() for c in range(256)])
print("per-CU: end time of its first four workgroups: min %.1f mean %.1f max %.1f" % (float(cu_end.min()), float(cu_end.mean()), float(cu_end.max())))
for name, v in (("sum batches", cu_b), ("sum list_n", cu_l)):
    print("corr(CU %s, CU end) = %.3f ; spread of %s: min %.0f mean %.0f max %.0f" % (name, float(torch.corrcoef(torch.stack([v, cu_end]))[0, 1]), name, float(v.min()), float(v.mean()), float(v.max())))
