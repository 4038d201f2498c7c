"""dev helper: per-wave phase stamps of k_render_bwd from the timing build (tools/mk_bwd_timing.sh ->
abx/libv_tim.so; run with UNICORN_RENDER_LIB pointing at it).  UR_DEV_SHAPE=batch,image,K,icosphere level."""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import native, synth  # noqa: E402

B, H, K, LEVEL = [int(x) for x in os.environ.get("UR_DEV_SHAPE", "32,128,50,3").split(",")]
dev = torch.device("cuda", 0)
sc = synth.make_scene(B, level=LEVEL, tex_size=64, seed=0, device=dev)
s = 1.0 / math.tan(math.radians(15.0))
F = sc["faces"].shape[0]
fv = native.project_faces(sc["verts"], sc["faces"], sc["R"], sc["T"], s, s)
first = torch.arange(B, device=dev, dtype=torch.int64) * F
num = torch.full((B,), F, device=dev, dtype=torch.int64)
sigma = 1e-4
blur = math.log(1.0 / 1e-4 - 1.0) * sigma
fuv = sc["verts_uvs"][sc["faces"]].contiguous()
nblk = B * ((H + 15) // 16) ** 2
tim = torch.zeros(nblk * 64, dtype=torch.int64, device=dev)
L = native.lib()
L.ur_dev_set_timing.argtypes = [ctypes.c_void_p]
L.ur_dev_set_timing.restype = None
img, layers, ws = native.render_layered_forward(fv, first, num, (H, H), blur, K, True, sc["maps"], fuv, True, sigma,
                                                (1, 1, 1), None, 1)
cot = torch.randn(B, 4, H, H, device=dev)


def run():
    return native.render_layered_backward(cot, first, num, B * F, (H, H), blur, K, True, sc["maps"], fuv, True, sigma,
                                          (1, 1, 1), None, 1, layers, ws)


for _ in range(3):
    run()
torch.cuda.synchronize()
L.ur_dev_set_timing(ctypes.c_void_p(tim.data_ptr()))
run()
torch.cuda.synchronize()
L.ur_dev_set_timing(None)
t = tim.view(nblk * 4, 16).cpu().double()
q = torch.tensor([.1, .5, .9, .99, 1.0], dtype=torch.double)


def show(name, v):
    print("%-26s p10 %6.2f p50 %6.2f p90 %6.2f p99 %6.2f max %6.2f  mean %6.2f" % ((name,) + tuple(v.quantile(q).tolist()) + (float(v.mean()),)))


us = lambda a: a / 100.0  # wall_clock64 ticks at 100 MHz
t0 = t[:, 0][t[:, 0] > 0].min()
live = t[:, 6] > 0
w = t[live]
print("waves", t.shape[0], "reaching the end", int(live.sum()), "kernel span us", float(us(w[:, 6].max() - t0)))
# core-clock ticks per us, from the two clocks stamped at both ends of each wave
ratio = float(((w[:, 14] - w[:, 13]) / us(w[:, 6] - w[:, 0]).clamp(min=1)).median())
print("clock64 ticks per us: %.1f" % ratio)
show("start", us(w[:, 0] - t0))
show("counts + scan", us(w[:, 1] - w[:, 0]))
show("table init + barrier", us(w[:, 2] - w[:, 1]))
show("phase A", us(w[:, 3] - w[:, 2]))
show("phase B", us(w[:, 4] - w[:, 3]))
show("barrier", us(w[:, 5] - w[:, 4]))
show("flush", us(w[:, 6] - w[:, 5]))
show("whole wave", us(w[:, 6] - w[:, 0]))
nb = w[:, 8].clamp(min=1)
print("items: total %.0f ; per wave mean %.1f max %.0f ; batches per wave mean %.2f max %.0f" % (
    float(w[:, 7].sum()), float(w[:, 7].mean()), float(w[:, 7].max()), float(w[:, 8].mean()), float(w[:, 8].max())))
print("phase B per batch (us): %.2f" % (float(us(w[:, 4] - w[:, 3]).sum()) / float(w[:, 8].sum())))
for name, col in (("loads (row/record/uv)", 9), ("eval_face + dalpha", 10), ("texels + bary path", 11), ("scatter + dist path", 12)):
    print("  per batch %-24s %.2f us" % (name, float(w[:, col].sum()) / ratio / float(w[:, 8].sum())))
for lo, hi in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 8)):
    sel = (w[:, 8] > lo) & (w[:, 8] <= hi)
    if sel.any():
        print("  waves with %d < batches <= %d: %5d ; A %.1f ; B %.1f ; barrier %.1f ; flush %.1f ; whole %.1f" % (
            lo, hi, int(sel.sum()), float(us(w[sel, 3] - w[sel, 2]).mean()), float(us(w[sel, 4] - w[sel, 3]).mean()),
            float(us(w[sel, 5] - w[sel, 4]).mean()), float(us(w[sel, 6] - w[sel, 5]).mean()), float(us(w[sel, 6] - w[sel, 0]).mean())))
fb = w[:, 15].long()
tfb, ffb, ntex = (fb & 0xfffff).sum().item(), ((fb >> 20) & 0xfffff).sum().item(), ((fb >> 40) & 0xfffff).sum().item()
print("texel adds %d ; texel-table fallbacks %d (%.2f%%) ; face-table fallbacks %d (%.2f%% of items)" % (
    ntex, tfb, 100.0 * tfb / max(ntex, 1), ffb, 100.0 * ffb / float(w[:, 7].sum())))
# when do workgroups end?
end = us(w[:, 6] - t0)
show("end time", end)
