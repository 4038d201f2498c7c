"""dev helper: per-wave phase stamps of k_raster<2> from the timing build (tools/mk_fwd_timing.sh ->
abx/libf_tim.so; run with UNICORN_RENDER_LIB pointing at it).  UR_DEV_SHAPE=batch,image,K,icosphere level."""
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
WPB = int(os.environ.get("UR_DEV_WPB", "4"))  # waves per bin: 8 when the launch is small enough for eight-wave bins
tim = torch.zeros(nblk * WPB * 16, dtype=torch.int64, device=dev)
L = native.lib()
L.ur_dev_set_timing.argtypes = [ctypes.c_void_p]
L.ur_dev_set_timing.restype = None


def run():
    if os.environ.get("UR_DEV_TWO_STEP"):  # projected faces in: per-face prep kernel, box scan in the rasterizer
        return native.render_layered_forward(fv, first, num, (H, H), blur, K, True, sc["maps"], fuv, True, sigma,
                                             (1, 1, 1), None, 1)
    return native.render_meshes_forward(sc["verts"], sc["faces"], sc["R"], sc["T"], s, s, first, num, (H, H), blur, K,
                                        True, sc["maps"], fuv, True, sigma, (1, 1, 1), None, 1)


for _ in range(3):
    run()
torch.cuda.synchronize()
L.ur_dev_set_timing(ctypes.c_void_p(tim.data_ptr()))
run()
torch.cuda.synchronize()
L.ur_dev_set_timing(None)
t = tim.view(nblk * WPB, 16).cpu().double()
if os.environ.get("UR_DEV_DUMP"):
    torch.save(tim.view(nblk * WPB, 16).cpu(), os.environ["UR_DEV_DUMP"])
q = torch.tensor([.1, .5, .9, .99, 1.0], dtype=torch.double)


def show(name, v):
    print("%-26s p10 %7.2f p50 %7.2f p90 %7.2f p99 %7.2f max %7.2f  mean %7.2f  sum %9.0f" % (
        (name,) + tuple(v.quantile(q).tolist()) + (float(v.mean()), float(v.sum()))))


us = lambda a: a / 100.0  # wall_clock64 ticks at 100 MHz
CLK = float(os.environ.get("UR_DEV_CLK", "2100"))  # clock64 ticks per us (see bwd_phase_times.py)
cu = lambda a: a / CLK
t0 = t[:, 0][t[:, 0] > 0].min()
live = t[:, 3] > 0
w = t[live]
print("waves", t.shape[0], "reaching the end", int(live.sum()), "kernel span us", float(us(w[:, 3].max() - t0)))
show("start", us(w[:, 0] - t0))
show("scan + fine step", us(w[:, 1] - w[:, 0]))
show("  face scan rounds", cu(w[:, 4]))
show("  pair batches", cu(w[:, 5]))
show("  full-list insertions", cu(w[:, 6]))
show("  in-loop compaction", cu(w[:, 7]))
show("  chunk-end barrier", cu(w[:, 8]))
show("tail: sort", us(w[:, 2] - w[:, 1]))
show("tail: shade + blend", us(w[:, 3] - w[:, 2]))
show("whole wave", us(w[:, 3] - w[:, 0]))
show("end time", us(w[:, 3] - t0))
print("pairs %.0f in %.0f batches (fill %.2f), faces with pairs %.0f ; us per batch %.2f ; full-list insertions %.0f at %.2f us ; "
      "compactions %.0f at %.2f us ; items %.0f" % (
          float(w[:, 10].sum()), float(w[:, 11].sum()), float(w[:, 10].sum() / (64 * w[:, 11].sum())), float(w[:, 13].sum()),
          float(cu(w[:, 5]).sum() / w[:, 11].sum()), float(w[:, 9].sum()), float(cu(w[:, 6]).sum() / max(1.0, float(w[:, 9].sum()))),
          float(w[:, 12].sum()), float(cu(w[:, 7]).sum() / max(1.0, float(w[:, 12].sum()))), float(w[:, 14].sum())))
dur = us(w[:, 3] - w[:, 0])
idx = torch.nonzero(live)[:, 0]
for i in torch.argsort(us(w[:, 3] - t0), descending=True)[:8].tolist():
    print("  late wave: wg %5d wave %d start %7.1f end %7.1f ; scan %.1f pairs %.1f (%d batches) full-list %.1f (%d hits) "
          "compaction %.1f barrier %.1f sort %.1f shade %.1f ; items %d" % (
              int(idx[i]) // WPB, int(idx[i]) % WPB, float(us(w[i, 0] - t0)), float(us(w[i, 3] - t0)), float(cu(w[i, 4])),
              float(cu(w[i, 5])), int(w[i, 11]), float(cu(w[i, 6])), int(w[i, 9]), float(cu(w[i, 7])), float(cu(w[i, 8])),
              float(us(w[i, 2] - w[i, 1])), float(us(w[i, 3] - w[i, 2])), int(w[i, 14])))
# how full is the machine over time: waves alive in each tenth of the kernel
span = float(us(w[:, 3].max() - t0))
st, en = us(w[:, 0] - t0), us(w[:, 3] - t0)
print("waves alive at 10%..100% of the span:", [int(((st <= span * k / 10) & (en >= span * k / 10)).sum()) for k in range(1, 11)])
