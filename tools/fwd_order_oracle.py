"""dev experiment: how much of k_raster<2>'s length is the STATIC placement of bins on compute units?  Needs the
timing + schedule-hook variant (NAME=f_ord bash tools/mk_fwd_timing.sh <hook replacements>): run 1 takes
the library's own schedule and stamps every workgroup; the bins are then re-dealt over the 256 CUs (workgroup i of the
first 1024 sits on CU i % 256) by a cost taken from run 1 -- the workgroup's measured duration, its pairs, or its layer
rows -- longest first onto the CU with the least cost so far, and run 2 takes that schedule.  An oracle: the cost is
known only after rendering.  UR_DEV_SHAPE=batch,image,K,icosphere level."""
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
first = torch.arange(B, device=dev, dtype=torch.int64) * F
num = torch.full((B,), F, device=dev, dtype=torch.int64)
sigma = 1e-4
blur = math.log(1.0 / 1e-4 - 1.0) * sigma
fuv = sc["verts_uvs"][sc["faces"]].contiguous()
nblk = B * ((H + 15) // 16) ** 2
EMPTY = 1 << 30
L = native.lib()
L.ur_dev_set_timing.argtypes = [ctypes.c_void_p]
L.ur_dev_set_timing.restype = None
L.ur_dev_set_order.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
L.ur_dev_set_order.restype = None


def run():
    return native.render_meshes_forward(sc["verts"], sc["faces"], sc["R"], sc["T"], s, s, first, num, (H, H), blur, K,
                                        True, sc["maps"], fuv, True, sigma, (1, 1, 1), None, 1)


def stamped(order_in):
    tim = torch.zeros(nblk * 64, dtype=torch.int64, device=dev)
    out = torch.zeros(nblk, dtype=torch.int32, device=dev)
    L.ur_dev_set_order(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(order_in.data_ptr()) if order_in is not None else None)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    L.ur_dev_set_timing(ctypes.c_void_p(tim.data_ptr()))
    img = run()[0]
    torch.cuda.synchronize()
    L.ur_dev_set_timing(None)
    # untimed-by-stamps kernel time over a few runs (events around the whole forward: includes the small kernels)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    L.ur_dev_set_order(None, None)
    t = tim.view(nblk * 4, 16).cpu().double()
    live = t[:, 3] > 0
    t0 = t[:, 0][t[:, 0] > 0].min()
    end = torch.where(live, (t[:, 3] - t0) / 100.0, torch.zeros(nblk * 4, dtype=torch.double))
    start = torch.where(t[:, 0] > 0, (t[:, 0] - t0) / 100.0, torch.zeros(nblk * 4, dtype=torch.double))
    wg = dict(dur=(end.view(nblk, 4).max(1).values - start.view(nblk, 4).min(1).values), pairs=t[:, 10].view(nblk, 4).sum(1),
              rows=t[:, 14].view(nblk, 4).sum(1), live=live.view(nblk, 4).any(1))
    return img, out.cpu(), wg, float(end.max()), e0.elapsed_time(e1) / 20 * 1e3


def redeal(order, cost, live):
    """order: bin per schedule position (with the empty flag); cost / live per position -> a new schedule"""
    pos = [i for i in range(nblk) if live[i]]
    pos.sort(key=lambda i: -float(cost[i]))
    cu_cost, cu_bins = [0.0] * 256, [[] for _ in range(256)]
    rest = []
    for i in pos:
        free = [c for c in range(256) if len(cu_bins[c]) < 4]
        if not free:
            rest.append(i)
            continue
        c = min(free, key=lambda c: cu_cost[c])
        cu_cost[c] += float(cost[i])
        cu_bins[c].append(i)
    new = []
    for slot in range(4):
        for c in range(256):
            if slot < len(cu_bins[c]):
                new.append(int(order[cu_bins[c][slot]]))
    assert len(new) == min(1024, len(pos)), "a CU was left with a free slot"
    new += [int(order[i]) for i in rest]
    new += [int(order[i]) for i in range(nblk) if not live[i]]   # bins without rows (flagged empty or not): last
    assert sorted(x & (EMPTY - 1) for x in new) == list(range(nblk))
    return torch.tensor(new, dtype=torch.int32, device=dev)


img0, order0, wg0, span0, fwd0 = stamped(None)
print(f"library schedule: k_raster<2> span {span0:.1f} us (timing build), whole forward {fwd0:.1f} us")
def by_cost(order, cost, live):
    """plain longest-first order (what a launch of several rounds wants: every freed slot takes the next-longest bin)"""
    pos = sorted((i for i in range(nblk) if live[i]), key=lambda i: -float(cost[i]))
    new = [int(order[i]) for i in pos] + [int(order[i]) for i in range(nblk) if not live[i]]
    return torch.tensor(new, dtype=torch.int32, device=dev)


LPT = os.environ.get("UR_DEV_ORACLE_MODE", "cu") == "lpt"
for name in ("dur", "pairs", "rows"):
    new = by_cost(order0, wg0[name], wg0["live"]) if LPT else redeal(order0, wg0[name], wg0["live"])
    img1, _, wg1, span1, fwd1 = stamped(new)
    assert torch.equal(img0, img1), "the schedule changed the picture"
    print(f"bins {'in longest-first order of' if LPT else 're-dealt over the CUs by'} run 1's {name:5s}: span {span1:.1f} us, whole forward {fwd1:.1f} us")
