"""dev probe: how much would the fused backward gain if its bins were placed by the exact item counts the forward
leaves behind (layers per pixel)?  Rewrites the schedule in the workspace on the host and times k_render_bwd."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import native, synth
B, H, K, LEVEL = [int(x) for x in os.environ.get('UR_DEV_SHAPE', '32,128,50,3').split(',')]
dev = torch.device("cuda", 0)
sc = synth.make_scene(B, level=LEVEL, tex_size=64, seed=0, device=dev)
s = 1.0 / math.tan(math.radians(15.0))
F = sc["faces"].shape[0]
NF = B * F
fv = native.project_faces(sc["verts"], sc["faces"], sc["R"], sc["T"], s, s)
first = torch.arange(B, device=dev, dtype=torch.int64) * F
num = torch.full((B,), F, device=dev, dtype=torch.int64)
sigma = 1e-4; blur = math.log(1.0 / 1e-4 - 1.0) * sigma
fuv = sc["verts_uvs"][sc["faces"]].contiguous()
bins = (H // 16) ** 2
img, layers, ws = native.render_layered_forward(fv, first, num, (H, H), blur, K, True, sc["maps"], fuv, True, sigma, (1, 1, 1), None, 1)
cot = torch.randn(B, 4, H, H, device=dev)
off = 6 * NF * 16 + ((NF + 3) // 4 * 4) * 4
order = ws[off:off + 4 * B * bins].view(torch.int32)
default = order.clone()
counts = layers[B * bins * 4 * K * 6 * 64:][:B * bins * 256].view(B * bins, 256).sum(1)
print("bins with items", int((counts > 0).sum()), "items", int(counts.sum()))


def timed(tag):
    for _ in range(5):
        native.render_layered_backward(cot, first, num, NF, (H, H), blur, K, True, sc["maps"], fuv, True, sigma, (1, 1, 1), None, 1, layers, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 100
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = native.render_layered_backward(cot, first, num, NF, (H, H), blur, K, True, sc["maps"], fuv, True, sigma, (1, 1, 1), None, 1, layers, ws)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("%-28s %.2f us per backward (fill + k_render_bwd)" % (tag, e0.elapsed_time(e1) * 1000 / n), flush=True)
    return out


ref = timed("default order")
srt = torch.argsort(counts, descending=True, stable=True).to(torch.int32)
order.copy_(srt); timed("sorted by items")
sn = srt.clone()
for row in (1, 3):
    sn[row * 256:(row + 1) * 256] = srt[row * 256:(row + 1) * 256].flip(0)
order.copy_(sn); timed("sorted by items, folded")
# greedy: heaviest first into the column with the smallest sum that still has room (4 per column)
c = counts.cpu().tolist(); idx = srt.cpu().tolist()
sums = [0] * 256; fill = [[] for _ in range(256)]
for i in idx[:1024]:
    col = min((j for j in range(256) if len(fill[j]) < 4), key=lambda j: sums[j])
    fill[col].append(i); sums[col] += c[i]
lpt = [fill[col][row] for row in range(4) for col in range(256)] + idx[1024:]
print("column sums: default-sorted min/max %d/%d ; greedy min/max %d/%d" % (
    min(sum(c[idx[r * 256 + col]] for r in range(4)) for col in range(256)), max(sum(c[idx[r * 256 + col]] for r in range(4)) for col in range(256)), min(sums), max(sums)))
order.copy_(torch.tensor(lpt, dtype=torch.int32, device=dev)); out = timed("greedy by items")
order.copy_(default); timed("default order again")
