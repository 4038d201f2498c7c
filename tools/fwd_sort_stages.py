"""dev helper: stage stamps inside the fused tail's sort phase (build: tools/mk_fwd_sort_timing.sh; dump:
UR_DEV_DUMP=... tools/fwd_phase_times.py).  usage: fwd_sort_stages.py dump.pt"""
import sys
import torch
t = torch.load(sys.argv[1])
w = t[t[:, 3] > 0]
CLK = 2100.0
q = torch.tensor([.1, .5, .9, 1.0], dtype=torch.double)
sort = (w[:, 2] - w[:, 1]).double() / 100.0
print("sort (wall) p10/p50/p90/max", [round(x, 2) for x in sort.quantile(q).tolist()], "mean", round(float(sort.mean()), 2))
n16, n32, n64 = w[:, 4] & 0xffff, (w[:, 4] >> 16) & 0xffff, (w[:, 4] >> 32) & 0xffff
for name, k in (("pre-sort drop", 6), ("register network", 7), ("bitonic 16", 9), ("bitonic 32", 12), ("bitonic 64", 13)):
    v = w[:, k].double() / CLK
    print("%-18s p10/p50/p90/max %s mean %.2f" % (name, [round(x, 2) for x in v.quantile(q).tolist()], float(v.mean())))
for name, n, k, per in (("16", n16, 9, 4), ("32", n32, 12, 2), ("64", n64, 13, 1)):
    it = torch.ceil(n.double() / per)
    print("lists of up to %s: per wave mean %.2f max %d ; passes mean %.2f ; us per pass %.3f" % (
        name, float(n.double().mean()), int(n.max()), float(it.mean()), float((w[:, k].double() / CLK).sum() / max(1.0, float(it.sum())))))
late = torch.argsort(w[:, 3], descending=True)[:6]
for i in late.tolist():
    print("late wave sort %.1f : drop %.2f net %.2f b16 %.2f (%d lists) b32 %.2f (%d) b64 %.2f (%d)" % (
        float(sort[i]), float(w[i, 6]) / CLK, float(w[i, 7]) / CLK, float(w[i, 9]) / CLK, int(n16[i]), float(w[i, 12]) / CLK, int(n32[i]),
        float(w[i, 13]) / CLK, int(n64[i])))
