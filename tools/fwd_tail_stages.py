"""dev helper: the stage stamps inside the fused tail's shading rounds, from a dump written by tools/fwd_phase_times.py
(UR_DEV_DUMP=...) run on a build with those stamps (tools/mk_fwd_tail_timing.sh:
slots 6, 7, 9, 12, 13 of a wave's record = marks + keys, face load + eval, texels, composite, row stores).
usage: fwd_tail_stages.py dump.pt"""
import sys, torch
t = torch.load(sys.argv[1]).double()
w = t[t[:, 3] > 0]
CLK = 2100.0
names = {6: "marks + keys", 7: "face load + eval", 9: "texels + write", 12: "composite", 13: "rows store"}
shade = (w[:, 3] - w[:, 2]) / 100.0
q = torch.tensor([.1, .5, .9, 1.0], dtype=torch.double)
print("shade (wall) p10/p50/p90/max", [round(x, 2) for x in shade.quantile(q).tolist()], "mean", round(float(shade.mean()), 2))
rounds = torch.ceil(w[:, 14] / 128.0)
print("items mean", float(w[:, 14].mean()), "rounds mean", float(rounds.mean()))
for k, n in names.items():
    v = w[:, k] / CLK
    print("%-18s p10/p50/p90/max %s mean %.2f ; per round %.2f" % (n, [round(x, 2) for x in v.quantile(q).tolist()], float(v.mean()), float(v.sum() / rounds.sum())))
late = torch.argsort(w[:, 3], descending=True)[:6]
for i in late.tolist():
    print("late wave items %d shade %.1f :" % (int(w[i, 14]), float(shade[i])), {n: round(float(w[i, k] / CLK), 2) for k, n in names.items()})
