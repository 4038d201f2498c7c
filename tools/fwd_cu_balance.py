"""dev helper: what the per-wave stamps of the forward's timing build (tools/fwd_phase_times.py with UR_DEV_DUMP=<file>)
say about balance: how a workgroup's duration relates to its own work (pairs, layer rows) and to its rank in the
schedule, and how far the compute units' finishing times and work sums are apart.  Workgroup i of the first round sits
on CU i % 256 (tools/micro/placement.hip).  usage: python tools/fwd_cu_balance.py gpurun_out/fwd_tim_c4.pt"""
import sys

import torch

t = torch.load(sys.argv[1]).double()
nw = t.shape[0]
nwg = nw // 4
live = t[:, 3] > 0
t0 = t[:, 0][t[:, 0] > 0].min()
zero = torch.zeros(nw, dtype=torch.double)
end = torch.where(live, (t[:, 3] - t0) / 100.0, zero)     # wall_clock64: 100 MHz
start = torch.where(t[:, 0] > 0, (t[:, 0] - t0) / 100.0, zero)
dur = end - start


def corr(a, b):
    a, b = a - a.mean(), b - b.mean()
    return float((a * b).sum() / (a.norm() * b.norm()))


wg_live = live.view(nwg, 4).any(1)
wg_end = end.view(nwg, 4).max(1).values
wg_start = start.view(nwg, 4).min(1).values
wg_dur = wg_end - wg_start
wg_pairs, wg_rows = t[:, 10].view(nwg, 4).sum(1), t[:, 14].view(nwg, 4).sum(1)
first = wg_live & (torch.arange(nwg) < 1024)
print(f"workgroups {nwg}, with rows {int(wg_live.sum())}, of them in the first 1024 (all resident from t = 0) {int(first.sum())}; "
      f"kernel span {float(wg_end.max()):.1f} us, mean workgroup {float(wg_dur[wg_live].mean()):.1f} us, "
      f"total wave time / 4096 slots {float(dur[live].sum()) / 4096:.1f} us")
qs = torch.tensor([.1, .5, .9, 1.0], dtype=torch.double)
print("first-round workgroup duration p10/p50/p90/max:", [round(v, 1) for v in torch.quantile(wg_dur[first], qs).tolist()])
rank = torch.arange(nwg, dtype=torch.double)
print(f"corr(duration, own pairs) {corr(wg_dur[first], wg_pairs[first]):.2f}, (duration, own layer rows) "
      f"{corr(wg_dur[first], wg_rows[first]):.2f}, (duration, rank in the schedule) {corr(wg_dur[first], rank[first]):.2f}")
cu_end, cu_pairs, cu_rows = torch.zeros(256, dtype=torch.double), torch.zeros(256, dtype=torch.double), torch.zeros(256, dtype=torch.double)
for i in torch.nonzero(first)[:, 0].tolist():
    c = i % 256
    cu_end[c] = max(cu_end[c], wg_end[i])
    cu_pairs[c] += wg_pairs[i]
    cu_rows[c] += wg_rows[i]
print("per CU (first round): last end min/p50/max", [round(float(v), 1) for v in (cu_end.min(), cu_end.median(), cu_end.max())],
      "; pairs min/p50/max", [int(v) for v in (cu_pairs.min(), cu_pairs.median(), cu_pairs.max())],
      "; layer rows min/p50/max", [int(v) for v in (cu_rows.min(), cu_rows.median(), cu_rows.max())])
print(f"corr(CU end, CU pairs) {corr(cu_end, cu_pairs):.2f}, (CU end, CU layer rows) {corr(cu_end, cu_rows):.2f}")
