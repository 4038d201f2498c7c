"""dev helper: after `bash tools/prof_workload.sh <tag> <workload>` on the GPU box, fold the passes under
gpurun_out/<tag>_<workload>/ into profiles/: <name>_<workload>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, our
kernels), <name>_<workload>_pmc_summary.txt (mean counter value per dispatch) and the per-launch counter means bench.py
reads (profiles/r03_counters.json, stamped with the sha256 of the kernel sources the passes ran on).
usage: python tools/collect_counters.py <tag> <workload> <name in profiles/, e.g. r02_v1>"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unicorn_amd import native  # noqa: E402  (source_hash only: no GPU, no library load)

tag, wl, name = sys.argv[1:4]
src = f"gpurun_out/{tag}_{wl}"
KERNELS = {"k_raster<2": "k_raster<2>", "k_render_bwd": "k_render_bwd", "k_project_prep_mesh": "k_project_prep_mesh",
           "k_bin_order": "k_bin_order", "k_bin_sum": "k_bin_sum", "k_project_faces_bwd_mesh": "k_project_faces_bwd_mesh",
           "k_camera_bwd_mesh": "k_camera_bwd_mesh",
           "k_face_prep": "k_face_prep", "k_project_faces(": "k_project_faces"}


def short(kernel_name):
    for pat, s in KERNELS.items():
        if pat in kernel_name:
            return s
    return None


means = collections.defaultdict(dict)
lines = []
for d in sorted(glob.glob(f"{src}/pmc_*")):
    if not os.path.isdir(d):
        continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                a = acc[(k, r["Counter_Name"])]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        means[k][c] = round(v / n, 1)
        lines.append(f"{k:28s} {c:24s} {v / n:16.1f}  (n={n})")
open(f"profiles/{name}_{wl}_pmc_summary.txt", "w").write("\n".join(lines) + "\n")

stats = glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(stats)))
w = csv.writer(open(f"profiles/{name}_{wl}_kernel_stats.csv", "w"), quoting=csv.QUOTE_ALL)
for r in [rows[0]] + [r for r in rows[1:] if short(r[0])]:
    w.writerow(r)
for r in rows[1:]:
    if short(r[0]):
        means[short(r[0])]["kernel_stats_avg_ns"] = float(r[3])
        print(f"{short(r[0]):28s} calls {r[1]:>5s}  avg {float(r[3]) / 1e3:9.1f} us")

path = "profiles/r03_counters.json"
doc = json.load(open(path)) if os.path.exists(path) else {}
if doc.get("kernel_source_sha256") != native.source_hash():  # counters of other kernels do not mix with these
    doc = {}
doc = doc or {
    "kernel_source_sha256": native.source_hash(),
    "source": "rocprofv3 --pmc (separate passes: two SQ sets, FETCH_SIZE + GRBM_GUI_ACTIVE, WRITE_SIZE) of bench.py "
              "--steps 3 --warmup 2 --workload <w> (tools/prof_workload.sh), mean per dispatch; kernel_stats_avg_ns from "
              "rocprofv3 --kernel-trace --stats of the default bench.py run",
    "units": "SQ_INSTS_* in wave-instructions; SQ_*CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles summed over waves; "
             "FETCH_SIZE / WRITE_SIZE in KB (HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE tallies "
             "half of a wide coalesced read, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE summed over the 8 XCDs",
    "workloads": {}}
doc["workloads"][wl] = {k: v for k, v in means.items()}
doc["workloads"][wl]["_from"] = f"{name}_{wl}_pmc_summary.txt"
json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
for k in ("k_raster<2>", "k_render_bwd"):
    m = means.get(k, {})
    if "SQ_INSTS_VALU" in m and "kernel_stats_avg_ns" in m:
        t = m["kernel_stats_avg_ns"] * 1e-9
        print(f"{k}: VALU issue frac {m['SQ_INSTS_VALU'] / t / (1024 * 2.4e9 / 2.314):.3f}  HBM "
              f"{(2 * m.get('FETCH_SIZE', 0) + m.get('WRITE_SIZE', 0)) * 1024 / 1e6:.1f} MB/launch  LDS conflict ratio "
              f"{m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_ACTIVE_INST_LDS', 1), 1):.2f}")
