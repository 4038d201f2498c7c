"""dev helper: after `bash tools/round_profile.sh <tag>` on the GPU box, copy the summaries the docs cite from
gpurun_out/ into profiles/ (kernel stats of our kernels, PMC means per dispatch, traffic, bench lines).
usage: python tools/collect_profiles.py <tag> <name in profiles/, e.g. r01_v10>"""
import collections, csv, glob, json, shutil, subprocess, sys
tag, name = sys.argv[1], sys.argv[2]
with open(f"profiles/{name}_pmc_summary.txt", "w") as f:
    f.write(subprocess.check_output([sys.executable, "tools/pmc_summary.py", f"gpurun_out/pmc_{tag}_"], text=True))
src = glob.glob(f"gpurun_out/prof_{tag}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(src)))
w = csv.writer(open(f"profiles/{name}_kernel_stats.csv", "w"), quoting=csv.QUOTE_ALL)
for r in [rows[0]] + [r for r in rows[1:] if "k_" in r[0] and "at::" not in r[0]]:
    w.writerow(r)
    print(r[0][:60], r[1], r[3])


def mean(kind, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/pmc_{tag}_{kind}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


fe, wr = mean("fetch", "FETCH_SIZE"), mean("write", "WRITE_SIZE")
t = json.load(open("profiles/r01_traffic.json"))
for kname, key in (("k_raster<2>", "k_raster<2, false>"), ("k_render_bwd", "k_render_bwd(")):
    fk = [v for k, v in fe.items() if key in k][0]
    wk = [v for k, v in wr.items() if key in k][0]
    t[kname]["FETCH_SIZE"], t[kname]["WRITE_SIZE"] = round(fk, 1), round(wk, 1)
    print(kname, "HBM MB per launch", round((2 * fk + wk) * 1024 / 1e6, 1))
t["source"] = t["source"].rsplit("see ", 1)[0] + f"see {name}_pmc_summary.txt"
json.dump(t, open("profiles/r01_traffic.json", "w"), indent=1)
shutil.copy(f"gpurun_out/{tag}_bench.json", f"profiles/{name}_bench.json")
shutil.copy(f"gpurun_out/{tag}_workloads.jsonl", f"profiles/{name}_workloads.jsonl")
