"""Summarise rocprofv3 --pmc csv output: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections
for d in sorted(glob.glob(sys.argv[1] + "*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:60], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        for (kn, cn), (v, n) in sorted(acc.items()):
            if "k_r" in kn or "k_face" in kn or "k_proj" in kn or "k_bin" in kn:
                print(f"{kn:60s} {cn:24s} {v/n:16.1f}  (n={n})")
