import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_raster" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "true, true>" in r["Kernel_Name"]))
rows.sort()
# group launches: a gap of more than 20 us between a start and the previous end begins a new forward
groups=[]; cur=[]
for s,e,g in rows:
    if cur and s - max(x[1] for x in cur) > 5000: groups.append(cur); cur=[]
    cur.append((s,e,g))
if cur: groups.append(cur)
groups=groups[len(groups)//2:]
import statistics
span=[(max(x[1] for x in g)-min(x[0] for x in g))/1000 for g in groups]
print(sys.argv[1], "forwards", len(groups), "kernels per forward", statistics.mean(len(g) for g in groups), "raster span us median %.1f min %.1f" % (statistics.median(span), min(span)),
      "; eight-wave kernel us, four-wave kernel us:", [round(statistics.median((x[1]-x[0])/1000 for g in groups for x in g if x[2]==flag),1) for flag in (True, False) if any(x[2]==flag for g in groups for x in g)],
      "; start of the later minus start of the earlier us: %.1f" % statistics.median((max(x[0] for x in g)-min(x[0] for x in g))/1000 for g in groups))
