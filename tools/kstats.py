"""print the top rows of a rocprofv3 --kernel-trace --stats output directory"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    print(r["Name"][:72].ljust(72), r["Calls"].rjust(5), ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(10), r["Percentage"].rjust(7))
