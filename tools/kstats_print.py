"""Print a rocprofv3 kernel_stats.csv compactly: python tools/kstats_print.py <csv> [name filter ...]"""
import csv, sys
flt = sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("geomae::", "").replace("void ", "")
    if not flt or any(f in n for f in flt):
        print("%-38s calls %5s  avg %8.1f us  min %8.1f  max %8.1f  %5.2f %%" % (n[:38], r["Calls"], float(r["AverageNs"]) / 1e3,
              float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
