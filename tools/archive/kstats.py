"""Print the kernels matching any of the given substrings from a rocprofv3 kernel_stats.csv under a directory.
usage: kstats.py <dir> <substr> [substr ...]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in sys.argv[2:]):
        print("  %-64s calls %4s avg %7.1f us  min %6.1f" % (n[:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
