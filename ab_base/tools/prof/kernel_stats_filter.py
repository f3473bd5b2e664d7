"""python tools/prof/kernel_stats_filter.py <rocprofv3 output dir> <substring> [...]: calls and average duration of the kernels
whose name contains one of the substrings (from *kernel_stats.csv of a --kernel-trace --stats run)."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in sys.argv[2:]):
            print("  %-64s %4s %8.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
