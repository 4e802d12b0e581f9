"""Print the top kernels of a rocprofv3 --kernel-trace --stats run (csv output).  usage: prof_top.py DIR [N]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in list(csv.DictReader(open(f)))[:n]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), r["AverageNs"].rjust(12), r["Percentage"].rjust(7))
