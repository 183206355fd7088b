"""Print the top kernels of a rocprofv3 --kernel-trace --stats output directory: tools/stats_top.py gpurun_out/<dir> [N]"""
import csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in list(csv.DictReader(open(f)))[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    print(name.ljust(40), r["Calls"].rjust(5), f'{float(r["AverageNs"]) / 1e3:10.1f} us', f'{float(r["TotalDurationNs"]) / 1e6:9.2f} ms', r["Percentage"].rjust(7))
