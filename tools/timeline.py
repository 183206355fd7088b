"""Timeline of one encode (or decode) call from a rocprofv3 --kernel-trace csv: tools/timeline.py <dir> <first-kernel-prefix> [occurrence]
prints the kernels between that kernel's `occurrence`-th launch and the next one, with start/end relative to the first, per stream/queue."""
import csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
first = sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 2
span = float(sys.argv[4]) if len(sys.argv) > 4 else 12.0
starts = [i for i, r in enumerate(rows) if nm(r).startswith(first)]
# group launches closer than 3 ms into one call
calls = []
for i in starts:
    if not calls or int(rows[i]["Start_Timestamp"]) - int(rows[calls[-1]]["Start_Timestamp"]) > 3e6:
        calls.append(i)
i0 = calls[occ]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if s > span:
        break
    if e - s > 0.03:
        print(f"{s:8.3f} {e:8.3f} {e - s:7.3f}  q{r.get('Queue_Id', '?'):>3} {nm(r)[:40]}")
