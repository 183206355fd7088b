"""Per-queue timeline of the LAST decode (or encode) call in a rocprofv3 --kernel-trace csv: kernels longer than `minus` us.
tools/timeline2.py <dir> <first-kernel-prefix> [min_us]"""
import csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
first = sys.argv[2]
minus = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
idx = [i for i, r in enumerate(rows) if nm(r).startswith(first)]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in rows[i0:])
print(f"call length {(end - t0) / 1e6:.3f} ms")
for r in rows[i0:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if (e - s) * 1e3 >= minus:
        print(f"{s:8.3f} {e:8.3f} {e - s:7.3f}  q{r.get('Queue_Id', '?'):>3} {nm(r)[:40]}")
