"""Print a window of the kernel timeline (start / end in us, relative) from a rocprofv3
--kernel-trace CSV: shows which kernels of a pipelined context really overlap."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "ssimu2" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:lo + n]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ssimu2::", "")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{name:22s} q={r.get('Queue_Id','?'):>3s} start {s:9.1f} end {e:9.1f} dur {e - s:7.1f}")
