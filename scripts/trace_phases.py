import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_march(" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("n", len(d))
import statistics
for a in range(0, len(d), 50):
    seg = d[a:a+50]
    print(a, f"mean {statistics.mean(seg):.1f} min {min(seg):.1f} max {max(seg):.1f} med {statistics.median(seg):.1f}")
