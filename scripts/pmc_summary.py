"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel name."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][-48:]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, ctrs in acc.items():
    print(name)
    for c, v in sorted(ctrs.items()):
        v2 = sorted(v)
        print(f"   {c:28s} n={len(v):4d} median={v2[len(v2)//2]:.4g} max={v2[-1]:.4g}")
