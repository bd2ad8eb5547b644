"""sums one PMC counter per kernel from a rocprofv3 --pmc output directory: pmc_summary.py <dir> <COUNTER> <kernel substring>"""
import csv, glob, sys
d, counter, kname = sys.argv[1], sys.argv[2], sys.argv[3]
tot, n = 0.0, 0
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == counter and kname in r.get("Kernel_Name", ""):
            tot += float(r["Counter_Value"]); n += 1
print(f"{counter} {kname}: dispatches {n} total {tot} per_dispatch {tot / max(1, n)}")
