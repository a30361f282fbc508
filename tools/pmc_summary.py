"""Summarise a rocprofv3 counter_collection.csv: mean counter value per kernel name."""
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        m = re.search(r"(k_\w+(<[^>]*>)?)", row["Kernel_Name"])
        name = m.group(1) if m else row["Kernel_Name"][:50]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    print(name)
    for c, v in sorted(cs.items()):
        # max = the full-size dispatches when small launches of the same kernel (set-up solves) dilute the mean
        print("   %-32s n=%-4d mean=%.4g max=%.4g" % (c, len(v), sum(v) / len(v), max(v)))
