"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel per dispatch."""
import csv, sys, collections, re
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(path)
    for k, d in acc.items():
        if not k.startswith("k_"):
            continue
        print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
