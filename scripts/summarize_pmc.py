"""Per-launch averages of the rocprofv3 --pmc passes collected by collect_profiles.sh (counter_collection.csv files)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            m = re.search(os.environ.get("PMC_KERNEL_RE", r"k_(?:pass|sweep|probe)\w*(?:<[^>]*>)?"), k)
            if not m:
                continue
            k = m.group(0).replace(" ", "")
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
out = {k: {c: round(v[0] / v[1], 1) for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}
json.dump(out, sys.stdout, indent=1)
