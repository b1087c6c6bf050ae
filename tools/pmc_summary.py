"""Per-kernel averages of the counters rocprofv3 --pmc wrote (csv) under <dir>/p*/; kernels whose name contains <substr>."""
import csv, glob, re, sys, collections
root, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if sub not in k:
                continue
            m = re.search(r"(\w*" + re.escape(sub) + r"\w*(<[^>]*>)?)", k)
            key = (m.group(1) if m else k[:60], row["Counter_Name"])
            acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:62s} {c:32s} {v / n:16.1f}  (n={n})")
