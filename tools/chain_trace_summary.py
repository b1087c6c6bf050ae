"""Per-launch view of a rocprofv3 --kernel-trace of tools/exp_chain.py: durations of the kpp_step_kernel launches and the gaps between
them (end of one launch -> start of the next on the stream), grouped by duration class.
usage: chain_trace_summary.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
import numpy as np
rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            if "kpp_step_kernel" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
a = np.array(rows, dtype=np.int64)
# a gap > 5 ms separates repetitions / chain calls
gaps_all = a[1:, 0] - a[:-1, 1]
cut = np.nonzero(gaps_all > 5_000_000)[0]
if len(cut):                                  # the longest run of launches without such a gap
    bounds = np.concatenate([[0], cut + 1, [len(a)]])
    k = int(np.argmax(np.diff(bounds)))
    a = a[bounds[k]:bounds[k + 1]]
dur = (a[:, 1] - a[:, 0]) / 1e3
gap = np.concatenate([[0], (a[1:, 0] - a[:-1, 1]) / 1e3])
print("launches", len(a), "span ms %.2f" % ((a[-1, 1] - a[0, 0]) / 1e6), "sum of durations ms %.2f" % (dur.sum() / 1e3),
      "sum of gaps ms %.2f" % (gap.sum() / 1e3))
edges = [0, 4, 6, 8, 10, 14, 20, 30, 45, 70, 1e9]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (dur >= lo) & (dur < hi)
    if m.any():
        print("duration %5.0f-%-5.0f us: %6d launches, %7.2f ms, mean %.1f us, mean gap before %.1f us" %
              (lo, hi, m.sum(), dur[m].sum() / 1e3, dur[m].mean(), gap[m].mean()))
for q in (0.1, 0.5, 0.9, 0.99):
    print("gap q%.2f %.1f us" % (q, np.quantile(gap, q)))
# by position in the chain: tenths of the launches
for k in range(10):
    s = slice(k * len(a) // 10, (k + 1) * len(a) // 10)
    print("tenth %d: mean duration %.1f us, mean gap %.1f us, launches > 20 us: %d" % (k, dur[s].mean(), gap[s].mean(), (dur[s] > 20).sum()))
