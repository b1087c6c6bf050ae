"""Per-kernel medians / sums of the k++ chain kernels from a rocprofv3 kernel trace (last bench step)."""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
st = np.array([int(r['Start_Timestamp']) for r in rows]); en = np.array([int(r['End_Timestamp']) for r in rows])
d = (en - st) / 1e3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for k in ['kpp_push', 'kpp_leaf', 'kpp_sum', 'kpp_tree', 'kpp_scan', 'kpp_pick', 'kpp_draw', 'kpp_commit', 'kpp_finish']:
    sel = [i for i in range(len(names)) if k in names[i]]
    if not sel: continue
    sel = sel[len(sel) * (steps - 1) // steps:]
    print(f"{k:12s} n={len(sel):5d} median={np.median(d[sel]):6.2f} p90={np.percentile(d[sel], 90):6.2f} sum_ms={d[sel].sum() / 1e3:6.2f}")
