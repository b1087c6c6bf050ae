"""Turns gpurun_out/profiles_TAG (written by profiles/collect.sh on the GPU box) into the committed summaries:
kernel stats, bench lines, per-kernel PMC means and traffic_latest.json (read by bench.py for roofline.traffic).
usage: python profiles/summarize.py TAG"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_f"
src, dst = os.path.join(ROOT, "gpurun_out", "profiles_" + tag), os.path.join(ROOT, "profiles")


def last_json(path):
    with open(path) as f:
        lines = [l for l in f if l.startswith("{")]
    return json.loads(lines[-1])


shutil.copy(glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True)[0], f"{dst}/{tag}_c2_kernel_stats.csv")
solo = glob.glob(src + "/trace_solo/**/*kernel_stats.csv", recursive=True)
if solo:
    shutil.copy(solo[0], f"{dst}/{tag}_c2_solo_kernel_stats.csv")
for name, out in (("bench_solo_under_rocprof.json", "c2_bench_solo_under_rocprof"), ("bench.json", "c2_bench"), ("bench_under_rocprof.json", "c2_bench_under_rocprof"),
                  ("bench_c3.json", "c3_bench"), ("bench_swiss.json", "swiss_bench"), ("bench_c4.json", "c4_bench"),
                  ("bench_real.json", "real_bench"), ("bench_c5cb.json", "c5cb_bench"), ("bench_c5prior.json", "c5prior_bench")):
    if os.path.exists(os.path.join(src, name)):
        with open(f"{dst}/{tag}_{out}.json", "w") as f:
            json.dump(last_json(os.path.join(src, name)), f, indent=1)
sw = glob.glob(src + "/trace_swiss/**/*kernel_stats.csv", recursive=True)
if sw:
    shutil.copy(sw[0], f"{dst}/{tag}_swiss_solve_kernel_stats.csv")
    shutil.copy(os.path.join(src, "swiss_solve.log"), f"{dst}/{tag}_swiss_solve.log")


def pmc_summary(sub, counter):
    rows = list(csv.DictReader(open(glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True)[0])))
    per = defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    out = f"{dst}/{tag}_c2_pmc_{counter.lower()}_summary.csv"
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", f"{counter}_KiB_mean_per_dispatch", f"{counter}_KiB_total"])
        for k, v in sorted(per.items(), key=lambda kv: -sum(x for _, x in kv[1])):
            tot = sum(x for _, x in v)
            w.writerow([k, len(v), round(tot / len(v), 1), round(tot, 1)])
    return per, out


fetch, f_out = pmc_summary("pmc_fetch", "FETCH_SIZE")
write, w_out = pmc_summary("pmc_write", "WRITE_SIZE")
pick = lambda per, name: sorted(next(v for k, v in per.items() if name in k))
bench = last_json(os.path.join(src, "pmc_fetch.json"))
kernel = bench["roofline"]["kernel"]                             # the sweep kernel this build's bench ran
sw_f, sw_w = pick(fetch, kernel), pick(write, kernel)
fixed_point = kernel == "sweep_chunk32u_kernel"
cal_kernel = "colmin32_kernel" if fixed_point else "colmin_kernel"
cm = pick(fetch, cal_kernel)
g = bench["config"]["graph"]
known = g["nodes"] * 512 * (4.0 if fixed_point else 8.0)          # colmin reads the K x N distance rows once
raw = cm[-1][1] * 1024.0
fetch_raw = sum(x for _, x in sw_f) * 1024.0 / len(sw_f)
wr = sum(x for _, x in sw_w) * 1024.0 / len(sw_w)
traffic = {
    "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes), bench.py --steps 1 "
              f"--warmup 0 --no-cpu-baseline, workload c2, MI355X ({tag}; profiles/collect.sh)",
    "units": "FETCH_SIZE/WRITE_SIZE are reported in KiB; gfx950 FETCH_SIZE counts half of the bytes of coalesced row "
             "reads (MI355X_MICROARCH.md, HBM section)",
    "calibration": {"kernel": cal_kernel, "known_read_bytes": known, "fetch_size_bytes_raw": raw,
                    "ratio_known_over_raw": known / raw},
    "kernel": kernel, "tag": tag, "launches": len(sw_f),
    "fetch_size_bytes_raw_per_launch": fetch_raw, "fetch_bytes_corrected_per_launch": (known / raw) * fetch_raw,
    "write_bytes_per_launch": wr, "hbm_bytes_per_launch": (known / raw) * fetch_raw + wr,
    "per_launch_fetch_KiB": [round(x) for _, x in sw_f],
    "note": "L2-side (fabric) requests: Infinity-Cache hits are included, so this is an upper bound of HBM traffic. "
            f"Algorithmic bytes per launch are {bench['roofline']['algorithmic_bytes_per_launch']:.3e}.",
    "per_kernel_summaries": [os.path.relpath(f_out, ROOT), os.path.relpath(w_out, ROOT)],
}
with open(f"{dst}/traffic_latest.json", "w") as f:
    json.dump(traffic, f, indent=1)
print(json.dumps({k: traffic[k] for k in ("launches", "hbm_bytes_per_launch", "calibration")}, indent=1))
