#!/bin/bash
# Regenerates the raw material of profiles/ on the 1-GPU MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash profiles/collect.sh TAG'
# then, back in the container:  python profiles/summarize.py TAG
# Every profiler run is its own process under its own timeout; counters are collected in passes of their own
# (--pmc with --kernel-trace only), as the pool requires.
set -u
TAG=${1:-r01_f}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
# the same command with ONE build after the other: the solo per-launch average of the sweep kernel in a rocprof CSV (roofline.frac)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_solo -o t -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --pipeline 1 --no-cpu-baseline > $OUT/bench_solo_under_rocprof.json 2> $OUT/trace_solo.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 1
for w in c3 swiss c4 real c5cb; do
  timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err || exit 1
done
timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --workload c5prior --steps 100 --warmup 10 > $OUT/bench_c5prior.json 2> $OUT/bench_c5prior.err || exit 1
# the long-geodesics solve alone under the kernel trace (per-launch durations of the push sweeps)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_swiss -o t -- \
    python3 $GRAFT_REPO_ROOT/tools/exp_solve_jvp.py swiss 2 > $OUT/swiss_solve.log 2> $OUT/trace_swiss.err || exit 1
echo collected into $OUT
