# per-launch sweep durations (GPU box)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk; timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -o s -- python3 $GRAFT_REPO_ROOT/scratch/exp_sweep_only.py > /tmp/pk.log 2>&1
grep checksum /tmp/pk.log; python3 $GRAFT_REPO_ROOT/scratch/sweep_trace.py /tmp/pk 8
GEO_SSSP_TRACE=1 python3 $GRAFT_REPO_ROOT/scratch/exp_sweep_only.py 2>&1 | grep sssp | tail -8
