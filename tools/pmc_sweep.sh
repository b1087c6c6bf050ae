# PMC passes over the sweep-only script (GPU box); one counter group per run, each under its own timeout.
# (TA_* counters abort the profiler on this pool -- left out.)
cd /tmp && export TMPDIR=/tmp
i=2
for grp in "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sweep/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/exp_sweep_only.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_sweep_$i.log 2>&1 || { echo "pass $i failed"; exit 1; }
  echo "pass $i done"
done
