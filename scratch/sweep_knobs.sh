# per-launch sweep durations for sparse/map threshold settings (GPU box)
cd /tmp && export TMPDIR=/tmp
for cfg in "8 2" "4 2" "2 1"; do
  set -- $cfg
  export GEO_SSSP_SPARSE_DIV=$1 GEO_SSSP_MAP_DIV=$2
  rm -rf /tmp/pk; timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -o s -- python3 $GRAFT_REPO_ROOT/scratch/exp_sweep_only.py > /tmp/pk.log 2>&1
  echo "sparse_div=$1 map_div=$2: $(grep checksum /tmp/pk.log)"; python3 $GRAFT_REPO_ROOT/scratch/sweep_trace.py /tmp/pk 12
done
