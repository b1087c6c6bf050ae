for cap in 128 256 512 1024 2048; do
  echo "grouped cap=$cap: $(GEO_SSSP_GROUPED_CAP=$cap timeout -k 10 120 python scratch/exp_sweep_only.py 60000 swiss 2>/dev/null | tail -1)"
done
