for j in 0 1; do
  if [ $j = 1 ]; then export GEO_SSSP_JACOBI=1; fi
  GEO_SSSP_TRACE=1 GEO_SSSP_SPARSE_DIV=1 GEO_SSSP_MAP_DIV=1 timeout -k 10 200 python scratch/exp_two_batches.py 0 32 > gpurun_out/two_j$j.log 2>&1
  echo "jacobi=$j $(grep -E '^sweeps' gpurun_out/two_j$j.log | tr '\n' ' ') improvements per pair: $(grep sssp gpurun_out/two_j$j.log | sed -E 's/.*~([0-9]+) improved.*/\1/' | awk '{s+=$1} END {print s/(60000*16*32)}')"
done
