# chain time vs push grid cap / sweeps (run on the GPU box)
for g in 256 128 64 32; do
  echo "GEO_KPP_GRID=$g"; GEO_KPP_GRID=$g python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['ms_per_step'], j['stages_ms']['kmedoids'], j['parity_selfcheck'])"
done
