#!/bin/bash
# gpurun helper: correctness of the shortest-path / chain kernels, then chain call log, sweep trace, short bench.
mkdir -p gpurun_out; tag=${1:-probe}
timeout -k 10 400 python -m pytest tests/test_gpu_sssp.py tests/test_gpu_reference_scenarios.py -q -x > gpurun_out/${tag}_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/${tag}_pytest.log; tail -5 gpurun_out/${tag}_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 120 python scratch/exp_chain.py gauss 60000 512 > gpurun_out/${tag}_chain_res.log 2>&1; tail -12 gpurun_out/${tag}_chain_res.log
GEO_KPP_RESIDENT=0 timeout -k 10 120 python scratch/exp_chain.py gauss 60000 512 > gpurun_out/${tag}_chain_step.log 2>&1; tail -6 gpurun_out/${tag}_chain_step.log
GEO_SSSP_TRACE=1 timeout -k 10 120 python scratch/exp_sweep_only.py 60000 gauss > gpurun_out/${tag}_sweep.log 2>&1; tail -12 gpurun_out/${tag}_sweep.log
timeout -k 10 120 python scratch/exp_sweep_only.py 60000 gauss 2>&1 | tail -1
timeout -k 10 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['launches_per_step'], d['stages_ms'])"
