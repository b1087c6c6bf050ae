#!/bin/bash
# gpurun helper: GPU test suite, then a short bench; nothing further runs after a step that hit its time limit.
mkdir -p gpurun_out
tag=${1:-r2}
timeout -k 10 ${2:-900} python -m pytest tests -m gpu -q --durations=12 > gpurun_out/${tag}_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/${tag}_pytest.log
tail -25 gpurun_out/${tag}_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 240 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"
cat gpurun_out/${tag}_bench.json
