#!/bin/bash
# gpurun helper: run the steps of a plan file one after the other; each line is  name|seconds|command .
# Output of a step goes to gpurun_out/<tag>_<name>.log; nothing further runs after a step that hit its time limit.
tag=$1; plan=$2
mkdir -p gpurun_out
while IFS='|' read -r name secs cmd; do
  [ -z "$name" ] && continue
  case "$name" in \#*) continue;; esac
  echo "== $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/${tag}_${name}.log" 2>&1
  rc=$?
  echo "== $name rc=$rc"; tail -6 "gpurun_out/${tag}_${name}.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name hit its limit: stopping"; exit $rc; fi
done < "$plan"
exit 0
