#!/bin/bash
# Developer tool: one gpurun call = GPU tests, then whatever commands follow, each under its own timeout; nothing further is
# started once a step has timed out (a hung kernel must be read, not retried).
#   tools/gpu_session.sh "<cmd1>" "<cmd2>" ...     (output of step k -> gpurun_out/session/step_k.log)
mkdir -p gpurun_out/session
k=0
for cmd in "$@"; do
  k=$((k+1))
  echo "== step $k: $cmd"
  timeout -k 10 "${STEP_TIMEOUT:-900}" bash -c "$cmd" > gpurun_out/session/step_$k.log 2>&1
  rc=$?
  tail -n "${STEP_TAIL:-25}" gpurun_out/session/step_$k.log
  echo "== step $k rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $k timed out: stopping"; exit 1; fi
done
exit 0
