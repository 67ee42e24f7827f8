#!/bin/bash
# One gpurun session: runs the given steps in order, each under its own timeout, logging to
# gpurun_out/.  An ordinary failure (non-zero exit) is recorded and the session continues; a step
# that is killed by its timeout stops the session (no further GPU work after a hang).
#   tools/gpu_session.sh <name>:<timeout_s>:<command> ...
mkdir -p gpurun_out
summary=gpurun_out/session_summary.txt
: > "$summary"
for spec in "$@"; do
  name="${spec%%:*}"; rest="${spec#*:}"; tmo="${rest%%:*}"; cmd="${rest#*:}"
  echo "=== $name (timeout ${tmo}s): $cmd" | tee -a "$summary"
  start=$(date +%s)
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "    rc=$rc  $(( $(date +%s) - start ))s" | tee -a "$summary"
  tail -n 3 "gpurun_out/$name.log" | sed 's/^/    | /' | tee -a "$summary"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
    echo "    step timed out: stopping the session" | tee -a "$summary"
    exit 1
  fi
done
exit 0
