#!/bin/bash
# Runs GPU steps one after the other, each under its own time limit.  A step that ends in a
# test failure (exit 1) does not stop the next one; a step that was killed, timed out, aborted or
# crashed (124, 134, 137, 139, negative signals) ends the session there.
#   usage: tools/gpu_steps.sh "<seconds>|<log name>|<command>" ...
mkdir -p gpurun_out
status=0
for step in "$@"; do
  limit="${step%%|*}"; rest="${step#*|}"; name="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== $name (limit ${limit}s): $cmd"
  timeout -k 10 "$limit" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name exit $rc"
  tail -n 6 "gpurun_out/$name.log"
  if [ $rc -ne 0 ]; then status=$rc; fi
  case $rc in
    0|1|2|5) ;;
    *) echo "=== stopping: $name ended with $rc"; exit $rc ;;
  esac
done
exit $status
