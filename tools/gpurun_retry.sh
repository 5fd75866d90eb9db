#!/bin/bash
# Submit one gpurun call; when no GPU slot / box is free (exit code 3: nothing ran, nothing charged) wait and submit again.
# Any other outcome (ran, refused, failed) is final: a command that RAN is never repeated.
#   tools/gpurun_retry.sh <timeout-seconds> '<command>'
t="$1"; shift
for attempt in 1 2 3 4 5 6 7 8 9 10 11 12; do
    /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
    rc=$?
    if [ "$rc" != "3" ]; then exit $rc; fi
    echo "[gpurun_retry] no slot free (attempt $attempt), waiting 90 s" >&2
    sleep 90
done
exit 3
