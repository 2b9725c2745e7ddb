#!/bin/bash
# gpurun with a bounded wait for a free slot: exit code 3 means "no box or slot free, nothing charged"; any other code is final.
#   tools/gpurun_retry.sh <timeout-seconds> '<command>'
T=$1; shift
for attempt in 1 2 3 4 5 6 7 8; do
    /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    echo "[gpurun_retry] no slot (attempt $attempt), waiting 150 s"
    sleep 150
done
exit 3
