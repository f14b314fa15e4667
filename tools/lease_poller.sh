#!/bin/bash
# Background poller for a closed GPU pool: every ~8 minutes, when the tree is in a consistent (built + committed) state
# (.tree_ok present), try `gpurun -- bash tools/lease_next.sh`; stop at the first call that is not refused.
#   nohup bash tools/lease_poller.sh > gpurun_out/poller.log 2>&1 &
cd "$(dirname "$0")/.."
n=0
while true; do
  n=$((n+1))
  if [ -f .tree_ok ]; then
    /usr/local/graft/bin/gpurun --timeout ${LEASE_TIMEOUT:-1800} -- 'bash tools/lease_next.sh' > gpurun_out/poll_call.log 2>&1
    rc=$?
    echo "$(date +%H:%M:%S) try $n rc=$rc $(grep -o 'status=[a-z_]*' gpurun_out/poll_call.log | head -1)"
    if ! grep -q 'status=refused' gpurun_out/poll_call.log && [ $rc -ne 3 ]; then
      cp gpurun_out/poll_call.log "gpurun_out/lease_$(date +%H%M).log"
      echo "lease ran (rc=$rc); poller stops"; exit 0
    fi
  else
    echo "$(date +%H:%M:%S) try $n skipped (.tree_ok absent)"
  fi
  sleep ${POLL_SLEEP:-480}
done
