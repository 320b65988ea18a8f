#!/bin/bash
# usage: tools/gpurun_retry.sh TIMEOUT 'command' -- gpurun with retries while the pod's GPU slots are busy (exit code 3: nothing charged)
T=$1; shift
for i in $(seq 30); do
	/usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
	rc=$?
	[ $rc -ne 3 ] && exit $rc
	sleep 90
done
exit 3
