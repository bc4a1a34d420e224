# usage: bash tools/dbg/gpurun_retry.sh <timeout> <logfile> <command...>: gpurun, retried while no box / slot is free (exit code 3)
t=$1; log=$2; shift 2
for i in $(seq 1 12); do
  gpurun --timeout $t -- "$@" > $log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
