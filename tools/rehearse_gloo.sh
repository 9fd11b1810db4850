#!/bin/bash
# Two ranks of bench.py on ONE GPU over gloo (the N > 1 code path without a second GPU), each under a watchdog that
# dumps the Python stacks and exits if the run takes longer than WATCHDOG seconds.   usage: tools/rehearse_gloo.sh [bench args]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29655} WORLD_SIZE=2
W=${WATCHDOG:-90}
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r python -u -X faulthandler -c "
import faulthandler, runpy, sys
faulthandler.dump_traceback_later($W, exit=True)
sys.argv = ['bench.py', '--gpus', '2', '--backend', 'gloo', '--no-cpu-baseline'] + sys.argv[1:]
runpy.run_path('bench.py', run_name='__main__')" "$@" > gpurun_out/rehearse_rank$r.out 2> gpurun_out/rehearse_rank$r.err &
  pids[$r]=$!
done
rc=0
for r in 0 1; do wait ${pids[$r]} || rc=$?; done
grep -h '^{' gpurun_out/rehearse_rank0.out
exit $rc
