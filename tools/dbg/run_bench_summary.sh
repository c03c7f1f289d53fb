#!/bin/bash
# usage: bash tools/dbg/run_bench_summary.sh <out-name> [bench args]
out=$1; shift
timeout -k 10 600 python bench.py "$@" > gpurun_out/$out.log 2>&1; echo bench rc=$?
tail -1 gpurun_out/$out.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], 'seq/s', d['ms_per_step'], 'ms/step', d['roofline'])
for k,v in d['kernels'].items(): print(k.ljust(40), str(v['avg_us']).rjust(10), 'us', str(v['share']).rjust(8), str(round(v['work_per_launch']/v['avg_us']/1e6,1)).rjust(9), 'T(FLOP|B)/s')
print(d.get('cpu_baseline'))
"
