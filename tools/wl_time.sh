#!/bin/bash
# step time of one workload under an environment:   tools/wl_time.sh <workload> <steps> [VAR=value ...]
# prints: the environment, ms per step, launches per step by kernel family, the guard's state
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
W=$1; ST=$2; shift 2
mkdir -p gpurun_out/wl_time
env "$@" timeout 400 python bench.py --workload $W --steps $ST --warmup 2 --no-cpu-baseline --no-alt-mode --no-roofline --no-other-configs 2>gpurun_out/wl_time/err_$W.log | tail -1 | python -c "
import json,sys
t=sys.stdin.read().strip()
try:
    r=json.loads(t); c=r['config']
    print('$W', '$*', 'ms_per_step', r['ms_per_step'], {k:v for k,v in (c.get('products_per_step') or {}).items() if v}, c.get('guard'))
except Exception as e:
    print('$W', '$*', 'FAILED', t[-300:])
"
[ -s gpurun_out/wl_time/err_$W.log ] && grep -v 'amdgpu.ids' gpurun_out/wl_time/err_$W.log | grep -i 'error\|assert\|Traceback' | head -3
