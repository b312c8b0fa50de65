#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python - <<'PY' > gpurun_out/cpuinfo.log 2>&1
import os
print('cpu_count', os.cpu_count())
print('affinity', len(os.sched_getaffinity(0)))
for p in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, 'n/a')
os.system('lscpu | head -20')
os.system('nproc')
PY
timeout -k 10 600 python tools/phase_cycles.py 2>&1 | tee gpurun_out/phase_cycles.log
