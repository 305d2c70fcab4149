#!/bin/bash
# experiment: bulk outputs as write-through (agent-scope) stores; AGCN_WT bits: 1 res_W / res_L / L_sum, 2 reserve, 4 Y, 8 dX / Gp / dZ
for wt in 8 9 10 8 9 10; do
  echo "== AGCN_WT=$wt"
  if [ "$wt" != "0" ]; then export AGCN_HIP_LIB=$PWD/adaptive-graph-convolutional-network_amd/lib/libagcn_hip_wt$wt.so; else unset AGCN_HIP_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --skip-other-mode --no-secondary --repeats 9 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['config'].get('single_step_launches',{}).get('ms_per_step'))
"
done
