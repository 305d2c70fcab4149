#!/bin/bash
# A/B of the parameter-gradient hand-off (arrival counters / tagged granules, 8 or 16 splits) on the bench's secondary workloads
for cfg in "AGCN_PG_ONEPASS=1" "AGCN_PG_TAG_SPLITS=8" "AGCN_PG_TAG_SPLITS=16" "AGCN_PG_ONEPASS=2"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --no-cpu-baseline --skip-other-mode --repeats 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])
for w in d['secondary']['workloads']:
    if 'HIV' in w.get('workload','') or 'stack' in w.get('workload','') or 'Reslap 64->64' in w.get('workload',''):
        print('%-120s %s' % (w['workload'][:120], w.get('ms_per_step', w.get('ms_per_batch'))))
"
done
