#!/bin/bash
# the ResAGCNResLap network line of bench.py with / without this round's Reslap members and tagged parameter gradients
for cfg in "AGCN_RESLAP_COOP=3" "AGCN_RESLAP_COOP=0" "AGCN_RESLAP_COOP=0 AGCN_PG_ONEPASS=1"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --no-cpu-baseline --skip-other-mode --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for w in d['secondary']['workloads']:
    if 'residual' in w.get('workload','') or 'Reslap 64->64' in w.get('workload',''):
        print('%-100s %s' % (w['workload'][:100], w.get('ms_per_step')))
"
done
