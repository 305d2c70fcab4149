#!/bin/bash
for i in 1 2 3 4 5; do timeout 200 python tools/bench_shapes.py --only tox21-l3,tox21-l4 --steps 300 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('%-30s %8.2f us' % (d.get('workload','?')[:30], d.get('us_per_step', 0)))
"; done
echo "== counters"; AGCN_PG_ONEPASS=1 timeout 200 python tools/bench_shapes.py --only tox21-l3,tox21-l4 --steps 300 2>/dev/null | cut -c1-120
