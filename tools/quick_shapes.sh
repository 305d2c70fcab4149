#!/bin/bash
# GPU quick check: the molecule-layer shapes with the parameter-gradient kernel flavours side by side
for tf in 16 32; do
  echo "== AGCN_PG_TAG_TF=$tf"
  AGCN_PG_TAG_TF=$tf timeout 600 python tools/bench_shapes.py --only tox21-l1,tox21-l2,tox21-l3,tox21-l4,reslap,stack --steps 200 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('%-60s %8.2f us' % (d.get('workload','?')[:60], d.get('us_per_step', 0)))
"
done
echo "== arrival counters"
AGCN_PG_ONEPASS=1 timeout 600 python tools/bench_shapes.py --only tox21-l1,tox21-l2,tox21-l3,tox21-l4,reslap,stack --steps 200 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('%-60s %8.2f us' % (d.get('workload','?')[:60], d.get('us_per_step', 0)))
"
