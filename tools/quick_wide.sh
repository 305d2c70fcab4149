#!/bin/bash
# A/B of the tagged parameter-gradient kernel's number of row splits on the wide molecule layers (AGCN_PG_TAG_SPLITS)
for S in 8 4 5 6; do
  echo "== splits $S"
  AGCN_PG_TAG_SPLITS=$S timeout 300 python tools/bench_shapes.py --only tox21-l1,tox21-l2,tox21-l3,tox21-l4 --prof 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if not l.startswith('{'): continue
    d = json.loads(l)
    print('  %-10s %7.2f us  %s' % (d['workload'], d['us_per_step'], d.get('kernel_us')))
"
done
