#!/bin/bash
# usage: tools/kprof.sh <tag> [ENV=VAL ...]   -- rocprofv3 kernel stats of 100 eager bench steps
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/kprof/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --no-cpu-baseline --no-secondary --repeats 1 --skip-other-mode --mode ${AGCN_BENCH_MODE:-paper} --steps 100 > $out/log.txt 2>&1
echo "== $tag $@"
python3 - <<PY
import csv,glob
f=glob.glob("$out/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "agcn" in r["Name"]:
        print("  %-62s %8.2f us (min %.2f)" % (r["Name"][:62], float(r["AverageNs"])/1000, float(r["MinNs"])/1000))
PY
