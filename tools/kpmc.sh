#!/bin/bash
# usage: tools/kpmc.sh <tag> "<counters>"  -- rocprofv3 PMC pass over 20 eager bench steps
tag=$1; ctr=$2
out=$GRAFT_REPO_ROOT/gpurun_out/kpmc/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --no-cpu-baseline --no-secondary --repeats 1 --skip-other-mode --mode ${AGCN_BENCH_MODE:-paper} --steps 20 --warmup 2 > $out/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
fs=glob.glob("$out/*/*counter_collection.csv")
if not fs:
    print(open("$out/log.txt").read()[-2000:])
else:
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "agcn" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:55]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(k)
        print("    " + "  ".join("%s=%.0f" % (c, sum(x)/len(x)) for c,x in sorted(v.items())))
PY
