#!/bin/bash
# rocprofv3 kernel durations of tools/ubench/rows_split_bench (the program itself right after `--`)
out=$GRAFT_REPO_ROOT/gpurun_out/prof_rsb
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- $GRAFT_REPO_ROOT/tools/ubench/rows_split_bench "$@" > $out/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$out/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("  %-90s n=%4s %8.2f us (min %.2f)" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1000, float(r["MinNs"])/1000))
PY
