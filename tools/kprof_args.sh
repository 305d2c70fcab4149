#!/bin/bash
# usage: tools/kprof_args.sh <tag> <bench args...>  -- rocprofv3 kernel stats + trace excerpt of eager bench steps
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/kprof/$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --repeats 1 --skip-other-mode --steps 50 "$@" > $out/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$out/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "agcn" in r["Name"]:
        print("  %-62s %8.2f us (min %.2f) x%s" % (r["Name"][:62], float(r["AverageNs"])/1000, float(r["MinNs"])/1000, r["Calls"]))
f=glob.glob("$out/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "agcn" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
mid=len(rows)*3//4
while mid < len(rows) and "forward" not in rows[mid]["Kernel_Name"] and "<0" not in rows[mid]["Kernel_Name"]: mid+=1
t0=int(rows[mid]["Start_Timestamp"])
for r in rows[mid:mid+14]:
    print("   %8.2f +%7.2f  %s" % ((int(r["Start_Timestamp"])-t0)/1000,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000,r["Kernel_Name"][:50]))
PY
