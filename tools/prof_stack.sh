cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/stackprof -- python3 $GRAFT_REPO_ROOT/tools/bench_shapes.py --only stack --steps 50 > $GRAFT_REPO_ROOT/gpurun_out/stackprof.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/stackprof/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-95s %6s %9.2f us  %5.1f%%" % (r["Name"][:95], r["Calls"], float(r["AverageNs"])/1000, 100*float(r["TotalDurationNs"])/tot))
PY
