#!/bin/bash
# SQ wave-state counters of the kernels of a tools/bench_shapes.py workload (own --pmc pass):
#   SQ_WAVE_CYCLES = SQ_WAIT_ANY (parked at s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY (+ idle)
# usage: tools/sq_shapes.sh <tag> <bench_shapes args...>
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/sq_shapes/$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_BUSY_CYCLES --output-format csv -d $out/p -- python3 $GRAFT_REPO_ROOT/tools/bench_shapes.py --eager --steps 10 --warmup 2 "$@" > $out/log.txt 2>&1
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "agcn" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    res[k] = m
    print("%-44s wait %.2f  issue-stall %.2f  active %.2f (valu %.2f) of wave cycles; valu insts %.3g smem %.3g" % (
        k[:44], m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        m.get("SQ_ACTIVE_INST_VALU", 0) / wc, m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_SMEM", 0)))
json.dump(res, open("$out/${tag}_sq.json", "w"), indent=1, sort_keys=True)
PY
