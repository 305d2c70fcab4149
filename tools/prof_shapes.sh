#!/bin/bash
# rocprofv3 evidence for the secondary workloads of tools/bench_shapes.py (the program itself right after `--`):
#   pass 1  --kernel-trace --stats            per-kernel average duration
#   pass 2  --pmc FETCH_SIZE, pass 3 --pmc WRITE_SIZE   HBM traffic (own passes; FETCH_SIZE x2 on gfx950, MI355X_MICROARCH.md "HBM")
#   pass 4  --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE   matrix-pipe utilisation
# usage: tools/prof_shapes.sh <tag> <bench_shapes args...>     e.g.  r03_modelnet --only modelnet-64-128 --modelnet-batch 32
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_shapes/$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/tools/bench_shapes.py --eager --steps 20 --warmup 3"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B "$@" > $out/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/FETCH_SIZE -- $B "$@" > $out/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/WRITE_SIZE -- $B "$@" > $out/write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -- $B "$@" > $out/mfma.log 2>&1
python3 - <<PY
import csv, glob, collections, json, shutil
out = "$out"
res = collections.defaultdict(dict)
fs = glob.glob(out + "/stats/*/*kernel_stats.csv")
if fs:
    shutil.copy(fs[0], out + "/${tag}_kernel_stats.csv")
    for r in csv.DictReader(open(fs[0])):
        if "agcn" in r["Name"]:
            k = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            res[k]["avg_us"] = round(float(r["AverageNs"]) / 1000, 2)
            res[k]["calls"] = int(r["Calls"])
def pmc(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "agcn" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
for k, v in pmc("FETCH_SIZE").items():
    x = v.get("FETCH_SIZE", [])
    if x: res[k]["fetch_MB_x2_corrected"] = round(sum(x) / len(x) * 1024 * 2 / 1e6, 2)   # counter unit KiB; 64 B tallied per 128-B request
for k, v in pmc("WRITE_SIZE").items():
    x = v.get("WRITE_SIZE", [])
    if x: res[k]["write_MB"] = round(sum(x) / len(x) * 1024 / 1e6, 2)
for k, v in pmc("mfma").items():
    b, g = v.get("SQ_VALU_MFMA_BUSY_CYCLES", []), v.get("GRBM_GUI_ACTIVE", [])
    if b and g:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; each XCD has 32 CUs x 4 SIMDs = 128 matrix pipes
        res[k]["mfma_util_of_chip"] = round((sum(b) / len(b)) / ((sum(g) / len(g)) * 128), 4)
for k in sorted(res):
    print("%-60s %s" % (k[:60], res[k]))
json.dump(res, open(out + "/${tag}_counters.json", "w"), indent=1, sort_keys=True)
PY
grep -h '"workload"' $out/stats.log | head
