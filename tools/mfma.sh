#!/bin/bash
# MFMA-pipe counters of each kernel (own --pmc pass, nothing but --kernel-trace with it):
#   SQ_VALU_MFMA_BUSY_CYCLES  cycles the matrix pipe is busy, summed over SIMDs (32 per v_mfma_f32_16x16x4_f32)
#   SQ_INSTS_VALU_MFMA_MOPS_F32  f32 MFMA work issued (units of 512 FLOP)
#   SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE  kernel duration in SQ / GPU cycles
# usage: tools/mfma.sh [bench args]
out=$GRAFT_REPO_ROOT/gpurun_out/mfma
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/p -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --no-cpu-baseline --no-secondary --repeats 1 --skip-other-mode --steps 20 --warmup 2 "$@" > $out/log.txt 2>&1
python3 - <<PY
import csv,glob,collections,json
fs=glob.glob("$out/p/*/*counter_collection.csv")
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    if "agcn" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0].replace("void ","")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res={}
for k,v in acc.items():
    m={c:sum(x)/len(x) for c,x in v.items()}
    gui=m.get("GRBM_GUI_ACTIVE",0)
    busy=m.get("SQ_VALU_MFMA_BUSY_CYCLES",0)
    # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs; each XCD has 32 CUs x 4 SIMDs = 128 matrix pipes
    m["mfma_util_of_chip"]=busy/(gui*128) if gui else None
    res[k]=m
    print("%-46s mfma_busy %12.0f  mops_f32 %10.0f  gui %8.0f  sq_busy %10.0f waves %7.0f  util %.4f" % (k[:46], busy, m.get("SQ_INSTS_VALU_MFMA_MOPS_F32",0), gui, m.get("SQ_BUSY_CYCLES",0), m.get("SQ_WAVES",0), m["mfma_util_of_chip"] or 0))
json.dump(res, open("$out/mfma.json","w"), indent=1)
PY
tail -2 $out/log.txt
