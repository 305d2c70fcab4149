#!/bin/bash
# HBM traffic of each kernel from the TCC counters (MI355X_MICROARCH.md "HBM"): separate --pmc passes for
# FETCH_SIZE and WRITE_SIZE (they do not fit one pass).  On gfx950 FETCH_SIZE reports half the bytes of a
# wide coalesced read stream -> doubled below; WRITE_SIZE is exact.  Units: KiB per dispatch.
out=$GRAFT_REPO_ROOT/gpurun_out/traffic
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --no-cpu-baseline --no-secondary --repeats 1 --skip-other-mode --mode ${AGCN_BENCH_MODE:-paper} --steps 20 --warmup 2 > $out/$c.log 2>&1
done
python3 - <<PY
import csv,glob,collections,json
res=collections.defaultdict(dict)
for c in ("FETCH_SIZE","WRITE_SIZE"):
    fs=glob.glob("$out/%s/*/*counter_collection.csv" % c)
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "agcn" in r["Kernel_Name"] and r["Counter_Name"]==c:
            acc[r["Kernel_Name"].split("(")[0].replace("void ","")].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        res[k][c]=sum(v)/len(v)
out={}
for k,v in res.items():
    f=v.get("FETCH_SIZE",0)*1024*2   # gfx950: FETCH_SIZE counts 64 B per 128-B request
    w=v.get("WRITE_SIZE",0)*1024
    out[k]={"fetch_bytes_corrected":int(f),"write_bytes":int(w),"hbm_bytes":int(f+w)}
    print("%-50s fetch %9.0f B (x2 corrected)  write %9.0f B  total %9.0f B" % (k[:50], f, w, f+w))
json.dump(out, open("$out/traffic_${AGCN_BENCH_MODE:-paper}.json","w"), indent=1)
PY
