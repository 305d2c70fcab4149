# Collects the measurements committed under profiles/ for a round (run on the GPU box through gpurun).
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/final
rm -rf $O $GRAFT_REPO_ROOT/gpurun_out/traffic $GRAFT_REPO_ROOT/gpurun_out/mfma; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 400 python bench.py > $O/bench_paper.json 2>$O/bench_paper.err
timeout 250 python bench.py --mode reference --no-cpu-baseline --no-secondary --repeats 1 --skip-other-mode > $O/bench_reference.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --no-cpu-baseline --no-secondary --repeats 1 --skip-other-mode --steps 100 > $O/kstats.log 2>&1 )
bash tools/traffic.sh > $O/traffic_paper.txt 2>&1
AGCN_BENCH_MODE=reference bash tools/traffic.sh > $O/traffic_reference.txt 2>&1
cp gpurun_out/traffic/traffic_*.json $O/
bash tools/mfma.sh > $O/mfma.txt 2>&1; cp gpurun_out/mfma/mfma.json $O/
for N in 2 4 8; do for s in 0 1; do timeout 100 python bench.py --shard-of $N --shard $s --no-cpu-baseline --no-secondary --repeats 1 --skip-other-mode >> $O/shards.jsonl 2>/dev/null; done; done
for b in 512 1024 2048 4096; do timeout 200 python bench.py --batch $b --no-cpu-baseline --no-secondary --repeats 1 --skip-other-mode >> $O/batch_sweep.jsonl 2>/dev/null; done
AGCN_BENCH_SELFTEST=1 timeout 300 python bench.py --gpus 2 --steps 100 --no-cpu-baseline --no-secondary --repeats 1 --skip-other-mode > $O/bench_2ranks_one_gpu_gloo.json 2>/dev/null
timeout 900 python tools/bench_shapes.py --prof --epoch-graphs 41127 --modelnet-batch 32 > $O/shapes_paper.jsonl 2>/dev/null
timeout 300 python tools/bench_shapes.py --mode reference --only tox21-l1,tox21-l3,reslap,stack > $O/shapes_reference.jsonl 2>/dev/null
AGCN_SPEC=0 timeout 100 python bench.py --no-cpu-baseline --no-secondary --repeats 1 --skip-other-mode > $O/bench_general_fused_kernels.json 2>/dev/null
AGCN_FUSED_MODE=0 timeout 100 python bench.py --no-cpu-baseline --no-secondary --repeats 1 --skip-other-mode > $O/bench_slab_forced.json 2>/dev/null
ls -la $O
