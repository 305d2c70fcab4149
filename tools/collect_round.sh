# Collects the measurements committed under profiles/ for a round (run on the GPU box through gpurun):
#   usage: bash tools/collect_round.sh <tag>      e.g. r03   -> gpurun_out/final/<tag>_*
set -x
T=${1:-rNN}
O=$GRAFT_REPO_ROOT/gpurun_out/final
rm -rf $O $GRAFT_REPO_ROOT/gpurun_out/traffic $GRAFT_REPO_ROOT/gpurun_out/mfma; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 500 python bench.py > $O/${T}_bench_paper.json 2>$O/bench_paper.err
timeout 250 python bench.py --mode reference --no-cpu-baseline --no-secondary --skip-other-mode > $O/${T}_bench_reference.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -- python3 $GRAFT_REPO_ROOT/bench.py --no-graph --no-cpu-baseline --no-secondary --repeats 1 --skip-other-mode --steps 100 > $O/kstats.log 2>&1 )
cp $O/kstats/*/*kernel_stats.csv $O/${T}_kernel_stats.csv
bash tools/traffic.sh > $O/traffic_paper.txt 2>&1
AGCN_BENCH_MODE=reference bash tools/traffic.sh > $O/traffic_reference.txt 2>&1
cp gpurun_out/traffic/traffic_*.json $O/
bash tools/mfma.sh > $O/mfma.txt 2>&1; cp gpurun_out/mfma/mfma.json $O/${T}_mfma_counters.json
bash tools/prof_shapes.sh ${T}_modelnet --only modelnet-3-64,modelnet-64-128,modelnet-reslap --modelnet-batch 32 > $O/prof_modelnet.txt 2>&1
cp gpurun_out/prof_shapes/${T}_modelnet/${T}_modelnet_counters.json gpurun_out/prof_shapes/${T}_modelnet/${T}_modelnet_kernel_stats.csv $O/
# the 128-feature point-cloud layers (Point_AGCN.py's third / fourth convolution): kernel stats + TCC traffic + MFMA busy
bash tools/prof_shapes.sh ${T}_modelnet128 --only modelnet-128-128,modelnet-reslap-128 --modelnet-batch 32 > $O/prof_modelnet128.txt 2>&1
cp gpurun_out/prof_shapes/${T}_modelnet128/${T}_modelnet128_counters.json gpurun_out/prof_shapes/${T}_modelnet128/${T}_modelnet128_kernel_stats.csv $O/
bash tools/sq_shapes.sh ${T}_modelnet --only modelnet-64-128,modelnet-reslap --modelnet-batch 32 > $O/sq_modelnet.txt 2>&1
cp gpurun_out/sq_shapes/${T}_modelnet/${T}_modelnet_sq.json $O/
for N in 2 4 8; do for s in 0 1; do timeout 100 python bench.py --shard-of $N --shard $s --no-cpu-baseline --no-secondary --skip-other-mode >> $O/${T}_shard_emulation.jsonl 2>/dev/null; done; done
for b in 512 1024 2048 4096; do timeout 200 python bench.py --batch $b --no-cpu-baseline --no-secondary --skip-other-mode >> $O/${T}_batch_sweep.jsonl 2>/dev/null; done
AGCN_BENCH_SELFTEST=1 timeout 300 python bench.py --gpus 2 --steps 100 --no-cpu-baseline --skip-other-mode > $O/${T}_bench_2ranks_one_gpu_gloo.json 2>/dev/null
# BASELINE config 4's command (the HIV-shaped epoch, LPT shards, one bucket all-reduce per global batch): one GPU, fp32 / bf16
# features at rest, and two ranks sharing the GPU over gloo (control flow only)
for dt in f32 bf16; do timeout 200 python bench.py --workload hiv-epoch --dtype $dt --steps 160 --warmup 10 >> $O/${T}_hiv_epoch.jsonl 2>/dev/null; done
AGCN_BENCH_SELFTEST=1 timeout 300 python bench.py --gpus 2 --workload hiv-epoch --dtype bf16 --steps 80 --warmup 5 >> $O/${T}_hiv_epoch.jsonl 2>/dev/null
timeout 900 python tools/bench_shapes.py --prof --epoch-graphs 41127 --modelnet-batch 32 > $O/${T}_shapes_paper.jsonl 2>/dev/null
timeout 300 python tools/bench_shapes.py --mode reference --only tox21-l1,tox21-l3,reslap,stack,modelnet-3-64,modelnet-64-128,modelnet-reslap --modelnet-batch 32 > $O/${T}_shapes_reference.jsonl 2>/dev/null
# the reference's whole point-cloud network, eager, under rocprofv3: which kernels its step is made of (torch glue included)
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/netstats -- python3 $GRAFT_REPO_ROOT/tools/prof_network.py 10 > $O/netstats.log 2>&1 )
cp $O/netstats/*/*kernel_stats.csv $O/${T}_network_kernel_stats.csv
# HIV-shaped batches whose largest molecule is within / beyond the members' 128 nodes (wide members), SGC_LL_Reslap and SGC_LL
for m in paper reference; do for v in reslap sgc; do timeout 250 python tools/probe_reslap_reach.py $v $m 2>/dev/null | grep "^{" >> $O/${T}_member_reach.jsonl; done; done
AGCN_CLOUD=0 timeout 200 python tools/bench_shapes.py --only modelnet-64-128 --modelnet-batch 32 --prof > $O/${T}_modelnet_general_slab_kernels.jsonl 2>/dev/null
ls -la $O
