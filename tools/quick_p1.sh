# same-box A/B of two builds of the library: tools/build/libagcn_base.so (a copy of an earlier build) against the in-tree one
cd $GRAFT_REPO_ROOT
O=gpurun_out/p1; mkdir -p $O
for rep in 1 2 3; do
for v in base new; do
  L=adaptive-graph-convolutional-network_amd/lib/libagcn_hip.so; [ $v = base ] && L=tools/build/libagcn_base.so
  AGCN_HIP_LIB=$PWD/$L timeout 300 python bench.py --no-cpu-baseline --no-secondary --skip-other-mode > $O/bench_$v.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['config']['single_step_launches']['ms_per_step'], d['roofline']['avg_launch_us'])"
done
done
