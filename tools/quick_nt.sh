# same-box comparison of store-policy variants of the forward (libraries built by hand into tools/build/: spec_forward.hip compiled with
# -DAGCN_NT=<bits> or -DAGCN_WT=<bits> and linked with the other objects of lib/obj/): VARIANTS="nt1 nt3" bash tools/quick_nt.sh
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in hip ${VARIANTS:-nt1 nt3 nt7}; do
  L=tools/build/libagcn_$v.so; [ $v = hip ] && L=adaptive-graph-convolutional-network_amd/lib/libagcn_hip.so
  AGCN_HIP_LIB=$PWD/$L timeout 300 python bench.py --no-cpu-baseline --no-secondary --skip-other-mode 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
done
