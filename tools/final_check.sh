# End-of-round check on the GPU box: the whole GPU suite (parity margins), the stress and fuzz tools, the phase stamps of the final build,
# the same-box A/B against tools/build/libagcn_base.so (the round's first build) and the hand-off microbenchmark -> gpurun_out/finalcheck/
cd $GRAFT_REPO_ROOT
O=gpurun_out/finalcheck; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $O/tests.log; tail -3 $O/tests.log
cp gpurun_out/parity_margins.json $O/parity_margins.json
timeout 900 python tools/fuzz_spec.py 100 23 2>&1 | tail -2 > $O/fuzz.log; tail -1 $O/fuzz.log
timeout 1200 python tools/stress_coop.py 2>&1 | tail -8 > $O/stress_coop.log; tail -1 $O/stress_coop.log
timeout 600 python tools/stress_param_grads.py 2>&1 | tail -4 > $O/stress_pg.log; tail -1 $O/stress_pg.log
AGCN_STAMP_REPS=9 AGCN_STAMP_SEQ=fbfbfbfb timeout 200 python tools/stamps.py > $O/stamps_final.txt 2>&1
timeout 120 tools/ubench/handoff > $O/handoff_latency.txt 2>&1
for rep in 1 2 3; do
for v in base new; do
  L=adaptive-graph-convolutional-network_amd/lib/libagcn_hip.so; [ $v = base ] && L=tools/build/libagcn_base.so
  AGCN_HIP_LIB=$PWD/$L timeout 300 python bench.py --no-cpu-baseline --no-secondary --skip-other-mode > $O/bench_$v.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['config']['single_step_launches']['ms_per_step'], d['roofline']['avg_launch_us'])" >> $O/ab_first_vs_last_build.txt
done
done
cat $O/ab_first_vs_last_build.txt
