# same-box A/B of two builds of the library through AGCN_HIP_LIB: tools/build/libagcn_base.so (a copy of an EARLIER build -- check that
# revision out, `python -c "import __graft_entry__ as g; g.build()"`, `cp adaptive-graph-convolutional-network_amd/lib/libagcn_hip.so
# tools/build/libagcn_base.so`; an older ABI works as long as its structures only GREW since) against the in-tree one, three alternations of the
# headline bench; before it (FULL=1: the whole GPU suite, else) the parity tests that exercise the members, and the phase stamps of a
# steady-state launch sequence (tools/stamps.py --build first: the stamps library is a build of its own)
cd $GRAFT_REPO_ROOT
O=gpurun_out/p1; mkdir -p $O
if [ "$FULL" = 1 ]; then timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $O/tests.log
else timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "bf16 or full_size or members or spread or packed_vs_oracle or golden or tagged or bit_rep or epoch or reslap" 2>&1 | tail -4 > $O/tests.log; fi
tail -4 $O/tests.log
AGCN_STAMP_REPS=9 AGCN_STAMP_SEQ=fbfbfbfb timeout 200 python tools/stamps.py 2>&1 | grep -A1 "blocks\|over 9\|launch " | grep -v "^--\|stamp idx\|median of medians" > $O/stamps.txt; cat $O/stamps.txt
for rep in 1 2 3; do
for v in base new; do
  L=adaptive-graph-convolutional-network_amd/lib/libagcn_hip.so; [ $v = base ] && L=tools/build/libagcn_base.so
  AGCN_HIP_LIB=$PWD/$L timeout 300 python bench.py --no-cpu-baseline --no-secondary --skip-other-mode > $O/bench_$v.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['config']['single_step_launches']['ms_per_step'], d['roofline']['avg_launch_us'])"
done
done
