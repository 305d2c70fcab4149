# GPU quick check of the bf16-pipe phases (csrc/spec_bfpipe.h): the parity tests that exercise the specialised kernels, the HIV-shaped
# autograd step with fp32 and bf16 features (tools/hiv_step.py), the headline bench and its phase stamps -- pipe off (AGCN_BFPIPE=0) / on
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/bfp; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "bf16 or full_size or members or spread or packed_vs_oracle or golden or tagged or two_threads or bit_rep or epoch" 2>&1 | tail -6 > $O/tests.log
for m in 0 12; do
  AGCN_BFPIPE=$m timeout 300 python tools/hiv_step.py > $O/step_$m.txt 2>&1
  AGCN_BFPIPE=$m timeout 300 python bench.py --no-cpu-baseline --no-secondary --skip-other-mode > $O/bench_$m.json 2>/dev/null
  AGCN_BFPIPE=$m timeout 200 python tools/stamps.py > $O/stamps_$m.txt 2>&1
done
tail -3 $O/tests.log
grep -h "us per step" $O/step_*.txt
python - <<'PY'
import json
for m in ("0","12"):
    d=json.load(open("gpurun_out/bfp/bench_%s.json"%m)); print(m, d["value"], d["ms_per_step"], d["config"]["single_step_launches"]["ms_per_step"], d["roofline"]["avg_launch_us"])
PY
