cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/full/tests.log; tail -3 gpurun_out/full/tests.log
timeout 900 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err; tail -c 300 gpurun_out/full/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/full/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["single_step_launches"]["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
for w in d["secondary"]["workloads"]:
    print("%-110s %s %s" % (w.get("workload","")[:110], w.get("ms_per_step", w.get("ms_per_batch")), w.get("value")))
PY
