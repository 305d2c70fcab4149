#!/bin/bash
# GPU quick check of the parameter-gradient kernels: the tagged hand-off tests, the phase stamps, the headline bench
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tagged or static_step" > gpurun_out/s3_t1.log 2>&1; tail -3 gpurun_out/s3_t1.log
timeout 200 python tools/stamps.py > gpurun_out/s3_stamps.txt 2>&1; tail -6 gpurun_out/s3_stamps.txt
timeout 300 python bench.py "$@" > gpurun_out/s3_bench.json 2> gpurun_out/s3_bench.err; tail -3 gpurun_out/s3_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/s3_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d.get("single_step_launches"))
PY
