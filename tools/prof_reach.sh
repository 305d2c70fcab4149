#!/bin/bash
# rocprofv3 kernel stats of tools/probe_reslap_reach.py (HIV-shaped batches: members / wide members / beyond): which kernel variants
# ran and for how long.  usage (GPU box): bash tools/prof_reach.sh <tag>   ->  gpurun_out/reach/<tag>_member_reach_<variant>_kernel_stats.csv
T=${1:-rNN}
O=$GRAFT_REPO_ROOT/gpurun_out/reach
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in reslap sgc; do
  rm -rf $O/$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- python3 $GRAFT_REPO_ROOT/tools/probe_reslap_reach.py $v > $O/$v.log 2>&1
  cp $O/$v/*/*kernel_stats.csv $O/${T}_member_reach_${v}_kernel_stats.csv
  echo "== $v"; cut -d, -f1-4 $O/${T}_member_reach_${v}_kernel_stats.csv | grep "agcn" | cut -c1-150 | head -12
done
