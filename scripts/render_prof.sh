#!/bin/bash
# usage (through gpurun): scripts/render_prof.sh TAG [test]   -> gpurun_out/TAG_render_kernel_stats.csv
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$2" = "test" ]; then
  python -m pytest tests/test_render_gpu.py tests/test_pipeline_gpu.py tests/test_fullsize_gpu.py -x -q > gpurun_out/${TAG}_test.log 2>&1; tail -5 gpurun_out/${TAG}_test.log
fi
export APS_RENDER_WORKERS=1
python3 scripts/probe/probe_render.py 3 || exit 1
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python3 scripts/probe/probe_render.py 3 > /dev/null 2>&1
cp /tmp/prof/p_kernel_stats.csv gpurun_out/${TAG}_render_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${TAG}_render_kernel_stats.csv")))
tot=0
for r in rows:
    n=r["Name"]
    if "synth" in n or "at::native" in n: continue
    per=float(r["TotalDurationNs"])/3e6
    tot+=per
    if per>0.05: print(f"{per:8.3f} ms/render  {int(r['Calls'])//3:4d} calls  {n[:110]}")
print(f"{tot:8.3f} ms/render kernels total")
PY
