#!/bin/bash
# Render stage of the bench scene under rocprofv3 --kernel-trace --stats: wall time per render, crc32 of the panorama bytes and
# the average time of every rw_* kernel.  usage (on the GPU box): bash scripts/render_kernel_times.sh [tag] ; environment
# switches of the library (APS_*) are inherited, so two calls with different switches make an A/B with a byte comparison.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-run}
rm -rf /tmp/prof_$tag
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 scripts/probe/probe_render.py 4 > $R/gpurun_out/render_kernel_times_$tag.txt 2>&1) || exit 1
grep "render\|crc32" $R/gpurun_out/render_kernel_times_$tag.txt | grep -v rocprofv3
python3 - $tag <<'PY' | tee -a $R/gpurun_out/render_kernel_times_$tag.txt
import csv, glob, sys
f = glob.glob(f"/tmp/prof_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "rw_" in r["Name"]:
        print(f'  {r["Name"][:70]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
