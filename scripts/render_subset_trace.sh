#!/bin/bash
# kernel timeline of ONE render of a rank's share of the tiles (default 0/8) on the bench scene: what a rank of eight pays beside
# its tiles' kernels (image preparation, tables, gaps between launches).  usage: bash scripts/render_subset_trace.sh [first/step]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/rs
(cd $R && rocprofv3 --kernel-trace --output-format csv -d /tmp/rs -o p -- python3 scripts/probe/probe_render.py 4 8x8 multiband ${1:-0/8} > $R/gpurun_out/render_subset.txt 2>&1) || { tail -5 $R/gpurun_out/render_subset.txt; exit 1; }
grep "^render" $R/gpurun_out/render_subset.txt
python3 - <<'PY' | tee -a $R/gpurun_out/render_subset.txt
import csv
rows = sorted(csv.DictReader(open("/tmp/rs/p_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "to_rgba_batch" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"]); prev_end = t0
for r in rows[last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  +{(e - s) / 1e3:8.1f} us  grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]}  {r["Kernel_Name"].split("(")[0][-56:]}')
    prev_end = max(prev_end, e)
PY
