#!/bin/bash
# kernel timeline of the last full matching call of scripts/probe/probe_match_stage.py (64 x 4K scene, 2016 pairs): every launch with the
# gap in front of it - what the stage pays beside its three big kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ms
(cd $R && rocprofv3 --kernel-trace --output-format csv -d /tmp/ms -o p -- python3 scripts/probe/probe_match_stage.py > $R/gpurun_out/match_stage_trace.txt 2>&1) || { tail -5 $R/gpurun_out/match_stage_trace.txt; exit 1; }
python3 - <<'PY' | tee -a $R/gpurun_out/match_stage_trace.txt
import csv
rows = sorted(csv.DictReader(open("/tmp/ms/p_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "absmax" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"]); prev_end = t0; busy = 0
for r in rows[last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  +{(e - s) / 1e3:9.1f} us  grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]}  {r["Kernel_Name"].split("(")[0][-60:]}')
    busy += e - s; prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us")
PY
