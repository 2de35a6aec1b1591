#!/bin/bash
# usage (through gpurun): scripts/warp_variants.sh TAG  -> per-variant rw_warp / rw_down / rw_up kernel times of the render probe
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export APS_RENDER_WORKERS=1
for v in exact staged 6; do
  rm -rf /tmp/prof_$v
  if [ "$v" = "exact" ]; then export APS_RENDER_EXACT=1; unset APS_WARP_VARIANT; elif [ "$v" = "staged" ]; then unset APS_RENDER_EXACT; unset APS_WARP_VARIANT; else unset APS_RENDER_EXACT; export APS_WARP_VARIANT=$v; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python3 scripts/probe/probe_render.py 3 > /dev/null 2>&1 || { echo "variant $v failed"; exit 1; }
  python3 - "$v" <<PY
import csv, sys
rows = list(csv.DictReader(open(f"/tmp/prof_{sys.argv[1]}/p_kernel_stats.csv")))
out = []
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("rw_warp", "rw_down", "rw_up", "rw_trig")):
        out.append(f"{float(r['TotalDurationNs'])/3e6:7.3f} ms {n.split('(')[0][-60:]}")
print("variant", sys.argv[1]); print("\n".join("   " + o for o in sorted(out, reverse=True)))
PY
done
