#!/bin/bash
# A/B of the screening kernel's two MFMA shapes on the bench workload (same box, back to back).
set -e
mkdir -p gpurun_out
python -m pytest tests/test_match_gpu.py tests/test_global_gpu.py tests/test_golden.py -m gpu -x -q > gpurun_out/t_match.txt 2>&1 || { tail -30 gpurun_out/t_match.txt; exit 1; }
tail -1 gpurun_out/t_match.txt
for shape in 16 32 16 32; do
  APS_SCREEN_SHAPE=$shape python bench.py --steps 4 --warmup 2 --cpu-baseline off --end-to-end off --global-probe off > gpurun_out/b$shape.json 2> gpurun_out/b$shape.err
  python - <<PY
import json
d=json.load(open("gpurun_out/b$shape.json"))
print("shape $shape", d["value"], d["ms_per_step"], "matching", d["stages_ms_per_step"]["matching"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("match")}, d["config"]["int8_screen_survivor_share"])
PY
done
