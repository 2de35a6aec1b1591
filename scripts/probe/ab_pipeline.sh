set -e
B="python bench.py --steps 10 --warmup 2 --cpu-baseline off --end-to-end off --global-probe off --with-gain off"
show() { python - "$1" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'min', d['ms_per_step_min'], d['ms_per_step_series'], d['stages_ms_per_step'])
P
}
export APS_BENCH_PIPELINE=1 APS_BENCH_MAIN_PRIORITY=1
$B > gpurun_out/pl_on_hi.json 2> gpurun_out/pl_on_hi.err; show gpurun_out/pl_on_hi.json
APS_BENCH_PREFETCH_AT=ransac $B > gpurun_out/pl_on_hi_r.json 2> gpurun_out/pl_on_hi_r.err; show gpurun_out/pl_on_hi_r.json
APS_BENCH_PREFETCH_AT=ransac APS_SIFT_WORKERS=8 $B > gpurun_out/pl_on_hi_r_w8.json 2> gpurun_out/pl_on_hi_r_w8.err; show gpurun_out/pl_on_hi_r_w8.json
APS_BENCH_PREFETCH_AT=ransac APS_SIFT_WORKERS=12 $B > gpurun_out/pl_on_hi_r_w12.json 2> gpurun_out/pl_on_hi_r_w12.err; show gpurun_out/pl_on_hi_r_w12.json
