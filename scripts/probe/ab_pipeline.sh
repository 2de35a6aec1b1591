set -e
B="python bench.py --steps 10 --warmup 2 --cpu-baseline off --end-to-end off --global-probe off --with-gain off"
show() { python - "$1" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p=d.get('pipeline',{})
print(sys.argv[1], 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'min', d['ms_per_step_min'], d['ms_per_step_series'], 'seq', p.get('ms_per_step_sequential'), p.get('ms_per_step_sequential_median'), p.get('stages_ms_per_step_pipelined'))
P
}
$B > gpurun_out/pl_all.json 2> gpurun_out/pl_all.err; show gpurun_out/pl_all.json
APS_BENCH_PREFETCH_FIRST=3 $B > gpurun_out/pl_f3.json 2> gpurun_out/pl_f3.err; show gpurun_out/pl_f3.json
APS_BENCH_PREFETCH_FIRST=5 $B > gpurun_out/pl_f5.json 2> gpurun_out/pl_f5.err; show gpurun_out/pl_f5.json
APS_BENCH_PREFETCH_FIRST=2 $B > gpurun_out/pl_f2.json 2> gpurun_out/pl_f2.err; show gpurun_out/pl_f2.json
$B > gpurun_out/pl_all2.json 2> gpurun_out/pl_all2.err; show gpurun_out/pl_all2.json
