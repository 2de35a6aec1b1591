set -e
B="python bench.py --steps 10 --warmup 2 --cpu-baseline off --global-probe off --with-gain off"
show() { python - "$1" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p=d.get('pipeline',{})
print(sys.argv[1], 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'e2e', d['ms_per_step_end_to_end'], 'seq', p.get('ms_per_step_sequential'), p.get('ms_per_step_sequential_median'), p.get('stages_ms_per_step_pipelined'), d['stages_ms_per_step_end_to_end'])
P
}
$B > gpurun_out/pr_a.json 2> gpurun_out/pr_a.err; show gpurun_out/pr_a.json
APS_BENCH_MAIN_PRIORITY=0 APS_SIFT_STREAM_PRIORITY=-1 $B > gpurun_out/pr_b.json 2> gpurun_out/pr_b.err; show gpurun_out/pr_b.json
APS_BENCH_MAIN_PRIORITY=0 $B > gpurun_out/pr_c.json 2> gpurun_out/pr_c.err; show gpurun_out/pr_c.json
$B > gpurun_out/pr_a2.json 2> gpurun_out/pr_a2.err; show gpurun_out/pr_a2.json
APS_BENCH_MAIN_PRIORITY=0 APS_SIFT_STREAM_PRIORITY=-1 $B > gpurun_out/pr_b2.json 2> gpurun_out/pr_b2.err; show gpurun_out/pr_b2.json
