set -e
B="python bench.py --steps 6 --warmup 2 --cpu-baseline off --end-to-end off --global-probe off --with-gain off"
show() { python - "$1" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'min', d['ms_per_step_min'], d['stages_ms_per_step'])
P
}
$B > gpurun_out/ov_base.json 2> gpurun_out/ov_base.err; show gpurun_out/ov_base.json
APS_MATCH_OVERLAP_CHUNK=16 $B > gpurun_out/ov_c16_whole.json 2> gpurun_out/ov_c16_whole.err; show gpurun_out/ov_c16_whole.json
APS_MATCH_OVERLAP_CHUNK=16 APS_MATCH_SHARE_SIMD=1 $B > gpurun_out/ov_c16_share.json 2> gpurun_out/ov_c16_share.err; show gpurun_out/ov_c16_share.json
APS_MATCH_OVERLAP_CHUNK=16 APS_MATCH_SHARE_SIMD=1 APS_OVERLAP_PRIORITIES=1 $B > gpurun_out/ov_c16_share_prio.json 2> gpurun_out/ov_c16_share_prio.err; show gpurun_out/ov_c16_share_prio.json
APS_MATCH_OVERLAP_CHUNK=8 APS_MATCH_SHARE_SIMD=1 APS_OVERLAP_PRIORITIES=1 $B > gpurun_out/ov_c8_share_prio.json 2> gpurun_out/ov_c8_share_prio.err; show gpurun_out/ov_c8_share_prio.json
APS_MATCH_SHARE_SIMD=1 $B > gpurun_out/ov_share_only.json 2> gpurun_out/ov_share_only.err; show gpurun_out/ov_share_only.json
