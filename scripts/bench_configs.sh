#!/bin/bash
# Round 6: one bench line per BASELINE.json config (the default invocation is configs[2]); each line lands in gpurun_out/.
# usage: scripts/bench_configs.sh [tag]   (copy the files you want judged to profiles/)
tag=${1:-r06z}
mkdir -p gpurun_out
for cfg in 0 1 3 4; do
  steps=3; [ $cfg = 0 ] && steps=10; [ $cfg = 1 ] && steps=10
  python bench.py --config $cfg --steps $steps --warmup 1 > gpurun_out/${tag}_bench_cfg$cfg.json 2> gpurun_out/${tag}_bench_cfg$cfg.err || { echo "config $cfg failed"; tail -5 gpurun_out/${tag}_bench_cfg$cfg.err; }
  echo "config $cfg: $(python -c "import json;d=json.load(open('gpurun_out/${tag}_bench_cfg$cfg.json'));print(d['value'],'MPix/s',d['ms_per_step'],'ms', d['stages_ms_per_step'], (d['roofline'] or {}).get('kernel','')[:40], (d['roofline'] or {}).get('frac'), d.get('cpu_baseline',{}).get('value'))")"
done
python bench.py --config 1 --reference-defaults --steps 10 --warmup 1 > gpurun_out/${tag}_bench_cfg1_reference_defaults.json 2> gpurun_out/${tag}_bench_cfg1_reference_defaults.err || tail -5 gpurun_out/${tag}_bench_cfg1_reference_defaults.err
echo "config 1, reference defaults: $(python -c "import json;d=json.load(open('gpurun_out/${tag}_bench_cfg1_reference_defaults.json'));print(d['value'],'MPix/s',d['ms_per_step'],'ms', d['stages_ms_per_step'])")"
