"""GPU: bench.py's own N > 1 path, rehearsed on one GPU.  `python bench.py --gpus 2` starts two ranks itself; with
APS_BENCH_REHEARSE=1 both sit on device 0 and talk over gloo (RCCL refuses two ranks on one GPU), so the shard arithmetic of the
bench line, the max-over-ranks reduction and rank 0's single JSON line are walked for real.  The stitch must not depend on the
rank count: verified pairs, feature counts and the cropped panorama's size equal the one-rank run's.  Three timed steps, so that
the pipelined form (the next step's extraction started from the after_matching hook of every rank) is walked at N = 2 as well."""
import json
import os
import subprocess
import sys
from importlib import import_module

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(n_gpus, env_extra):
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "APS_PARALLEL_FORCE_COLLECTIVES"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_gpus), "--steps", "3", "--warmup", "1",
                        "--grid", "3x3", "--size", "1280x720", "--bands", "3", "--cpu-baseline", "off", "--end-to-end", "off",
                        "--global-probe", "off", "--with-gain", "off"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_with_two_ranks_rehearsed_on_one_gpu_equals_one_rank(gpu):
    import_module(gpu.__name__ + ".pipeline").release_device_memory()  # (two more processes are about to share this GPU)
    one = _bench(1, {})
    two = _bench(2, {"APS_BENCH_REHEARSE": "1"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    for key in ("pairs_verified", "pairs_matched", "features_per_view", "panorama", "input_mpix"):
        assert one["config"][key] == two["config"][key], (key, one["config"][key], two["config"][key])
    assert one["pipeline"]["on"] and two["pipeline"]["on"]
    assert two["value"] > 0 and two["roofline"] is not None and two["steps"] == 3
