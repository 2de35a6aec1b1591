"""GPU: the multi-rank driver's code path on the real collective backend.  A one-GPU box cannot hold two RCCL ranks
("Duplicate GPU detected"), and the 2-rank equality test therefore runs over gloo; here the SAME N > 1 code - chunked
descriptor exchange with asynchronous handles, image all-gather, packed all-reduces, RANSAC record exchange, per-component
and per-tile render shards, tile gather to the root - runs over "nccl" with a one-rank group (the test hook
APS_PARALLEL_FORCE_COLLECTIVES=1 disables the single-rank shortcuts), and must reproduce the plain run byte for byte."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_once():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_PORT=str(port))
    env.pop("APS_PARALLEL_FORCE_COLLECTIVES", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_one_rank_runner.py")], env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=600)


def test_forced_collectives_over_nccl_equal_the_plain_run(gpu):
    # The child is a second process on the same GPU.  Late in a full-suite run this process holds most of the device in its
    # workspace pools and in torch's cache (the 256 x 4K and 500-view tests ran before): RCCL's 512 MB bring-up allocation in
    # the child then fails with "Failed to CUDA calloc" (seen in round 5, one run in three).  Hand the caches back first.
    import torch
    from importlib import import_module

    import_module(gpu.__name__ + ".pipeline").release_device_memory()
    free, total = torch.cuda.mem_get_info()
    assert free > 16 << 30, "only %.1f of %.1f GB of device memory free before the child starts" % (free / 2**30, total / 2**30)
    r = _run_once()
    if r.returncode != 0 and "AssertionError" not in r.stdout + r.stderr:
        # The child failed before it could compare anything (seen once in round 5, in the middle of a full-suite run, with
        # nothing but torch's shutdown warning in the captured tail): bring-up of the process group beside the parent's own
        # GPU context.  One second attempt on a new port; a COMPARISON failure (AssertionError) is never retried.
        print("first attempt failed without an assertion:\n" + r.stdout[-1500:] + "\n" + r.stderr[-1500:])
        out_dir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out_dir):  # (kept for inspection: a retry must not hide what happened)
            with open(os.path.join(out_dir, "nccl_one_rank_first_attempt.txt"), "w") as f:
                f.write("rc %d\n--- stdout ---\n%s\n--- stderr ---\n%s\n" % (r.returncode, r.stdout[-20000:], r.stderr[-20000:]))
        r = _run_once()
    if not (r.returncode == 0 and "OK" in r.stdout):
        # (the runner prints its traceback to stdout behind a marker; RCCL's banner and torch's shutdown warning are noise)
        tail = r.stdout[r.stdout.find("RUNNER FAILED"):] if "RUNNER FAILED" in r.stdout else r.stdout[-1500:]
        err = "\n".join(l for l in r.stderr.splitlines() if "ProcessGroupNCCL" not in l and "amdgpu.ids" not in l)[-2500:]
        pytest.fail("one-rank RCCL run failed (rc %d)\n--- stdout ---\n%s\n--- stderr ---\n%s" % (r.returncode, tail, err), pytrace=False)
