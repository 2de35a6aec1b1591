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


def test_forced_collectives_over_nccl_equal_the_plain_run(gpu):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_PORT=str(port))
    env.pop("APS_PARALLEL_FORCE_COLLECTIVES", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_one_rank_runner.py")], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
