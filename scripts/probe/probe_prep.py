"""The matcher's per-set preparation alone (absmax probe, prep_desc, q8_desc) on 64 sets of ~19.8 k descriptors: a pair list
that touches every set once (i, i+1) keeps the screen small.  Run under rocprofv3 --kernel-trace (scripts/prep_trace.sh)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
fm = import_module(apsamd.__name__ + ".featureMatching")
capi = apsamd._capi
n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device="cuda").manual_seed(1)
descs = []
for i in range(n_sets):
    q = torch.rand((19800 + 7 * i, 128), generator=g, device="cuda").pow_(3).mul_(255).round_()
    descs.append((q / q.norm(dim=1, keepdim=True)).contiguous())
pairs = [(i, i + 1) for i in range(0, n_sets - 1, 2)]
for r in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = fm.match_pairs_csr(descs, pairs, 0.6, 1.0, True, device_out=True)
    capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()
    print(f"call {r}: {(time.perf_counter() - t0) * 1e3:.2f} ms for {len(pairs)} pairs over {n_sets} sets", flush=True)
