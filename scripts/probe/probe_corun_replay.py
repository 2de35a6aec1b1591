"""Round 6: which SIFT kernels are disturbed by int8-MFMA neighbours, kernel by kernel (debug library: make -C <pkg>/csrc debug,
APS_LIB_PATH=<pkg>/lib/libaps_hip_dbg.so APS_DBG_REPLAY=R).  Every extraction launches the extrema sweep, refine_kernel, orient_kernel and
descr_kernel R more times on the inputs the first launch saw and compares on the device; the pyramid's checksum is compared with the quiet
pass.  Co-runner modes as in probe_overlap_race3.py (64 / 2 int8 MFMA, 8 f16 MFMA, 32 VALU, 0 idle)."""
import sys, ctypes, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
lib = apsamd._capi.lib
lib.aps_dbg_corun.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
lib.aps_dbg_replay_report.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
def report(reset=1):
    buf = ctypes.create_string_buffer(2048)
    lib.aps_dbg_replay_report(buf, 2048, reset)
    return buf.value.decode()
W, H, f = 3840, 2160, 8000.0
imgs, cams = synth.make_scene(4, 4, W, H, f, 0.4, device="cuda", finest_px=16.0)
torch.cuda.synchronize()
inp = pl.default_input(bands=5)
sig = lambda out: [(int(d.shape[0]), float(d.double().sum())) for d, _ in out]
ref_sig = sig(pl.sift_many(inp, imgs))
print("quiet:", report(), flush=True)
for mode in [int(v) for v in sys.argv[1:]] or [0, 64, 8, 2, 32]:
    t0 = time.perf_counter()
    futs = pl.sift_submit(inp, imgs)
    k = 0
    while not all(fu.done() for fu in futs):
        apsamd._capi.check(lib.aps_dbg_corun(mode, 4096, 400))
        k += 1
    s = sig([fu.result() for fu in futs])
    nd = sum(1 for i in range(len(s)) if s[i] != ref_sig[i])
    print("mode %3d: %2d of %d views differ, %d co-runner launches, %.1f s | %s" % (mode, nd, len(s), k, time.perf_counter() - t0, report()), flush=True)
