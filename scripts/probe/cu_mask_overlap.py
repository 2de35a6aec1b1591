"""Can the HBM-bound feature extraction and the MFMA-bound int8 screen share the chip side by side?  Streams with CU masks
(hipExtStreamCreateWithCUMask, handed to the library per thread through aps_set_stream): the 64 x 4K views' extraction and the
2016-pair matching alone on all CUs, alone on half of them (every XCD keeps half of its CUs: mask bits are dealt round-robin to
the XCDs), and both at once on disjoint halves / unmasked.  usage: cu_mask_overlap.py [sift_share_of_256_cus]"""
import ctypes as C, sys, threading, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
fm = import_module(apsamd.__name__ + ".featureMatching")
capi = apsamd._capi
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int
n_sift = int(sys.argv[1]) if len(sys.argv) > 1 else 128


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in bits) for w in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return st


imgs, cams = synth.make_scene(8, 8, 3840, 2160, 8000.0, 0.4, device="cuda", finest_px=16.0)
inp = pl.default_input(bands=5)
descs = [d for d, _ in pl.sift_many(inp, imgs)]
order = fm.pair_order(len(imgs))
A_bits = set(range(n_sift, 256))   # matching
B_bits = set(range(0, n_sift))     # extraction
WORKERS = 10


def sync():
    capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()


def run_sift(bits, out):
    def worker(k):
        if bits is not None:
            capi.check(capi.lib.aps_set_stream(masked_stream(bits)))
        for i in range(k, len(imgs), WORKERS):
            fm.sift_extract({"detector": "SIFT"}, imgs[i], device_out=True)
        capi.check(capi.lib.aps_synchronize())
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(k,)) for k in range(WORKERS)]
    [t.start() for t in th]; [t.join() for t in th]
    out["sift"] = (time.perf_counter() - t0) * 1e3


def run_match(bits, out):
    def body():
        if bits is not None:
            capi.check(capi.lib.aps_set_stream(masked_stream(bits)))
        t0 = time.perf_counter()
        fm.match_pairs_csr(descs, order, inp["Ratiothreshold"], inp["Matchingthreshold"], True, device_out=True)
        capi.check(capi.lib.aps_synchronize())
        out["match"] = (time.perf_counter() - t0) * 1e3
    t = threading.Thread(target=body); t.start(); t.join()


def both(sb, mb, out):
    t0 = time.perf_counter()
    ts = threading.Thread(target=run_sift, args=(sb, out))
    tm = threading.Thread(target=run_match, args=(mb, out))
    ts.start(); tm.start(); ts.join(); tm.join()
    out["total"] = (time.perf_counter() - t0) * 1e3


for rep in range(2):
    o = {}
    run_sift(None, o); run_match(None, o)
    print(f"alone, all CUs:         extraction {o['sift']:.1f} ms, matching {o['match']:.1f} ms, sum {o['sift'] + o['match']:.1f}", flush=True)
    o = {}
    run_sift(B_bits, o); run_match(A_bits, o)
    print(f"alone, masked ({n_sift}/{256 - n_sift} CUs): extraction {o['sift']:.1f} ms, matching {o['match']:.1f} ms", flush=True)
    o = {}
    both(B_bits, A_bits, o)
    print(f"together, disjoint masks: extraction {o['sift']:.1f} ms, matching {o['match']:.1f} ms, both done after {o['total']:.1f} ms", flush=True)
    o = {}
    both(None, None, o)
    print(f"together, unmasked:       extraction {o['sift']:.1f} ms, matching {o['match']:.1f} ms, both done after {o['total']:.1f} ms", flush=True)
