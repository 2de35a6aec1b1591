"""How a 737 MB device -> pinned-host copy on a side stream behaves beside kernels (bench.py's panorama download):
(1) how long the issuing host thread is held by copy_(non_blocking=True); (2) the same from a helper thread; (3) what a
bandwidth-bound kernel on another stream costs while the copy is in flight."""
import threading, time
import torch
n = 737_000_000
src = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
dst = torch.empty(n, dtype=torch.uint8, pin_memory=True)
a = torch.empty(256 << 20, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
side = torch.cuda.Stream(priority=-1)
torch.cuda.synchronize()


def kernel_ms(reps=5):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for r in range(reps):
        torch.mul(a, 1.0001, out=b)
        ev[r + 1].record()
    torch.cuda.synchronize()
    return [ev[r].elapsed_time(ev[r + 1]) for r in range(reps)]


print("2 GB elementwise kernel alone (ms):", ["%.2f" % t for t in kernel_ms()])
for trial in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        dst.copy_(src, non_blocking=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"copy_(non_blocking=True): issuing thread held {1e3 * (t1 - t0):.2f} ms, copy done after {1e3 * (t2 - t0):.2f} ms")
for chunks in (8, 32):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step = (n + chunks - 1) // chunks
    with torch.cuda.stream(side):
        for c in range(chunks):
            dst[c * step:(c + 1) * step].copy_(src[c * step:(c + 1) * step], non_blocking=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{chunks} chunks: issuing thread held {1e3 * (t1 - t0):.2f} ms, done after {1e3 * (t2 - t0):.2f} ms")


def bg():
    with torch.cuda.stream(side):
        dst.copy_(src, non_blocking=True)


for trial in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = threading.Thread(target=bg)
    th.start()
    time.sleep(0.001)
    ks = kernel_ms(8)
    t1 = time.perf_counter()
    th.join()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"copy from a helper thread: kernels beside it (ms) {['%.2f' % t for t in ks]}, kernels issued+done after {1e3 * (t1 - t0):.2f} ms, all done {1e3 * (t2 - t0):.2f} ms")
