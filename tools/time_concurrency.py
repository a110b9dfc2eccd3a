"""One stream per context: do concurrent host threads on ONE context lose against one context per thread?
Small launch-bound work (n = 256, 51-bit, L = 12: (1 x 76)(76 x 4) products) and large work (n = 2^14 products)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx

d = mx.DistType.FinRingDist()
us = mx.GpuDCRTPolyUniformSampler()

def bench(name, make_params, shape, iters, threads):
    r, k, c = shape
    def worker(p, out):
        a, b = us.sample_uniform(p, r, k, d), us.sample_uniform(p, k, c, d)
        x = a * b
        mx.gpu_device_sync()
        barrier.wait()
        for _ in range(iters):
            x = a * b
        out.append(x)
    for mode in ("one thread", "%d threads, one context" % threads, "%d threads, one context each" % threads):
        nt = 1 if mode == "one thread" else threads
        shared = make_params()
        ps = [make_params() if "each" in mode else shared for _ in range(nt)]
        barrier = threading.Barrier(nt + 1)
        outs = []
        ts = [threading.Thread(target=worker, args=(ps[i], outs)) for i in range(nt)]
        for t in ts: t.start()
        barrier.wait()
        t0 = time.perf_counter()
        for t in ts: t.join()
        mx.gpu_device_sync()
        el = time.perf_counter() - t0
        print(f"{name}: {mode:32s} {nt * iters / el:10.0f} products/s", flush=True)

bench("n=256 L=12 u64 (1x76)(76x4)", lambda: mx.GpuDCRTPolyParams(256, mx.gen_crt_basis(256, 12, 51), 17), (1, 76, 4), 2000, 4)
bench("n=2^14 L=4 (1x30)(30x120)  ", lambda: mx.GpuDCRTPolyParams(16384, mx.gen_crt_basis(16384, 4, 24), 12), (1, 30, 120), 300, 4)
