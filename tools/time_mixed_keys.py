"""Where the time of a mixed-key `preimage_batched_sharded` call goes (tools/README.md): the same 8 key groups of 2 requests
(a) one after another on one context, (b) dealt to 4 worker contexts but issued by ONE host thread, group by group,
(c) by a thread per worker (what the call does), (d) as (b) with one request per launch sequence replaced by the whole
group's host work measured without waiting for the device (issue time)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import trapdoor as T

p = mx.GpuDCRTPolyParams(256, mx.gen_crt_basis(256, 12, 51), 17)
s = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
us = mx.GpuDCRTPolyUniformSampler()
keys = [s.trapdoor(p, 2) for _ in range(8)]
targets = [[us.sample_uniform(p, 2, 4, mx.DistType.FinRingDist()) for _ in range(2)] for _ in range(8)]
W = 4
plan = []
for g, ((td, a), ts) in enumerate(zip(keys, targets)):
    w = g % W
    if w == 0:
        plan.append((p, td, a, ts))
    else:
        pw = T.worker_params(p, w)
        tdw, aw = td.replica_for(pw, a)
        plan.append((pw, tdw, aw, [t.to_params(pw) for t in ts]))

def timed(fn, reps=8):
    for _ in range(3):
        fn()
    out = []
    for _ in range(reps):
        mx.gpu_device_sync(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); mx.gpu_device_sync(); t2 = time.perf_counter()
        out.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    return statistics.median(o[0] for o in out), statistics.median(o[1] for o in out)

seq = lambda: [s.preimage_many(p, td, a, ts) for (td, a), ts in zip(keys, targets)]
one_thread = lambda: [s.preimage_many(pw, tdw, aw, ts) for pw, tdw, aw, ts in plan]
def threaded():
    from concurrent.futures import ThreadPoolExecutor
    def work(w):
        return [s.preimage_many(pw, tdw, aw, ts) for g, (pw, tdw, aw, ts) in enumerate(plan) if g % W == w]
    with ThreadPoolExecutor(W) as ex:
        return list(ex.map(work, range(W)))
for name, fn in (("one context, one after another", seq), ("4 worker contexts, one host thread", one_thread), ("4 worker contexts, a thread each", threaded)):
    issue, total = timed(fn)
    print(f"{name}: host issue {issue:.3f} ms, until the device is idle {total:.3f} ms")
