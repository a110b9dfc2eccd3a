"""debug: which step leaves a sticky HIP error on the main thread"""
import ctypes as C, os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mxx_amd as mx
from mxx_amd import _ffi
from mxx_amd.parallel import GpuComm, all_shard_ranges
from oracle import oracle as O
from concurrent.futures import ThreadPoolExecutor
hip = C.CDLL("libamdhip64.so.7")
hip.hipPeekAtLastError.restype = C.c_int
def peek(tag):
    e = hip.hipPeekAtLastError()
    print(f"{tag}: last error = {e}", flush=True)
    if e: hip.hipGetLastError()
gpu = mx
n = 256
moduli = O.gen_crt_basis(n, 2, 24)
p0 = mx.GpuDCRTPolyParams(n, moduli, 12)
p1 = mx.GpuDCRTPolyParams(n, moduli, 12, gpu_ids=p0.gpu_ids(), dnum=101)
ps = [p0, p1]
comm = GpuComm(ps); peek("comm")
a = O.matrix_ntt(O.random_matrix(41, 2, 3, moduli, n), moduli)
b = O.matrix_ntt(O.random_matrix(42, 3, 5, moduli, n), moduli)
ranges = all_shard_ranges(5, 2)
def work(rank):
    p, sr = ps[rank], ranges[rank]
    ga = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True)
    gb = gpu.GpuDCRTPolyMatrix.from_rns(p, np.ascontiguousarray(b[:, sr.start:sr.stop]), True)
    r = ga * gb
    e = hip.hipPeekAtLastError(); print("mul worker", rank, "last error", e, flush=True)
    return r
with ThreadPoolExecutor(2) as ex:
    blocks = list(ex.map(work, range(2)))
peek("after mul workers")
fulls = comm.all_gather_columns(blocks); peek("gather products")
want = O.matmul(a, b, moduli)
for f in fulls:
    assert np.array_equal(f.to_rns(), want)
peek("checked products")
sampler = gpu.GpuDCRTPolyTrapdoorSampler(ps[0], 4.578)
td0, a0 = sampler.trapdoor(ps[0], 1); peek("trapdoor")
tds, pubs = [td0, td0.to_params(ps[1])], [a0, a0.to_params(ps[1])]; peek("to_params")
target = gpu.GpuDCRTPolyUniformSampler().sample_uniform(ps[0], 1, 5, gpu.DistType.FinRingDist()); peek("sample")
t_rns = target.to_rns()
ranges = all_shard_ranges(5, 2)
def pre(rank):
    p, sr = ps[rank], ranges[rank]
    t = gpu.GpuDCRTPolyMatrix.from_rns(p, np.ascontiguousarray(t_rns[:, sr.start:sr.stop]), True)
    x = sampler.preimage(p, tds[rank], pubs[rank], t)
    e = hip.hipPeekAtLastError(); print("worker", rank, "last error", e, flush=True)
    return x
with ThreadPoolExecutor(2) as ex:
    xs = list(ex.map(pre, range(2)))
peek("after workers")
fulls = comm.all_gather_columns(xs); peek("gather")
for pub, f in zip(pubs, fulls):
    assert pub * f == gpu.GpuDCRTPolyMatrix.from_rns(f.params, t_rns, True)
peek("check")
comm.close(); peek("close")
del fulls, xs; gc.collect(); peek("del outs")
del tds, pubs, td0, a0, target; gc.collect(); peek("del trapdoors")
del sampler; gc.collect(); peek("del sampler")
del ps, p1, comm; gc.collect(); peek("del ctx1")

p4 = mx.GpuDCRTPolyParams(4, O.gen_crt_basis(4, 2, 17), 1); peek("ctx n=4")
x4 = O.random_matrix(1, 2, 3, p4.moduli(), 4)
m4 = gpu.GpuDCRTPolyMatrix.from_rns(p4, x4, True); peek("from_rns n=4")
