"""Do two independent chains of small dependent kernels overlap on MI355X?  (decides whether independent parts of the
WGAN phases are worth running on separate streams.)  Chains of M x 512 x 512 K-split GEMMs, each launch reading the
previous launch's output; eager (one host thread feeding both streams) and as hipGraph replays.
usage: python tools/chains_bench.py [M] [N]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from mssrnet_style_transfer_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 160
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda")
ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)


def make_chain():
    w = [(torch.randn(512, 512, device=dev) / 512 ** 0.5).bfloat16() for _ in range(8)]
    a = torch.randn(M, 512, device=dev).bfloat16()
    b = torch.empty_like(a)
    return w, a, b


def run_chain(ch, n):
    w, a, b = ch
    for i in range(n):
        ops.gemm(a, w[i % 8], True, torch.bfloat16, out=b)
        a, b = b, a


c1, c2 = make_chain(), make_chain()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one(n):
    run_chain(c1, n)


def two(n):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    # interleave the launches of the two chains from this one thread
    w1, a1, b1 = c1
    w2, a2, b2 = c2
    for i in range(n):
        with torch.cuda.stream(s1):
            ops.gemm(a1, w1[i % 8], True, torch.bfloat16, out=b1)
        with torch.cuda.stream(s2):
            ops.gemm(a2, w2[i % 8], True, torch.bfloat16, out=b2)
        a1, b1, a2, b2 = b1, a1, b2, a2
    cur.wait_stream(s1)
    cur.wait_stream(s2)


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def graphed(fn):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    return g.replay


print("M=%d, %d launches per chain" % (M, N))
print("eager  one chain of N      : %.3f ms  (%.2f us/launch)" % (timeit(lambda: one(N)), timeit(lambda: one(N)) * 1e3 / N))
print("eager  one chain of 2N     : %.3f ms" % timeit(lambda: one(2 * N)))
print("eager  two chains of N     : %.3f ms" % timeit(lambda: two(N)))
r1, r2, r3 = graphed(lambda: one(N)), graphed(lambda: one(2 * N)), graphed(lambda: two(N))
t1 = timeit(r1)
print("graph  one chain of N      : %.3f ms  (%.2f us/launch)" % (t1, t1 * 1e3 / N))
print("graph  one chain of 2N     : %.3f ms" % timeit(r2))
print("graph  two chains of N     : %.3f ms" % timeit(r3))
