"""Host-side cost of issuing the decode engine's forward sweep: the eager C call (one hipLaunchKernel per kernel) against
a hipGraph replay of the same sweep.  Host time = time until the call returns with an idle queue ahead of it (the GPU
is drained before each measurement, so the call never blocks on a full queue); wall = until the GPU is done.
usage: python tools/graph_host_bench.py [rows]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssrnet_style_transfer_amd import ops
from mssrnet_style_transfer_amd.config import MSSRConfig
from mssrnet_style_transfer_amd.style_generator_builder import build_all
from mssrnet_style_transfer_amd.decode_engine import soft_decode
dev = torch.device("cuda", 0)
ops.set_compute_dtype(torch.bfloat16)
cfg = MSSRConfig()
nets = build_all(cfg, device=dev, seed=123)
model = nets["model"]; model.eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 240
T = 28
mem = torch.randn(17, B, cfg.d_model, device=dev).bfloat16()
lens = torch.randint(4, 18, (B,), device=dev)
src = torch.zeros(17, B, 1, dtype=torch.long, device=dev)
model.decoder.init_state(src, None, None)
def run():
    with torch.no_grad():
        return soft_decode(model, cfg, mem, lens, T)
def host_and_wall(fn, reps=5):
    fn(); torch.cuda.synchronize()
    h = w = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        h += t1 - t0; w += t2 - t0
    return h / reps * 1e3, w / reps * 1e3
print("B=%d T=%d eager sweep : host %.2f ms, wall %.2f ms" % ((B, T) + host_and_wall(run)))
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    run()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.graph(g):
    out = run()
torch.cuda.synchronize()
print("capture + instantiate: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
print("graph replay          : host %.2f ms, wall %.2f ms" % host_and_wall(g.replay))
