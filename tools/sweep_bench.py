"""GPU time of the decode sweeps alone, as the WGAN step runs them (bs80: 240 rows forward with 80 eval rows, row groups
18 / 18 / 28 steps, late-step regime; backward over 160 -> 80 rows), from events around mssr_decode_fwd / _bwd.
usage: python tools/sweep_bench.py [reps]    (MSSR_* environment switches apply)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssrnet_style_transfer_amd import decode_engine as DE, ops
from mssrnet_style_transfer_amd.config import MSSRConfig
from mssrnet_style_transfer_amd.style_generator_builder import build_all
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
ops.set_compute_dtype(torch.bfloat16)
cfg = MSSRConfig()
nets = build_all(cfg, device=dev, seed=123)
model = nets["model"]; model.train()
B, Ls, T0 = 80, 17, 18
T = T0 + cfg.max_extra_dec_steps
torch.manual_seed(1)
mem = torch.randn(Ls, 3 * B, cfg.d_model, device=dev).bfloat16().requires_grad_(True)
lens = torch.randint(4, Ls + 1, (B,), device=dev).repeat(3)
src = torch.zeros(Ls, 3 * B, 1, dtype=torch.long, device=dev)
groups = [(T0, 0, B), (T0, B, 2 * B), (T, 2 * B, 3 * B)]
ws = [torch.randn(te, b1 - b0, cfg.vocab_size, device=dev) * 1e-3 for te, b0, b1 in groups]
DE.sweep_events = []
for it in range(reps + 3):
    model.decoder.init_state(src, None, None)
    outs = DE.soft_decode(model, cfg, mem, lens, T, eval_rows=B, groups=groups)
    loss = sum((o * w).sum() for o, w in zip(outs[1:], ws[1:]))
    loss.backward()
    model.zero_grad(); mem.grad = None
torch.cuda.synchronize()
ev = DE.sweep_events[6:]            # skip the eager + recording iterations
for kind in ("decode_fwd", "decode_bwd"):
    ts = [a.elapsed_time(b) for k, a, b in ev if k == kind]
    print("%s: %.3f ms (min %.3f, %d sweeps; %d recorded graphs)" % (kind, sum(ts) / len(ts), min(ts), len(ts),
                                                                   DE.L.load().mssr_decode_graphs()))
