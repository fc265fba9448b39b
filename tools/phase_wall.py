"""GPU time of the pieces of one WGAN step (Trainer.train_wgan_step, bs80, bf16) from events recorded around them in
the running loop (no synchronisation inside a step; the GPU is the bottleneck, so event differences are GPU time).
usage: python tools/phase_wall.py [steps]"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssrnet_style_transfer_amd import decode_engine as DE, ops
from mssrnet_style_transfer_amd.config import MSSRConfig
from mssrnet_style_transfer_amd.style_generator_builder import build_all
from mssrnet_style_transfer_amd.synth import make_batch
from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ops.set_compute_dtype(torch.bfloat16)
cfg = MSSRConfig()
nets = build_all(cfg, device="cuda", seed=123)
batch = make_batch(80, 16, cfg.vocab_size, seed=123, device="cuda")
tr = Trainer(nets, cfg)
marks = []


def timed(name, fn):
    def wrapper(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        marks.append((name, e0, e1))
        return out
    return wrapper


for name in ("_generator_forward_batched", "_text_critic_batched", "_style_critic_batched", "_finish",
             "_train_generator_batched", "train_discriminator", "train_wgan_step"):
    setattr(tr, name, timed(name, getattr(tr, name)))
for _ in range(4):
    tr.train_wgan_step(batch)
torch.cuda.synchronize()
marks.clear()
DE.sweep_events = []
for _ in range(K):
    tr.train_wgan_step(batch)
torch.cuda.synchronize()
tot = collections.OrderedDict()
for name, e0, e1 in marks + DE.sweep_events:
    tot.setdefault(name, []).append(e0.elapsed_time(e1))
for name, v in tot.items():
    print("%-30s %3.1f calls/step  %7.2f ms/step" % (name, len(v) / K, sum(v) / K))
# gaps between markers inside the generator phase (per step)
G = [m for m in marks if m[0] == "_train_generator_batched"]
F = [m for m in marks if m[0] == "_finish"][1::2]          # the generator phase's clip + Adam
B = [m for m in DE.sweep_events if m[0] == "decode_bwd"]
GF = [m for m in marks if m[0] == "_generator_forward_batched"]
FW = [m for m in DE.sweep_events if m[0] == "decode_fwd"]
n = min(len(G), len(F), len(B))
print("generator phase: critics forward + backward down to the sweep %.2f ms, sweep -> clip+Adam (weight gradients, encoder / "
      "style generator backward) %.2f ms, clip + Adam %.2f ms" % (
          sum(G[i][1].elapsed_time(B[i][1]) for i in range(n)) / n, sum(B[i][2].elapsed_time(F[i][1]) for i in range(n)) / n,
          sum(F[i][1].elapsed_time(F[i][2]) for i in range(n)) / n))
print("generator forward: before the sweep %.2f ms, after it %.2f ms" % (
    sum(GF[i][1].elapsed_time(FW[i][1]) for i in range(n)) / n, sum(FW[i][2].elapsed_time(GF[i][2]) for i in range(n)) / n))
