"""The critics' phase alone, for a kernel profile: one shared generator forward, then N calls of
train_discriminator(batch, _shared=...) on its outputs.  Profile two values of N and subtract
(tools/dphase_diff.py) -- what is left is exactly one D phase's kernels.
usage: python tools/dphase_prof.py N"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssrnet_style_transfer_amd import ops
from mssrnet_style_transfer_amd.config import MSSRConfig
from mssrnet_style_transfer_amd.style_generator_builder import build_all
from mssrnet_style_transfer_amd.synth import make_batch
from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer, set_mode
N = int(sys.argv[1])
ops.set_compute_dtype(torch.bfloat16)
cfg = MSSRConfig()
nets = build_all(cfg, device="cuda", seed=123)
batch = make_batch(80, 16, cfg.vocab_size, seed=123, device="cuda")
tr = Trainer(nets, cfg)
labels, _, real_src, real_len, vocab_target = tr.unpack_batch(batch)
set_mode(tr.style_generator, True)
set_mode(tr.model, True)
with torch.no_grad():
    _, _, _, fake, rep_d = tr._generator_forward_batched(labels, real_src, real_len, vocab_target.size(0), with_eval=True)
fake, rep_d = fake.detach(), rep_d.detach()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    tr.train_discriminator(batch, _shared=(rep_d, fake))
e1.record()
torch.cuda.synchronize()
print("D phase: %.3f ms per call over %d calls" % (e0.elapsed_time(e1) / max(N, 1), N))
