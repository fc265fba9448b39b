"""Where the HOST spends a WGAN step: cProfile of Trainer.train_wgan_step over K steps (no synchronisation inside), the
top functions by own time and by cumulative time.  usage: python tools/host_profile.py [steps]"""
import cProfile, io, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssrnet_style_transfer_amd import ops
from mssrnet_style_transfer_amd.config import MSSRConfig
from mssrnet_style_transfer_amd.style_generator_builder import build_all
from mssrnet_style_transfer_amd.synth import make_batch
from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ops.set_compute_dtype(torch.bfloat16)
cfg = MSSRConfig()
nets = build_all(cfg, device="cuda", seed=123)
batch = make_batch(80, 16, cfg.vocab_size, seed=123, device="cuda")
tr = Trainer(nets, cfg)
for _ in range(3):
    tr.train_wgan_step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    tr.train_wgan_step(batch)
    torch.cuda.synchronize()       # isolated steps: host time with an idle queue
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print("==== per %d steps, by %s" % (K, key))
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[4:60]))
