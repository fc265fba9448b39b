"""Every GEMM launch of one WGAN step by kernel and shape (ops.start_profile: HIP events around each dispatch):
launches, mean microseconds, TFLOP/s.  usage: python tools/gemm_records.py [kernel-name prefix]"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssrnet_style_transfer_amd import ops
from mssrnet_style_transfer_amd.config import MSSRConfig
from mssrnet_style_transfer_amd.style_generator_builder import build_all
from mssrnet_style_transfer_amd.synth import make_batch
from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
prefix = sys.argv[1] if len(sys.argv) > 1 else ""
ops.set_compute_dtype(torch.bfloat16)
cfg = MSSRConfig()
nets = build_all(cfg, device="cuda", seed=123)
batch = make_batch(80, 16, cfg.vocab_size, seed=123, device="cuda")
tr = Trainer(nets, cfg)
for _ in range(3):
    tr.train_wgan_step(batch)
torch.cuda.synchronize()
ops.start_profile()
tr.train_wgan_step(batch)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for r in ops.stop_profile():
    if r["kernel"].startswith(prefix):
        a = agg.setdefault((r["kernel"], r["shape"]), [0, 0.0, r["flops"]])
        a[0] += 1
        a[1] += r["ms"]
tot = sum(a[1] for a in agg.values())
print("%d shapes, %.3f ms per step" % (len(agg), tot))
for (k, shp), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%5d x %8.2f us = %7.3f ms  %7.1f TFLOP/s  %-44s batch,m,n,k=%s" % (n, ms / n * 1e3, ms, fl / (ms / n * 1e-3) / 1e12, k, shp))
