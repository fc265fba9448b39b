"""Every GEMM launch of one WGAN step (HIP events per dispatch, ops.start_profile), grouped by kernel and shape:
launch count, total time, TFLOP/s and GB/s of algorithmic bytes.  Excludes the decode sweep's per-step GEMMs when
--no-decode is given (M <= 240 skinny launches).   python tools/gemm_shapes.py [--workload wgan|reconstruct|long]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssrnet_style_transfer_amd import ops  # noqa: E402
from mssrnet_style_transfer_amd.config import MSSRConfig  # noqa: E402
from mssrnet_style_transfer_amd.style_generator_builder import build_all  # noqa: E402
from mssrnet_style_transfer_amd.synth import make_batch  # noqa: E402
from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer  # noqa: E402


def main():
    wl = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "wgan"
    ops.set_compute_dtype(torch.bfloat16)
    cfg = MSSRConfig()
    nets = build_all(cfg, device="cuda", seed=123)
    batch = make_batch(256, 128, cfg.vocab_size, seed=123, device="cuda", min_words=64) if wl == "long" else \
        make_batch(80, 16, cfg.vocab_size, seed=123, device="cuda")
    trainer = Trainer(nets, cfg)
    step = (lambda: trainer.train_reconstruct(batch)) if wl == "reconstruct" else (lambda: trainer.train_wgan_step(batch))
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ops.start_profile()
    step()
    rec = ops.stop_profile()
    agg = collections.OrderedDict()
    for r in rec:
        a = agg.setdefault((r["kernel"], r["shape"]), [0, 0.0, r["flops"], r["bytes"]])
        a[0] += 1
        a[1] += r["ms"]
    tot = sum(v[1] for v in agg.values())
    print("%d GEMM launches, %.2f ms" % (len(rec), tot))
    for (k, shp), (n, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print("%5d x %-34s %-28s %8.3f ms  avg %7.1f us  %7.1f TFLOP/s %7.1f GB/s" % (
            n, k, "b%d %dx%dx%d" % shp, ms, ms / n * 1e3, fl * n / ms / 1e9, by * n / ms / 1e6))


if __name__ == "__main__":
    main()
