#!/usr/bin/env python
"""Headline benchmark: train tokens/sec of the MSSRNet WGAN step (12L d512 ff1024, bs80/GPU).

One "step" = train_discriminator(batch) + train_generator(batch) on the same synthetic
Yelp-shaped batch (BASELINE.json configs[1]), README flags, dropout on,
Adam steps included.  tokens = non-pad target tokens of the batch (the reference's own
normaliser, onmt/trainer_wgan_wh.py:389-391).  N > 1: one process per GPU under
torch.distributed.run, batch-sharded (bs80 per GPU, weak scaling) with bucketed RCCL gradient
all-reduce.  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 5 --warmup 2
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=80)
    ap.add_argument("--max-words", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def roofline(records, dtype):
    """Dominant GEMM kernel instance of one instrumented step: algorithmic bytes (or flops) of all
    its launches / the sum of their HIP-event durations."""
    agg = {}
    for r in records:
        a = agg.setdefault(r["kernel"], dict(ms=0.0, bytes=0, flops=0, launches=0, key=r["in_dtype"]))
        a["ms"] += r["ms"]
        a["bytes"] += r["bytes"]
        a["flops"] += r["flops"]
        a["launches"] += 1
    if not agg:
        return None
    name, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
    sec = a["ms"] * 1e-3
    key = a["key"]
    ridge = MFMA_PEAK_TFLOPS[key] * 1e12 / (HBM_PEAK_GBS * 1e9)
    if a["flops"] / max(a["bytes"], 1) < ridge:
        ach, peak, unit, bound = a["bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
    else:
        ach, peak, unit, bound = a["flops"] / sec / 1e12, MFMA_PEAK_TFLOPS[key], "TFLOP/s", "mfma"
    # traffic: HBM bytes from rocprofv3 PMC counters need a separate profiler pass; not collected in-process
    return dict(bound=bound, achieved=round(ach, 2), peak=peak, unit=unit, frac=round(ach / peak, 5), traffic=None,
                kernel=name, launches_per_step=a["launches"], avg_launch_us=round(a["ms"] * 1e3 / a["launches"], 2),
                algorithmic_bytes_per_launch=int(a["bytes"] / a["launches"]),
                gemm_time_share={k: round(v["ms"], 3) for k, v in sorted(agg.items())})


def cpu_baseline(nets, cfg, batch):
    """The oracle (a CPU port of the reference path, oracle/) timed on this box's host cores on ONE
    full WGAN step of the same batch.  Reported, not a target."""
    from types import SimpleNamespace
    from oracle import phases as PH
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    states = {k: {n: v.detach().float().cpu() for n, v in m.state_dict().items()} for k, m in nets.items()}
    onets = PH.make_nets(states, cfg)
    b = SimpleNamespace(src=batch.src.cpu(), src_lengths=batch.src_lengths.cpu(), tgt=batch.tgt.cpu(),
                        labels=batch.labels.cpu())
    t0 = time.perf_counter()
    PH.train_discriminator(onets, b, cfg)
    PH.train_generator(onets, b, cfg)
    dt = time.perf_counter() - t0
    return dict(value=round(batch.n_tokens / dt, 2), unit="tokens/s", cores=threads, kind="port",
                sample="1 full WGAN step (D+G phase) of the same bs%d batch, fp32, no warm-up, %.1f s" % (
                    batch.batch_size, dt))


def main():
    args = parse()
    from mssrnet_style_transfer_amd import ops
    from mssrnet_style_transfer_amd.config import MSSRConfig
    from mssrnet_style_transfer_amd.parallel import GradSync, init_distributed
    from mssrnet_style_transfer_amd.style_generator_builder import build_all
    from mssrnet_style_transfer_amd.synth import make_batch
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    import torch.distributed as dist

    rank, local_rank, world = init_distributed()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no CPU fallback exists for the product path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ops.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)

    cfg = MSSRConfig()
    nets = build_all(cfg, device=dev, seed=123)               # same weights on every rank
    batch = make_batch(args.batch, args.max_words, cfg.vocab_size, seed=123 + rank, device=dev)
    torch.manual_seed(1000 + rank)
    trainer = Trainer(nets, cfg, grad_sync=GradSync(world) if world > 1 else None)

    def step():
        d = trainer.train_discriminator(batch)
        g = trainer.train_generator(batch)
        return d, g

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logs = step()
    sync()
    elapsed = time.perf_counter() - t0
    tokens = torch.tensor([float(batch.n_tokens * args.steps)], device=dev)
    tmax = torch.tensor([elapsed], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tokens, op=dist.ReduceOp.SUM)
    elapsed, tokens = float(tmax), float(tokens)

    roof = None
    if not args.no_roofline and rank == 0:
        ops.start_profile()
        step()
        roof = roofline(ops.stop_profile(), args.dtype)
    if world > 1:
        dist.barrier()
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(nets, cfg, batch)

    if rank == 0:
        out = {
            "metric": "train tokens/sec, MSSRNet WGAN step (12L d512, bs80)",
            "value": round(tokens / elapsed, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: full WGAN step = train_discriminator + train_generator, "
                                   "12L d512 ff1024 main model, 3L style generator/teacher, text+style critics with GP, "
                                   "bs%d sentences/GPU, <=%d words, V=%d, dropout 0.1, Adam" % (
                                       args.batch, args.max_words, cfg.vocab_size),
                       "tokens_per_step_per_gpu": batch.n_tokens,
                       "parallelism": "dp%d (batch-sharded, bucketed RCCL grad all-reduce)" % world,
                       "losses": {k: round(v, 4) for k, v in {**logs[0], **logs[1]}.items()}},
            "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
