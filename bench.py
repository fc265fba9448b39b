#!/usr/bin/env python
"""Headline benchmark: train tokens/sec of the MSSRNet WGAN step (12L d512 ff1024, bs80/GPU).

One "step" = train_discriminator(batch) + train_generator(batch) on the same synthetic
Yelp-shaped batch (BASELINE.json configs[1]), README flags, dropout on,
Adam steps included.  tokens = non-pad target tokens of the batch (the reference's own
normaliser, onmt/trainer_wgan_wh.py:389-391).  N > 1: one process per GPU under
torch.distributed.run, batch-sharded (bs80 per GPU, weak scaling) with bucketed RCCL gradient
all-reduce.  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 20 --warmup 5

--workload selects the other measurement cases of SURVEY.md section 8d (same JSON contract, their own metric
string; the headline metric is the default `wgan`): `reconstruct` (train_reconstruct phase), `cycle` (the
README schedule 25 reconstruct : 5 discriminator : 1 generator as one step), `teacher` (configs[0]: 3-forward
teacher step, bs16), `greedy` (configs[3]: greedy decode bs512, capped at --max-length steps because
random-init weights never emit </s>), `long` (configs[4]: WGAN step at 128 words, bs256).
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="wgan", choices=["wgan", "reconstruct", "cycle", "teacher", "greedy", "long"])
    ap.add_argument("--max-length", type=int, default=20, help="greedy workload: decode steps per sentence")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--max-words", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dry-run-gloo", action="store_true",
                    help="launcher rehearsal on CPU: start the ranks, join a gloo group, all-reduce a rank census, "
                         "print it and exit (no GPU, no model)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no job around it (WORLD_SIZE unset): start the N ranks ourselves, one
    process per GPU, as the reference's launcher does (onmt/bin/train_wgan_wh.py:67-97 spawns one process per GPU;
    onmt/utils/distributed.py:20-32 joins them over tcp://master).  This parent never initialises the GPU
    (torch.cuda.device_count() does not on this image) and never exec()s: the ranks are fresh children of
    `python -m torch.distributed.run`, rank 0's JSON line is relayed, a failing child makes the parent fail."""
    import socket
    import subprocess
    n = args.gpus
    if not args.dry_run_gloo:
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s); refusing to report a %d-GPU line from "
                             "fewer ranks" % (n, have, n))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.lstrip().startswith("{"):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0:
        raise SystemExit("bench.py: the %d-rank job exited with status %d" % (n, rc))
    if line is None:
        raise SystemExit("bench.py: the %d-rank job printed no result line" % n)
    print(line)


def dry_run_gloo(args):
    """Launcher rehearsal (CPU): every rank joins a gloo group and adds itself to a census."""
    import torch.distributed as dist
    from mssrnet_style_transfer_amd.parallel import init_distributed
    rank, local_rank, world = init_distributed(backend="gloo")
    census = torch.zeros(max(world, 1))
    census[rank] = 1.0
    if world > 1:
        dist.all_reduce(census)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "joined": int(census.sum()),
                          "rccl": {"backend": dist.get_backend() if world > 1 else None,
                                   "world_size": dist.get_world_size() if world > 1 else 1}}))
    if world > 1:
        dist.destroy_process_group()


def roofline(records, dtype, workload="wgan"):
    """Dominant GEMM kernel instance of one instrumented step: algorithmic bytes (or flops) of all its
    launches / the sum of their durations.  Durations are HIP events handed to hipExtLaunchKernelGGL
    (mssr_prof_*, csrc/core.hip): begin / end of the dispatch itself on the launch stream, the interval
    rocprofv3's kernel trace reports."""
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
    traffic, src, stale = pmc_traffic(name, workload)
    return dict(bound=bound, achieved=round(ach, 2), peak=peak, unit=unit, frac=round(ach / peak, 5), traffic=traffic,
                traffic_source=src, traffic_stale=stale,
                kernel=name, launches_per_step=a["launches"], avg_launch_us=round(a["ms"] * 1e3 / a["launches"], 2),
                algorithmic_bytes_per_launch=int(a["bytes"] / a["launches"]),
                gemm_time_share={k: round(v["ms"], 3) for k, v in sorted(agg.items())})


def algorithmic(wl, args, world, sec_per_step, dtype):
    """SURVEY 8d's algorithmic work per step -- GEMM FLOPs (2*M*N*K) of the REFERENCE path on this batch shape, counted
    once on the reference trainer with torch.utils.flop_counter's formulas (tests/golden/flops.json, generated by
    tests/golden/make_golden.py flops) -- over the measured step time, against the dense MFMA peak."""
    if wl not in ("wgan", "reconstruct", "cycle") or args.batch != 80 or args.max_words != 16:
        return None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "flops.json")
    with open(path) as f:
        c = json.load(f)["configs1_b80_w16"]
    d, g, r = (c[k]["reference"] for k in ("train_discriminator", "train_generator", "train_reconstruct"))
    flops = {"wgan": d + g, "reconstruct": r, "cycle": 25 * r + 5 * d + g}[wl] * world
    key = "bf16" if dtype == "bf16" else "f32"
    tf = flops / sec_per_step / 1e12
    return dict(gemm_gflop_per_step=round(flops / 1e9, 1), tflops=round(tf, 2), mfma_peak_tflops=MFMA_PEAK_TFLOPS[key],
                frac=round(tf / MFMA_PEAK_TFLOPS[key], 5),
                source="reference path under torch.utils.flop_counter formulas (tests/golden/flops.json)")


def cpu_baseline(nets, cfg, batch, warmup=1, steps=3):
    """The oracle (a CPU port of the reference path, oracle/) timed on this box's host cores on full
    WGAN steps (D phase + G phase, Adam included) of the same batch: `warmup` untimed steps, then the
    mean of `steps` timed ones (>= 1 warm-up + >= 3 timed).  Reported, not a target."""
    from types import SimpleNamespace
    from oracle import phases as PH
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    states = {k: {n: v.detach().float().cpu() for n, v in m.state_dict().items()} for k, m in nets.items()}
    onets = PH.make_nets(states, cfg)
    b = SimpleNamespace(src=batch.src.cpu(), src_lengths=batch.src_lengths.cpu(), tgt=batch.tgt.cpu(),
                        labels=batch.labels.cpu())
    times = []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        PH.train_discriminator(onets, b, cfg)
        PH.train_generator(onets, b, cfg)
        times.append(time.perf_counter() - t0)
    timed = times[warmup:]
    dt = sum(timed) / len(timed)
    return dict(value=round(batch.n_tokens / dt, 2), unit="tokens/s", cores=threads, kind="port",
                sample="%d warm-up + %d timed full WGAN steps (D+G phase, Adam) of the same bs%d batch, fp32, "
                       "torch CPU ops on %d threads; %.2f s per step (min %.2f, max %.2f), warm-up step %.2f s" % (
                           warmup, steps, batch.batch_size, threads, dt, min(timed), max(timed),
                           times[0] if warmup else float("nan")))


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_pmc_traffic.json, written by tools/pmc_traffic.py from the FETCH_SIZE and WRITE_SIZE passes --
    separate runs, FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes).  PMC counters cannot be read
    from inside the process, so the figure is the profiled one of the newest round on file; None when no summary
    covers this kernel / workload.  The summary records a hash of the sources it was collected on
    (tools/pmc_traffic.py source_sha16); when the running tree differs, the figure is withheld (traffic = None) and
    `traffic_stale` says which file holds the outdated one."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_traffic import source_sha16
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        with open(path) as f:
            t = json.load(f)
        k = t.get("workloads", {}).get(workload, {}).get(kernel)
        if k:
            if t.get("source_sha16", {}).get(workload) != source_sha16(ROOT):
                return None, os.path.basename(path), True
            return int(k["traffic_bytes_per_launch"]), os.path.basename(path), False
    return None, None, None


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)                 # before anything touches the GPU
    if args.dry_run_gloo:
        return dry_run_gloo(args)
    from mssrnet_style_transfer_amd import ops
    from mssrnet_style_transfer_amd.config import MSSRConfig
    from mssrnet_style_transfer_amd.parallel import GradSync, init_distributed
    from mssrnet_style_transfer_amd.style_generator_builder import build_all
    from mssrnet_style_transfer_amd.synth import make_batch
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    import torch.distributed as dist

    rank, local_rank, world = init_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no CPU fallback exists for the product path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ops.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)

    cfg = MSSRConfig()
    nets = build_all(cfg, device=dev, seed=123)               # same weights on every rank
    wl = args.workload
    dflt_batch = {"teacher": 16, "greedy": 512, "long": 256}.get(wl, 80)
    dflt_words = 128 if wl == "long" else 16
    args.batch = args.batch or dflt_batch
    args.max_words = args.max_words or dflt_words
    batch = make_batch(args.batch, args.max_words, cfg.vocab_size, seed=123 + rank, device=dev,
                       min_words=64 if wl == "long" else 3)
    torch.manual_seed(1000 + rank)
    trainer = Trainer(nets, cfg, grad_sync=GradSync(world) if world > 1 else None)
    tokens_per_step = batch.n_tokens
    base = "12L d512 ff1024 main model, 3L style generator/teacher, text+style critics with GP, " \
           "bs%d sentences/GPU, <=%d words, V=%d, dropout 0.1, Adam" % (args.batch, args.max_words, cfg.vocab_size)

    if wl in ("wgan", "long"):
        metric = "train tokens/sec, MSSRNet WGAN step (12L d512, bs%d)" % args.batch
        what = ("configs[1]" if wl == "wgan" else "configs[4] (long-sequence stress)") + \
            ": full WGAN step = train_discriminator + train_generator, " + base

        def step():
            # = train_discriminator(batch) then train_generator(batch); the two phases' soft decodes share one sweep
            return trainer.train_wgan_step(batch)
    elif wl == "reconstruct":
        metric = "train tokens/sec, MSSRNet train_reconstruct phase (12L d512, bs%d)" % args.batch
        what = "train_reconstruct phase alone (SURVEY 8d), " + base

        def step():
            return (trainer.train_reconstruct(batch), {})
    elif wl == "cycle":
        metric = "train tokens/sec, MSSRNet schedule cycle 25 reconstruct : 5 discriminator : 1 generator (bs%d)" % args.batch
        what = "one step = the README schedule cycle (31 phase calls on the same batch), " + base
        tokens_per_step = batch.n_tokens * 31

        def step():
            for _ in range(25):
                r = trainer.train_reconstruct(batch)
            for _ in range(5):
                d = trainer.train_discriminator(batch)
            g = trainer.train_generator(batch)
            return ({**r, **d}, g)
    elif wl == "teacher":
        from mssrnet_style_transfer_amd.trainer_wgan_wh import teacher_step_loss
        from mssrnet_style_transfer_amd.style_generator_builder import build_classifier
        torch.manual_seed(123)
        clf = build_classifier(cfg).to(dev)
        clf.train()
        from mssrnet_style_transfer_amd.optim import Adam
        t_opt = Adam(clf)                       # torch.optim.Adam's defaults, as onmt/trainer_teacher.py:220
        metric = "train tokens/sec, teacher classifier step (3L 8x32, bs%d)" % args.batch
        what = "configs[0]: teacher step = 3 forwards (ids / ids+</s> / padded length) + backward + Adam, " \
               "bs%d, <=%d words" % (args.batch, args.max_words)

        def step():
            t_opt.zero_grad()
            loss, _ = teacher_step_loss(clf, batch, cfg)
            loss.backward()
            t_opt.step()
            return ({"teacher_nll": float(loss.detach())}, {})
    else:
        from mssrnet_style_transfer_amd.translate import greedy_translate
        desired = 1 - batch.labels
        tokens_per_step = args.batch * args.max_length
        metric = "decode tokens/sec, greedy style transfer (beam 1, bs%d)" % args.batch
        what = "configs[3]: StyleGenerator2 + encoder + bond + greedy decode on the decode engine, bs%d, <=%d words, " \
               "%d steps per sentence (random-init weights never emit </s>: fixed cap), ids copied to the host" % (
                   args.batch, args.max_words, args.max_length)

        def step():
            out = greedy_translate(nets["model"], nets["style_generator"], cfg, batch.src[1:], batch.src_lengths - 1,
                                   desired, max_length=args.max_length)
            return ({"mean_len": sum(len(o) for o in out) / len(out)}, {})

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
    tokens = torch.tensor([float(tokens_per_step * args.steps)], device=dev)
    tmax = torch.tensor([elapsed], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tokens, op=dist.ReduceOp.SUM)
    elapsed, tokens = float(tmax), float(tokens)

    roof = None
    if not args.no_roofline and rank == 0:
        ops.start_profile()
        step()
        roof = roofline(ops.stop_profile(), args.dtype, wl)
    if world > 1:
        dist.barrier()
    cpu = None
    if rank == 0 and world == 1 and wl == "wgan" and not args.no_cpu_baseline:
        cpu = cpu_baseline(nets, cfg, batch)

    if rank == 0:
        out = {
            "metric": metric,
            "value": round(tokens / elapsed, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": what,
                       "tokens_per_step_per_gpu": tokens_per_step,
                       "parallelism": "dp%d (batch-sharded, bucketed RCCL grad all-reduce)" % world,
                       "losses": {k: round(v, 4) for k, v in {**logs[0], **logs[1]}.items()}},
            "rccl": {"backend": dist.get_backend() if world > 1 else None,
                     "world_size": dist.get_world_size() if world > 1 else 1},
            "roofline": roof, "algorithmic": algorithmic(wl, args, world, elapsed / args.steps, args.dtype),
            "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
