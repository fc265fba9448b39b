"""Data-parallel WGAN step (Trainer.train_wgan_step, as bench.py --gpus N runs it) on the GPU kernels: two ranks (both on cuda:0, gloo transport, so it runs on a
one-GPU box), each with its own batch; after the phase every rank's gradients must equal the mean over ranks
of the gradients a single process computes for the same two batches (SUM then / world)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank_seed):
    from types import SimpleNamespace
    from mssrnet_style_transfer_amd import ops
    from mssrnet_style_transfer_amd.config import MSSRConfig
    from mssrnet_style_transfer_amd.style_generator_builder import build_all
    from mssrnet_style_transfer_amd.synth import make_batch
    ops.set_compute_dtype(torch.float32)
    cfg = MSSRConfig.small(learning_rate=0.0)            # lr 0: the phases leave the weights alone
    nets = build_all(cfg, device="cuda:0", seed=42)
    b = make_batch(6, 7, cfg.vocab_size, seed=rank_seed, device="cuda:0")
    return cfg, nets, b


def _grads(nets, names):
    return {n: {k: (None if p.grad is None else p.grad.detach().cpu().clone()) for k, p in nets[n].named_parameters()}
            for n in names}


def _wgan_step(tr, nets, b):
    """One Trainer.train_wgan_step -- the call `bench.py --gpus N` runs on every rank (shared decode sweep) -- ->
    {"D": critic gradients as the D half left them, "G": generator-side gradients}."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import wgan_step_with_capture
    assert tr.cfg.share_decode_sweep and tr.cfg.batch_generator_phase and tr.cfg.ap_gen_self_style_text
    torch.manual_seed(7)
    _, _, d_grads = wgan_step_with_capture(tr, b, nets)
    return {"D": {n: {k: (None if g is None else g.cpu()) for k, g in d.items()} for n, d in d_grads.items()},
            "G": _grads(nets, ("model", "style_generator"))}


def _worker(rank, world, port, out):
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=True)        # a stuck rank names its stack instead of hanging the run
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mssrnet_style_transfer_amd.parallel import GradSync
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    cfg, nets, b = _setup(100 + rank)
    tr = Trainer(nets, cfg, grad_sync=GradSync(world, bucket_bytes=1 << 16))
    res = _wgan_step(tr, nets, b)
    torch.save(res, os.path.join(out, "rank%d.pt" % rank))
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


def test_data_parallel_phases_match_mean_of_single_process_gradients(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % i), weights_only=True) for i in range(world)]
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    ref = []
    for r in range(world):
        cfg, nets, b = _setup(100 + r)
        tr = Trainer(nets, cfg)
        ref.append(_wgan_step(tr, nets, b))
    checked = 0
    for phase in ("D", "G"):
        for net in ref[0][phase]:
            # key-projection biases have a mathematically zero gradient (softmax shift invariance): what is
            # left there is rounding noise, so the absolute floor is set from the net's largest gradient
            floor = 1e-6 * max(float(g.abs().max()) for r in range(world) for g in ref[r][phase][net].values()
                               if g is not None)
            for k in ref[0][phase][net]:
                parts = [ref[r][phase][net][k] for r in range(world)]
                if any(p is None for p in parts):
                    continue
                want = sum(parts) / world
                scale = float(want.abs().max())
                for r in range(world):
                    g = got[r][phase][net][k]
                    assert g is not None, (phase, net, k)
                    assert float((g - want).abs().max()) <= 1e-5 * scale + floor, (phase, net, k, r)
                    checked += 1
    assert checked > 100


def _nccl_worker(rank, port, out):
    import faulthandler
    faulthandler.dump_traceback_later(120, exit=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from mssrnet_style_transfer_amd.parallel import GradSync
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    cfg, nets, b = _setup(100)
    sync = GradSync(1, bucket_bytes=1 << 16, always=True)
    tr = Trainer(nets, cfg, grad_sync=sync)
    assert sync.active
    res = {}
    res = _wgan_step(tr, nets, b)
    for step in range(2):                                  # more steps back to back: no host sync in between
        torch.manual_seed(7)
        d, g = tr.train_wgan_step(b)
    res["logs"] = {**d, **g}
    torch.cuda.synchronize()
    torch.save(res, os.path.join(out, "nccl.pt"))
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


def test_rccl_path_single_rank_identity_reduction(tmp_path):
    """The collectives exactly as the multi-GPU job issues them (RCCL all-reduce per bucket from gradient hooks on
    the side stream, waited for before clipping), on ONE rank where the reduction is the identity: gradients must
    equal a run without GradSync, and three steps back to back must neither stall nor diverge."""
    mp.spawn(_nccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    got = torch.load(os.path.join(str(tmp_path), "nccl.pt"), weights_only=True)
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    cfg, nets, b = _setup(100)
    tr = Trainer(nets, cfg)
    one = _wgan_step(tr, nets, b)
    n = 0
    for phase, ref in (("D", one["D"]), ("G", one["G"])):
        for net in ref:
            # float atomics (embedding / column-sum gradients) make two runs differ in the last bits
            floor = 1e-6 * max(float(g.abs().max()) for g in ref[net].values() if g is not None)
            for k, want in ref[net].items():
                if want is None:
                    continue
                g = got[phase][net][k]
                assert float((g - want).abs().max()) <= 1e-5 * float(want.abs().max()) + floor, (phase, net, k)
                n += 1
    assert n > 100 and all(v == v for v in got["logs"].values())
