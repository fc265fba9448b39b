"""world_size-2 gloo test of the data-parallel gradient reduction (parallel.GradSync): reduced
grad == (sum over ranks) / world, for the plain path, the overlapped (armed) path, several
accumulating backwards before one reduction, and parameters that received no gradient."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    torch.manual_seed(5)
    return nn.Sequential(nn.Linear(16, 32), nn.ReLU(), nn.Linear(32, 32), nn.ReLU(), nn.Linear(32, 4),
                         nn.Linear(4, 4))        # last layer stays unused -> no grad


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MSSR_DP_CHECK"] = "1"                  # ranks must agree on which parameters have no gradient
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mssrnet_style_transfer_amd.parallel import GradSync
    res = {}
    for mode in ("plain", "armed"):
        net = _net()
        # tiny buckets -> several collectives; the two larger matrices are hooked, the rest waits for the final sync
        sync = GradSync(world, bucket_bytes=2048, hook_min_numel=256)
        torch.manual_seed(100 + rank)
        x1, x2 = torch.randn(8, 16), torch.randn(8, 16)
        body = net[:5]
        body(x1).pow(2).sum().backward()               # first backward accumulates locally
        if mode == "armed":
            sync.arm(net)
        body(x2).sum().backward()                      # last backward (overlapped when armed)
        local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
        early = len(sync._pending)                     # collectives already launched from gradient hooks
        assert early == (2 if mode == "armed" else 0), (mode, early)
        sync([net])
        reduced = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
        # a parameter unused in this phase must not be touched by the optimizer (torch's Adam skips grad None; a
        # zero gradient would still decay its moments and advance its step count): seed non-zero moments first
        opt = torch.optim.Adam(net.parameters(), lr=1e-2)
        for p in net[5].parameters():
            opt.state[p] = dict(step=torch.tensor(3.0), exp_avg=torch.full_like(p, 0.5), exp_avg_sq=torch.full_like(p, 0.25))
        opt.step()
        res[mode] = (local, reduced, [p.detach().clone() for p in net.parameters()])
    torch.save(res, os.path.join(out, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_grad_sync_two_ranks(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % i), weights_only=True) for i in range(world)]
    for mode in ("plain", "armed"):
        n = len(r[0][mode][0])
        for i in range(n):
            locs = [r[k][mode][0][i] for k in range(world)]
            if locs[0] is None:
                for k in range(world):      # never received a gradient: none after the reduction either
                    assert r[k][mode][1][i] is None, (mode, i, k)
                continue
            want = sum(locs) / world
            for k in range(world):
                assert torch.allclose(r[k][mode][1][i], want, rtol=1e-6, atol=1e-7), (mode, i, k)


    # world = 2 against world = 1 on the mean gradients: same parameters after Adam, the unused layer unmoved
    for mode in ("plain", "armed"):
        net = _net()
        before = [p.detach().clone() for p in net.parameters()]
        for i, p in enumerate(net.parameters()):
            g = r[0][mode][1][i]
            p.grad = None if g is None else g.clone()
        opt = torch.optim.Adam(net.parameters(), lr=1e-2)
        for p in net[5].parameters():
            opt.state[p] = dict(step=torch.tensor(3.0), exp_avg=torch.full_like(p, 0.5), exp_avg_sq=torch.full_like(p, 0.25))
        opt.step()
        for i, p in enumerate(net.parameters()):
            for k in range(world):
                assert torch.equal(r[k][mode][2][i], p.detach()), (mode, i, k)
        for p, b in zip(list(net.parameters())[-2:], before[-2:]):
            assert torch.equal(p.detach(), b)


def test_single_rank_is_a_noop():
    from mssrnet_style_transfer_amd.parallel import GradSync
    net = _net()
    net[:5](torch.randn(3, 16)).sum().backward()
    before = [p.grad.clone() for p in net[:5].parameters()]
    s = GradSync(1)
    s.arm(net)
    s([net])
    for a, p in zip(before, net[:5].parameters()):
        assert torch.equal(a, p.grad)
