"""Single-rank breakdown of the data-parallel path: Trainer.train_wgan_step with
GradSync(always=True) on ONE GPU -- the RCCL all-reduce is then an identity copy -- against the same step without
synchronisation, and variants that take the path apart (hooks, bucket gather, side-stream waits, the collective).
What it says about N > 1: everything but "the collective itself" is bookkeeping a real job pays as well.
usage: python tools/dp_overhead.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from mssrnet_style_transfer_amd import ops, parallel
from mssrnet_style_transfer_amd.config import MSSRConfig
from mssrnet_style_transfer_amd.parallel import GradSync
from mssrnet_style_transfer_amd.style_generator_builder import build_all
from mssrnet_style_transfer_amd.synth import make_batch
from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
ops.set_compute_dtype(torch.bfloat16)
cfg = MSSRConfig()
batch = make_batch(80, 16, cfg.vocab_size, seed=123, device=dev)
orig_fill, orig_launch = GradSync._fill, GradSync._launch

def launch_nocoll(self, b):
    side = self._side_stream(b["flat"].device)
    side.wait_stream(torch.cuda.current_stream())
    b["launched"] = True
    self._pending.append((b, None))

def launch_samestream(self, b):
    work = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
    b["launched"] = True
    self._pending.append((b, work))

def launch_nothing(self, b):
    b["launched"] = True
    self._pending.append((b, None))

def run(tag, mk, fill=None, launch=None):
    GradSync._fill = fill or orig_fill
    GradSync._launch = launch or orig_launch
    nets = build_all(cfg, device=dev, seed=123)
    tr = Trainer(nets, cfg, grad_sync=mk())
    for _ in range(4): tr.train_wgan_step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(15): tr.train_wgan_step(batch)
    torch.cuda.synchronize()
    print("%-44s %.2f ms/step" % (tag, (time.perf_counter() - t0) / 15 * 1e3), flush=True)

from mssrnet_style_transfer_amd import _lib as L
run("no sync (sweeps replayed as hipGraphs)", lambda: None)
L.load().mssr_decode_graphs_enable(0)
run("no sync, sweeps launched eagerly", lambda: None)
L.load().mssr_decode_graphs_enable(1)


def replayed():
    g = GradSync(1, always=True)
    L.load().mssr_decode_graphs_enable(1)       # GradSync turns replay off; this variant shows why
    return g


run("full, sweeps replayed (the setting GradSync avoids)", replayed)
# every variant below: as GradSync runs it -- sweeps launched eagerly
run("full", lambda: GradSync(1, always=True))
run("no fill", lambda: GradSync(1, always=True), fill=lambda self, b: None)
run("no fill, no collective (side waits only)", lambda: GradSync(1, always=True), fill=lambda self, b: None, launch=launch_nocoll)
run("no fill, nothing launched", lambda: GradSync(1, always=True), fill=lambda self, b: None, launch=launch_nothing)
run("fill, nothing launched", lambda: GradSync(1, always=True), launch=launch_nothing)
run("fill + collective on the current stream", lambda: GradSync(1, always=True), launch=launch_samestream)
run("no hooks, fill + coll on current stream", lambda: GradSync(1, always=True, hook_min_numel=1 << 40), launch=launch_samestream)
dist.destroy_process_group()
