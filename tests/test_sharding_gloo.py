"""Multi-rank path on CPU (gloo, world_size 2): the op is per-sample independent, so N GPUs simply
take N/world slices of the batch with NO data-path collective.  The only cross-rank step a trainer
needs is the ordinary gradient all-reduce of the [C, nD] weight gradient.  These tests run the same
sharding helper bench.py uses, on the CPU dispatch key, and check both facts:
  * concatenating the per-rank outputs / input-grads reproduces the single-process result bit for bit,
  * all-reducing the per-rank weight gradients reproduces the single-process weight gradient.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, nd, pad, active, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torchshifts  # noqa: F401
    from bench import shard_range
    torch.manual_seed(0)  # every rank builds the same global batch, then keeps its slice
    shape = {1: (6, 4, 20), 2: (6, 4, 9, 12), 3: (6, 4, 5, 6, 8)}[nd]
    x = torch.rand(shape, dtype=torch.float64)
    go = torch.rand(shape, dtype=torch.float64)
    w = (torch.rand(shape[1], nd, dtype=torch.float64) - 0.5) * 5
    lo, hi = shard_range(shape[0], rank, world)
    xs = x[lo:hi].clone().requires_grad_(True)
    ws = w.clone().requires_grad_(True)
    op = getattr(torch.ops.torchshifts, "shift%dd" % nd)
    out = op(xs, ws, torch.Tensor(), pad, active)
    out.backward(go[lo:hi])
    gw = ws.grad.clone()
    dist.all_reduce(gw)  # the trainer's gradient all-reduce (sum); not part of the op
    outs = [torch.empty_like(out) if r == rank else torch.empty((shard_range(shape[0], r, world)[1] -
            shard_range(shape[0], r, world)[0],) + tuple(shape[1:]), dtype=torch.float64) for r in range(world)]
    gxs = [torch.empty_like(o) for o in outs]
    dist.all_gather(outs, out.detach())      # only to compare; the op itself exchanged nothing
    dist.all_gather(gxs, xs.grad)
    if rank == 0:
        xf = x.clone().requires_grad_(True)
        wf = w.clone().requires_grad_(True)
        ref = op(xf, wf, torch.Tensor(), pad, active)
        ref.backward(go)
        ok = (torch.equal(torch.cat(outs), ref.detach()) and torch.equal(torch.cat(gxs), xf.grad) and
              torch.allclose(gw, wf.grad, rtol=1e-12, atol=1e-12))
        with open(tmp, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nd,pad,active", [(2, 0, False), (2, 3, True), (3, 2, True), (1, 4, False)])
def test_batch_shard_world2(tmp_path, nd, pad, active):
    tmp = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), nd, pad, active, tmp), nprocs=2, join=True)
    assert open(tmp).read() == "ok"


def test_shard_range_partitions():
    sys.path.insert(0, ROOT)
    from bench import shard_range
    for n in (1, 2, 7, 64, 513):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
