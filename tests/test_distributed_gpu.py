"""GPU: two ranks sharing the one visible MI355X (gloo for the collectives, HIP kernels for everything else) run a
data-parallel step with gradient all-reduces started from inside the backward pass; the reduced gradients equal the
single-process full-batch gradients. (8-GPU RCCL runs are the driver's; this covers the same code path end to end on
the device: flat buffers, stage hooks, in-place gradient sinks, hipGraph chain replay under two processes.)"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_distributed_cpu import _tiny_model, _batch, ROOT

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _step(model, dp, xh, xo, mask, tgt, noise):
    model._gumbel_noise_override = noise
    model.eval()
    dp.zero_grad()
    out = model(xh, xo, mask, human_segmentation=torch.ones(xh.shape[:3], device=xh.device))
    (torch.nn.functional.nll_loss(out[4], tgt) + torch.nn.functional.nll_loss(out[5], tgt)).backward()
    dp.all_reduce_gradients()
    return dp.flat.grad.clone() * dp.grad_scale


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd.distributed import DataParallel
    torch.cuda.set_device(0)
    model = _tiny_model(seed=rank).to(DEV)
    dp = DataParallel(model, bucket_mb=1)
    xh, xo, mask, tgt, noise = (t.to(DEV) for t in _batch(4))
    sl = slice(rank * 2, rank * 2 + 2)
    grads = [_step(model, dp, xh[sl], xo[sl], mask[sl], tgt[sl], noise[:, sl]).cpu() for _ in range(3)]
    ret[rank] = grads  # three identical steps: the 2nd / 3rd replay the captured hipGraphs
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_process():
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd.distributed import DataParallel
    port = 33500 + os.getpid() % 2000
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    model = _tiny_model(seed=0).to(DEV)
    dp = DataParallel(model)
    xh, xo, mask, tgt, noise = (t.to(DEV) for t in _batch(4))
    ref = _step(model, dp, xh, xo, mask, tgt, noise).cpu()
    scale = max(1.0, float(ref.abs().max()))
    for r in (0, 1):
        for g in ret[r]:
            assert float((g - ref).abs().max()) < 2e-5 * scale
        assert torch.equal(ret[r][0], ret[r][1]) and torch.equal(ret[r][1], ret[r][2])  # deterministic, graphs included
    assert torch.equal(ret[0][0], ret[1][0])
