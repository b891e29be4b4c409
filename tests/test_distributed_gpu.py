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


# --------------------------------------------------------------------------------------------------------------------
# RCCL itself on this one-GPU box: a process group of ONE rank over the nccl backend (= RCCL on ROCm) with
# DataParallel(force_collectives=True) -- RCCL init with device_id, the broadcast, the stage-hooked asynchronous
# all-reduces issued from inside the backward pass, the sync-BN and loss-count reductions all execute on the device; with
# one rank they are identities, so the result must equal the collective-free step bit for bit.
def _nccl_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd.distributed import DataParallel, FusedAdam
    from twog_gcn_amd.losses import select_loss
    crit, _ = select_loss('2G-GCN', 'multiple', 'mphoi', dict(misc={}))
    xh, xo, mask, tgt, noise = (t.to(DEV) for t in _batch(4))
    seg_t = torch.zeros(xh.shape[:3], device=DEV)
    res = {}
    for forced in (True, False):
        model = _tiny_model(seed=0).to(DEV).train()
        dp = DataParallel(model, bucket_mb=1, sync_bn=True, count_weighted_loss=True, global_noise_seed=11,
                          force_collectives=forced)
        opt = FusedAdam(dp.flat, lr=1e-2)
        grads = []
        for _ in range(3):
            dp.zero_grad()
            out = model(xh, xo, mask, human_segmentation=torch.ones(xh.shape[:3], device=DEV))
            with dp.loss_scope():
                loss = sum(crit(out, [seg_t, seg_t, tgt, tgt, tgt, tgt]))
            loss.backward()
            dp.all_reduce_gradients()
            grads.append(dp.flat.grad.clone().cpu())
            opt.step(dp.grad_scale)
        torch.cuda.synchronize()
        res[forced] = (grads, dp.flat.flat.clone().cpu(), dp.collective_calls)
        dp.close()
    ret['backend'] = dist.get_backend()
    ret['forced'], ret['plain'] = res[True], res[False]
    dist.destroy_process_group()


def test_rccl_single_rank_collective_path_on_the_device():
    port = 36500 + os.getpid() % 2000
    ret = mp.Manager().dict()
    mp.spawn(_nccl_worker, args=(1, port, ret), nprocs=1, join=True)
    assert ret['backend'] == 'nccl'
    (g_f, p_f, calls_f), (g_p, p_p, calls_p) = ret['forced'], ret['plain']
    # per step: 1 sync-BN reduction + 1 loss-count reduction + >= 3 gradient buckets (three stages), + the broadcast
    assert calls_f >= 1 + 3 * 5 and calls_p == 0, (calls_f, calls_p)
    for a, b in zip(g_f, g_p):
        assert torch.equal(a, b)                  # a one-rank all-reduce is the identity
    assert torch.equal(p_f, p_p)
    assert all(torch.isfinite(g).all() for g in g_f)


# --------------------------------------------------------------------------------------------------------------------
# Ordering of the first gradient all-reduce around the persistent backward launches (profiles/HISTORY.md section 6 / 8, VERDICT r04
# item 8): RCCL's kernels hold compute units until the peers arrive, a persistent launch needs all of them -- so the
# stage-0 all-reduce (heads + segment level) is issued BEHIND the persistent frame-level backward launch, and nothing
# is in flight when the persistent segment-level backward launch runs. One rank over RCCL with force_collectives=True
# executes exactly that path on this box; the log of (launch, collective) events must show the order, the persistent
# launches must really have run, and the gradients must equal the collective-free step's.
def _order_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd import kernels
    from twog_gcn_amd.distributed import DataParallel
    from twog_gcn_amd.models import TGGCN
    K = kernels.get_kernels()
    log = []
    for name in ('twog_bigru_bwd_persistent', 'twog_segrnn_bwd_persistent', 'twog_bigru_fwd_persistent', 'twog_segrnn_fwd_persistent'):
        real = getattr(K.lib, name)
        setattr(K.lib, name, (lambda real, name: lambda *a: (log.append(name), real(*a))[1])(real, name))
    real_ready = DataParallel._stage_ready

    def ready(self, stage):
        log.append(f'stage{stage}')
        return real_ready(self, stage)

    DataParallel._stage_ready = ready
    torch.manual_seed(0)
    res = {}
    xh, xo, mask, tgt, noise = (t.to(DEV) for t in _batch(4, T=6))
    for forced in (True, False):
        torch.manual_seed(0)
        model = TGGCN(input_size=(2048 + 4 * 26, 2048), num_classes=(13, None), hidden_size=64, gcn_node=26,
                      attention_style='v3', discrete_optimization_strategy='gs', message_segment=True, message_type='v2',
                      message_granularity='v1', message_aggregation='att', object_segment_update_strategy='ind').to(DEV).train()
        dp = DataParallel(model, bucket_mb=1, force_collectives=forced)
        model._gumbel_noise_override = noise
        del log[:]
        dp.zero_grad()
        out = model(xh, xo, mask, human_segmentation=torch.ones(xh.shape[:3], device=DEV))
        (torch.nn.functional.nll_loss(out[4], tgt) + torch.nn.functional.nll_loss(out[5], tgt)).backward()
        dp.all_reduce_gradients()
        torch.cuda.synchronize()
        res[forced] = (list(log), dp.flat.grad.clone().cpu(), dp.collective_calls)
        dp.close()
    ret['forced'], ret['plain'] = res[True], res[False]
    dist.destroy_process_group()


def test_first_gradient_all_reduce_is_issued_behind_the_persistent_backward_launches():
    port = 38500 + os.getpid() % 2000
    ret = mp.Manager().dict()
    mp.spawn(_order_worker, args=(1, port, ret), nprocs=1, join=True)
    (log_f, g_f, calls_f), (log_p, g_p, calls_p) = ret['forced'], ret['plain']
    for name in ('twog_bigru_fwd_persistent', 'twog_segrnn_fwd_persistent', 'twog_segrnn_bwd_persistent', 'twog_bigru_bwd_persistent'):
        assert name in log_f, (name, log_f)
    i_seg, i_gru, i_s0 = log_f.index('twog_segrnn_bwd_persistent'), log_f.index('twog_bigru_bwd_persistent'), log_f.index('stage0')
    assert i_seg < i_gru < i_s0, log_f            # no collective in flight under either persistent backward launch
    assert log_f.index('stage0') < log_f.index('stage1') < log_f.index('stage2'), log_f
    assert calls_f >= 3 and calls_p == 0
    assert torch.equal(g_f, g_p) and torch.isfinite(g_f).all()


# --------------------------------------------------------------------------------------------------------------------
# a persistent backward launch that gives up leaves incomplete gradients; its error word is read late.
# Under a data-parallel wrapper those gradients must not reach the other ranks' optimizers: a guard launch in front of
# each stage's all-reduce marks the stage NaN on the device when a pending word is set, the sum carries the mark to
# every rank, and all_reduce_gradients() raises on all of them (the failing rank does NOT raise inside backward, where its
# peers would be left waiting in a collective).
def _late_failure_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd import kernels
    from twog_gcn_amd.distributed import DataParallel
    from twog_gcn_amd.models import TGGCN
    K = kernels.get_kernels()
    torch.manual_seed(0)
    xh, xo, mask, tgt, noise = (t.to(DEV) for t in _batch(4, T=6))
    model = TGGCN(input_size=(2048 + 4 * 26, 2048), num_classes=(13, None), hidden_size=64, gcn_node=26,
                  attention_style='v3', discrete_optimization_strategy='gs', message_segment=True, message_type='v2',
                  message_granularity='v1', message_aggregation='att', object_segment_update_strategy='ind').to(DEV).train()
    dp = DataParallel(model, bucket_mb=1, force_collectives=True)
    model._gumbel_noise_override = noise
    seg = torch.ones(xh.shape[:3], device=DEV)

    def step(fail_backward):
        dp.zero_grad()
        out = model(xh, xo, mask, human_segmentation=seg)
        loss = torch.nn.functional.nll_loss(out[4], tgt) + torch.nn.functional.nll_loss(out[5], tgt)
        if fail_backward:
            os.environ['TWOG_PERSIST_SPIN_LIMIT'] = '0'   # the library's test hook: every wait gives up at once
        try:
            loss.backward()                                # must NOT raise: the peers are entering their collectives
        finally:
            os.environ.pop('TWOG_PERSIST_SPIN_LIMIT', None)
        dp.all_reduce_gradients()

    os.environ['TWOG_PERSIST_CHECK'] = 'lazy'              # the steady state: error words read at the end of the pass
    step(False)
    torch.cuda.synchronize()
    ret['clean_finite'] = bool(torch.isfinite(dp.flat.grad).all())
    ret['persistent_ran'] = bool(K.last_segrnn_bwd_persistent and K.last_bigru_bwd_persistent)
    try:
        step(True)
        ret['raised'] = None
    except RuntimeError as e:
        ret['raised'] = str(e)[:300]
    torch.cuda.synchronize()
    begins = [b for b, _ in dp.flat.stage_ranges.values()]
    ret['markers_nan'] = [bool(torch.isnan(dp.flat.grad[b])) for b in begins]
    step(False)                                            # and the step after it is clean again (launch-per-step path)
    torch.cuda.synchronize()
    ret['after_finite'] = bool(torch.isfinite(dp.flat.grad).all())
    dp.close()
    dist.destroy_process_group()


def test_a_persistent_backward_launch_that_gives_up_poisons_the_all_reduce_and_raises_after_it():
    port = 40500 + os.getpid() % 2000
    ret = mp.Manager().dict()
    mp.spawn(_late_failure_worker, args=(1, port, ret), nprocs=1, join=True)
    assert ret['clean_finite'] and ret['persistent_ran'], dict(ret)
    assert ret['raised'] and 'persistent launch' in ret['raised'], dict(ret)
    assert any(ret['markers_nan']), dict(ret)              # the stage(s) issued behind the failed launches carry the mark
    assert ret['after_finite'], dict(ret)
