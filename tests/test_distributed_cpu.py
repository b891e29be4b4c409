"""CPU, 2 processes over gloo: the data-parallel wrapper (flat parameter/gradient buffers, chunked gradient all-reduce,
1/W folded into the fused Adam) gives every rank the SAME parameters as a single process that sees the whole batch,
for a model whose loss is a mean over clips. Uses the torch test double of the kernel interface (no GPU here)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny_model(seed=0):
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd.models import TGGCN
    torch.manual_seed(seed)
    return TGGCN(input_size=(2048 + 4 * 26, 2048), num_classes=(13, None), hidden_size=8, gcn_node=26,
                 attention_style='v3', discrete_optimization_strategy='gs', message_segment=True, message_type='v2',
                 message_granularity='v1', message_aggregation='att', object_segment_update_strategy='ind')


def _batch(bs, T=4, H=2, O=3, N=26, seed=1):
    g = torch.Generator().manual_seed(seed)
    xh = torch.rand(bs, T, H, 2048 + 4 * N, generator=g)
    xo = torch.rand(bs, T, O, 2048, generator=g)
    mask = torch.ones(bs, O)
    tgt = torch.randint(0, 13, (bs, T, H), generator=g)
    torch.manual_seed(1234)  # the Gumbel draw uses the global generator: make it identical in every process
    noise = torch.distributions.gumbel.Gumbel(0.0, 1.0).sample((T * O, bs, 2))
    return xh, xo, mask, tgt, noise


def _loss(model, xh, xo, mask, tgt, noise):
    model._gumbel_noise_override = noise
    model.eval()  # running-stat BatchNorm: no cross-clip coupling, so the sharded and the full batch are comparable
    out = model(xh, xo, mask, human_segmentation=torch.ones(xh.shape[:3]))
    return torch.nn.functional.nll_loss(out[4], tgt) + torch.nn.functional.nll_loss(out[5], tgt)


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd import kernels
    from twog_gcn_amd.distributed import DataParallel, FusedAdam
    from tests.fake_kernels import FakeKernels
    kernels._set_backend_for_tests(FakeKernels())
    torch.set_num_threads(2)
    model = _tiny_model(seed=rank)  # different init per rank: the wrapper must broadcast rank 0's parameters
    dp = DataParallel(model, bucket_mb=1)
    opt = FusedAdam(dp.flat, lr=1e-2)
    xh, xo, mask, tgt, noise = _batch(4)
    sl = slice(rank * 2, rank * 2 + 2)
    init = dp.flat.flat.clone()
    dp.zero_grad()
    _loss(model, xh[sl], xo[sl], mask[sl], tgt[sl], noise[:, sl]).backward()
    dp.all_reduce_gradients()
    grad = dp.flat.grad.clone() * dp.grad_scale
    opt.step(dp.grad_scale)
    ret[rank] = (init, grad, dp.flat.flat.clone())
    dist.destroy_process_group()


def test_two_rank_data_parallel_matches_single_process():
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd import kernels
    from twog_gcn_amd.distributed import DataParallel, FusedAdam
    from tests.fake_kernels import FakeKernels
    port = 29500 + os.getpid() % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    for a, b in zip(ret[0], ret[1]):
        assert torch.equal(a, b), 'ranks diverged'  # same broadcast init, same reduced gradient, same update
    # single-process reference on the full batch
    kernels._set_backend_for_tests(FakeKernels())
    try:
        model = _tiny_model(seed=0)
        dp = DataParallel(model)
        opt = FusedAdam(dp.flat, lr=1e-2)
        xh, xo, mask, tgt, noise = _batch(4)
        assert torch.equal(ret[0][0], dp.flat.flat), 'rank-0 parameters were not broadcast'
        dp.zero_grad()
        _loss(model, xh, xo, mask, tgt, noise).backward()
        ref = dp.flat.grad
        err = (ret[0][1] - ref).abs().max().item()
        assert err < 1e-5 * max(1.0, ref.abs().max().item()), err  # mean of shard means == mean over the full batch
        assert not torch.equal(ret[0][2], ret[0][0])  # the fused Adam step moved the parameters
    finally:
        kernels._set_backend_for_tests(None)


# --------------------------------------------------------------------------------------------------------------------
# exact equivalence mode: train-mode BatchNorm over the GLOBAL batch + global Gumbel noise
def _loss_train(model, xh, xo, mask, tgt):
    model.train()
    out = model(xh, xo, mask)   # learned gates: Gumbel noise is drawn inside forward
    return torch.nn.functional.nll_loss(out[4], tgt) + torch.nn.functional.nll_loss(out[5], tgt)


def _worker_sync(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd import kernels
    from twog_gcn_amd.distributed import DataParallel
    from tests.fake_kernels import FakeKernels
    kernels._set_backend_for_tests(FakeKernels())
    torch.set_num_threads(2)
    model = _tiny_model(seed=0)
    dp = DataParallel(model, bucket_mb=1, sync_bn=True, global_noise_seed=77)
    xh, xo, mask, tgt, _ = _batch(4)
    sl = slice(rank * 2, rank * 2 + 2)
    dp.zero_grad()
    _loss_train(model, xh[sl], xo[sl], mask[sl], tgt[sl]).backward()
    dp.all_reduce_gradients()
    bn = model.geometry_embedding_gcn.joint_embed.cnn[0].bn
    ret[rank] = (dp.flat.grad.clone() * dp.grad_scale, bn.running_mean.clone(), bn.running_var.clone())
    dist.destroy_process_group()


def test_sync_bn_and_global_noise_reproduce_the_full_batch_step():
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd import kernels
    from twog_gcn_amd.distributed import DataParallel
    from tests.fake_kernels import FakeKernels
    port = 31500 + os.getpid() % 2000
    ret = mp.Manager().dict()
    mp.spawn(_worker_sync, args=(2, port, ret), nprocs=2, join=True)
    kernels._set_backend_for_tests(FakeKernels())
    try:
        model = _tiny_model(seed=0)
        dp = DataParallel(model, sync_bn=True, global_noise_seed=77)   # world 1: same global noise, plain batch stats
        xh, xo, mask, tgt, _ = _batch(4)
        dp.zero_grad()
        _loss_train(model, xh, xo, mask, tgt).backward()
        ref = dp.flat.grad
        bn = model.geometry_embedding_gcn.joint_embed.cnn[0].bn
        for r in (0, 1):
            err = (ret[r][0] - ref).abs().max().item()
            assert err < 2e-5 * max(1.0, ref.abs().max().item()), (r, err)
            assert torch.allclose(ret[r][1], bn.running_mean, rtol=1e-5, atol=1e-7)
            assert torch.allclose(ret[r][2], bn.running_var, rtol=1e-5, atol=1e-7)
    finally:
        kernels._set_backend_for_tests(None)


# --------------------------------------------------------------------------------------------------------------------
# count-weighted loss normalisation: ragged clips (-1 targets), every term of the reference's criterion on
def _ragged_targets(bs, T=4, H=2, seed=3):
    g = torch.Generator().manual_seed(seed)
    lengths = [T, 1, 2, T][:bs]                     # rank 0 gets clips of 4 + 1 valid frames, rank 1 of 2 + 4
    cls = torch.randint(0, 13, (bs, T, H), generator=g)
    seg = (torch.rand(bs, T, H, generator=g) < 0.5).float()
    for b, n in enumerate(lengths):
        cls[b, n:] = -1
        seg[b, n:] = -1.0
    return cls, seg


def _criterion_loss(model, xh, xo, mask, cls, seg, noise):
    """The product criterion (losses.select_loss: budget + BCE on the learned human gates, four NLL terms), all weights on."""
    from twog_gcn_amd.losses import select_loss
    crit, _ = select_loss('2G-GCN', 'multiple', 'mphoi', dict(misc=dict(budget_loss=dict(add=True, human_weight=0.7),
                                                                          segmentation_loss=dict(add=True, weight=1.3),
                                                                          first_level_loss_weight=0.5)))
    model._gumbel_noise_override = noise
    model.eval()
    out = model(xh, xo, mask)                       # human AND object gates learned
    return sum(crit(out, [seg, seg, cls, cls, cls, cls]))


def _noise_all(bs, T=4, H=2, O=3):
    torch.manual_seed(4321)
    return torch.distributions.gumbel.Gumbel(0.0, 1.0).sample((T * (H + O), bs, 2))


def _worker_counts(rank, world, port, ret, weighted):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd import kernels
    from twog_gcn_amd.distributed import DataParallel
    from tests.fake_kernels import FakeKernels
    kernels._set_backend_for_tests(FakeKernels())
    torch.set_num_threads(2)
    model = _tiny_model(seed=0)
    dp = DataParallel(model, bucket_mb=1, count_weighted_loss=weighted)
    xh, xo, mask, _, _ = _batch(4)
    cls, seg = _ragged_targets(4)
    noise = _noise_all(4)
    sl = slice(rank * 2, rank * 2 + 2)
    dp.zero_grad()
    with dp.loss_scope():   # the count reduction is a collective: active only inside the step's scope
        loss = _criterion_loss(model, xh[sl], xo[sl], mask[sl], cls[sl], seg[sl], noise[:, sl])
    calls_in_scope = dp.collective_calls
    # a criterion call OUTSIDE the scope (rank-0-only validation, a second model) or under no_grad issues no collective
    from twog_gcn_amd import losses as _losses
    assert _losses.get_count_reducer() is None
    if rank == 0:
        _criterion_loss(model, xh[sl], xo[sl], mask[sl], cls[sl], seg[sl], noise[:, sl])
    with dp.loss_scope(), torch.no_grad():
        _criterion_loss(model, xh[sl], xo[sl], mask[sl], cls[sl], seg[sl], noise[:, sl])
    assert dp.collective_calls == calls_in_scope
    loss.backward()
    dp.all_reduce_gradients()
    ret[rank] = (dp.flat.grad.clone() * dp.grad_scale, float(loss.detach()), dp.collective_calls)
    dp.close()
    dist.destroy_process_group()


@pytest.mark.parametrize('weighted', [True, False])
def test_count_weighted_loss_reproduces_the_global_mean(weighted):
    """Two ranks with different numbers of valid targets: with count_weighted_loss the averaged rank gradients (and the
    averaged rank losses) are those of ONE process on the whole batch; without it they are a mean of per-rank means --
    measurably different here, which is what makes the positive case meaningful."""
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd import kernels, losses
    from twog_gcn_amd.distributed import DataParallel
    from tests.fake_kernels import FakeKernels
    port = 35500 + os.getpid() % 2000 + (7 if weighted else 0)
    ret = mp.Manager().dict()
    mp.spawn(_worker_counts, args=(2, port, ret, weighted), nprocs=2, join=True)
    kernels._set_backend_for_tests(FakeKernels())
    try:
        assert losses.get_count_reducer() is None
        model = _tiny_model(seed=0)
        dp = DataParallel(model)
        xh, xo, mask, _, _ = _batch(4)
        cls, seg = _ragged_targets(4)
        dp.zero_grad()
        loss = _criterion_loss(model, xh, xo, mask, cls, seg, _noise_all(4))
        loss.backward()
        ref, scale = dp.flat.grad, max(1.0, float(dp.flat.grad.abs().max()))
        err = max(float((ret[r][0] - ref).abs().max()) for r in (0, 1))
        mean_loss = 0.5 * (ret[0][1] + ret[1][1])
        if weighted:
            assert err < 2e-5 * scale, err
            assert abs(mean_loss - float(loss.detach())) < 1e-5 * max(1.0, abs(float(loss.detach())))
            assert ret[0][2] > 1   # the count reduction + the gradient buckets
        else:
            assert err > 1e-3 * scale, ('the ragged batch should make the unweighted average differ', err)
    finally:
        kernels._set_backend_for_tests(None)


def test_model_deepcopy_does_not_drag_the_wrapper_along():
    """An EMA copy (copy.deepcopy) or torch.save of a wrapped model must not copy the DataParallel wrapper, its flat
    buffers or its hooks: they live in ops' weak side table, not in the module's __dict__."""
    import copy
    import io
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd import kernels, ops
    from twog_gcn_amd.distributed import DataParallel
    from tests.fake_kernels import FakeKernels
    kernels._set_backend_for_tests(FakeKernels())
    try:
        model = _tiny_model(seed=0)
        dp = DataParallel(model, sync_bn=True, global_noise_seed=5)
        calls = []
        ops.set_grad_stage_hook(model, lambda stage: calls.append(stage))
        assert not any(k.startswith('_twog') or k in ('_bn_stats_reduce', '_noise_shard') for k in vars(model))
        ema = copy.deepcopy(model)
        assert ops.get_model_extra(ema, 'stage_hook') is None and ops.get_model_extra(ema, 'bn_stats_reduce') is None
        assert ops.get_model_extra(model, 'stage_hook') is not None
        buf = io.BytesIO()
        torch.save(model, buf)                      # pickles the module itself: nothing un-picklable hangs on it
        xh, xo, mask, tgt, noise = _batch(2)
        ema._gumbel_noise_override = noise
        out = ema(xh, xo, mask, human_segmentation=torch.ones(xh.shape[:3]))
        torch.nn.functional.nll_loss(out[4], tgt).backward()
        assert calls == []                          # the copy's backward never triggers the original's hook
        dp.close()
        assert ops.get_model_extra(model, 'bn_stats_reduce') is None and ops.get_model_extra(model, 'noise_shard') is None
    finally:
        kernels._set_backend_for_tests(None)


def _wrap_and_drop_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import gc
    import weakref
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd import kernels, ops
    from twog_gcn_amd.distributed import DataParallel
    from tests.fake_kernels import FakeKernels
    kernels._set_backend_for_tests(FakeKernels())
    torch.set_num_threads(2)
    # (a) wrapper AND model dropped without close(): both go, the side table forgets the model
    model = _tiny_model(seed=0)
    dp = DataParallel(model, sync_bn=True, global_noise_seed=5, count_weighted_loss=True, force_collectives=True)
    assert ops.get_model_extra(model, 'stage_hook') is not None
    refs = (weakref.ref(model), weakref.ref(dp), weakref.ref(dp.flat.flat))
    n_before = len(ops._MODEL_EXTRAS)
    del dp, model
    gc.collect()
    a = [r() is None for r in refs] + [len(ops._MODEL_EXTRAS) == n_before - 1]
    # (b) only the wrapper dropped (re-wrap of a live model): its hooks and reducers leave the table
    model = _tiny_model(seed=0)
    dp = DataParallel(model, sync_bn=True, global_noise_seed=5, force_collectives=True)
    r = weakref.ref(dp)
    del dp
    gc.collect()
    b = [r() is None, ops.get_model_extra(model, 'stage_hook') is None, ops.get_model_extra(model, 'bn_stats_reduce') is None,
         ops.get_model_extra(model, 'noise_shard') is None]
    # the model still runs (no dangling hook fires in its backward pass)
    xh, xo, mask, tgt, noise = _batch(2)
    model._gumbel_noise_override = noise
    out = model(xh, xo, mask, human_segmentation=torch.ones(xh.shape[:3]))
    torch.nn.functional.nll_loss(out[4], tgt).backward()
    # (c) re-wrap: the successor is constructed BEFORE the first wrapper is collected (`dp = DataParallel(model, ...)`
    # rebinding the name); the old wrapper's finalizer / close() must leave the successor's reducers alone
    model = _tiny_model(seed=0)
    dp1 = DataParallel(model, sync_bn=True, global_noise_seed=5, force_collectives=True)
    dp2 = DataParallel(model, sync_bn=True, global_noise_seed=6, force_collectives=True)
    red2, shard2 = ops.get_model_extra(model, 'bn_stats_reduce'), ops.get_model_extra(model, 'noise_shard')
    del dp1
    gc.collect()
    c = [ops.get_model_extra(model, 'bn_stats_reduce') is red2, ops.get_model_extra(model, 'noise_shard') is shard2,
         red2 is not None, shard2 is not None, ops.get_model_extra(model, 'stage_hook') is not None]
    dp3 = DataParallel(model, sync_bn=True, global_noise_seed=7, force_collectives=True)
    red3 = ops.get_model_extra(model, 'bn_stats_reduce')
    dp2.close()   # an explicit close of the older wrapper: same rule
    c += [ops.get_model_extra(model, 'bn_stats_reduce') is red3, ops.get_model_extra(model, 'noise_shard') is not None]
    dp3.close()
    c += [ops.get_model_extra(model, 'bn_stats_reduce') is None, ops.get_model_extra(model, 'noise_shard') is None]
    ret['a'], ret['b'], ret['c'] = a, b, c
    dist.destroy_process_group()


def test_discarded_wrapper_and_model_are_collected_without_close():
    """the side table's values must not own the wrapper -- a wrapper (and its model, and the flat
    parameter / gradient buffers) dropped WITHOUT close() is collected; a dropped wrapper of a live model takes its stage
    hook, BatchNorm reducer and noise shard with it."""
    port = 37500 + os.getpid() % 2000
    ret = mp.Manager().dict()
    mp.spawn(_wrap_and_drop_worker, args=(1, port, ret), nprocs=1, join=True)
    assert all(ret['c']), ret['c']
    assert all(ret['a']), ret['a']
    assert all(ret['b']), ret['b']


# A persistent launch of ONE rank's backward pass gives up late: no rank may step on the all-reduced
# gradients, and no rank may be left waiting in a collective. Two gloo ranks; the torch test double of the kernel interface
# with the two hooks of the HIP backend the protocol uses -- guard_persistent (on the device: marks the tensors NaN if an error
# word is pending) and verify_persistent (reads the words, raises) -- where rank 0 plays the rank whose launch gave up.
def _late_failure_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd import kernels
    from twog_gcn_amd.distributed import DataParallel
    from tests.fake_kernels import FakeKernels

    class Failing(FakeKernels):
        failed = False            # "an error word of this pass is set"

        def guard_persistent(self, dev, outs):
            if self.failed:
                for o in outs:
                    o.fill_(float('nan'))
            return True

        def verify_persistent(self, dev=None):
            if self.failed:
                self.failed = False
                raise RuntimeError('twog_segrnn_bwd_persistent: a persistent launch could not keep its grid resident')

    fk = Failing()
    kernels._set_backend_for_tests(fk)
    torch.set_num_threads(2)
    model = _tiny_model(seed=0)
    dp = DataParallel(model, bucket_mb=1)
    xh, xo, mask, tgt, noise = _batch(4)
    sl = slice(rank * 2, rank * 2 + 2)
    out = {}
    for step, fail in enumerate((False, True, False)):
        dp.zero_grad()
        loss = _loss(model, xh[sl], xo[sl], mask[sl], tgt[sl], noise[:, sl])
        fk.failed = fail and rank == 0     # a launch of THIS backward pass gives up
        loss.backward()                    # never raises: the peers are in their collectives
        try:
            dp.all_reduce_gradients()
            out[step] = ('ok', bool(torch.isfinite(dp.flat.grad).all()))
        except RuntimeError as e:
            out[step] = ('raised', str(e)[:80])
    ret[rank] = out
    dist.destroy_process_group()


def test_a_late_persistent_failure_on_one_rank_raises_on_every_rank_after_the_collectives():
    port = 29500 + (os.getpid() * 7 + 3) % 2000
    ret = mp.Manager().dict()
    mp.spawn(_late_failure_worker, args=(2, port, ret), nprocs=2, join=True)
    for rank in (0, 1):
        o = ret[rank]
        assert o[0] == ('ok', True), (rank, o)
        assert o[1][0] == 'raised', (rank, o)          # BOTH ranks: the one whose launch gave up and its peer
        assert o[2] == ('ok', True), (rank, o)          # and the next step is clean on both
    assert 'persistent launch' in ret[0][1][1] and 'rank of this group' in ret[1][1][1], dict(ret)
