"""CPU: the host-side composition (2g-gcn_amd/ops.py + models.py: buffer layouts, GEMM operand forms, hand-derived
backward) executed with the torch test double of the kernel interface, against golden vectors captured from the real
reference. The HIP kernels themselves are checked on the GPU (tests/test_kernels_gpu.py, test_parity_gpu.py)."""
import numpy as np
import pytest
import torch

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import kernels as twog_kernels
from twog_gcn_amd import ops
from twog_gcn_amd.models import TGGCN, select_model
from tests.fake_kernels import FakeKernels
from tests.helpers import G4_CASES, G4_CASES_R2_BUILT, load_g4, det_state_dict, g4_inputs, sample_grad
from oracle import detgen


@pytest.fixture()
def fake_backend():
    twog_kernels._set_backend_for_tests(FakeKernels())
    yield
    twog_kernels._set_backend_for_tests(None)


def build_model(meta):
    N = meta['N']
    m = TGGCN(input_size=(2048 + 4 * N, 2048), num_classes=tuple(meta['classes']), **meta['cfg'])
    sd = det_state_dict(meta['state_dict_shapes'], seed=meta['seed'], gain=meta['gain'])
    m.load_state_dict(sd)
    return m


@pytest.mark.parametrize('name', G4_CASES + G4_CASES_R2_BUILT)
def test_full_forward_backward_vs_reference(name, fake_backend):
    z, meta = load_g4(name)
    m = build_model(meta)
    m.train()
    noise = torch.from_numpy(z['gumbel_noise'])
    m._gumbel_noise_override = noise if len(noise) else None
    out = m(**g4_inputs(z))
    n_out = len([k for k in z.files if k.startswith('out')])
    assert len(out) == n_out
    for i, o in enumerate(out):
        ref = z[f'out{i}']
        assert tuple(o.shape) == ref.shape, (i, o.shape, ref.shape)
        if ref.ndim == 3 and np.all((ref == 0) | (ref == 1)) and i < n_out - 4:
            assert np.array_equal(o.detach().numpy(), ref), f'out{i}'
        else:
            assert np.abs(o.detach().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), f'out{i}'
    bn = m.geometry_embedding_gcn.joint_embed.cnn[0].bn
    assert np.allclose(bn.running_mean.numpy(), z['bn_running_mean'], rtol=1e-5, atol=1e-6)
    assert np.allclose(bn.running_var.numpy(), z['bn_running_var'], rtol=1e-5, atol=1e-6)
    if name == 'c2_dot_st':
        # the reference's own backward is broken for 'st' (upstream bug); check ours at least runs
        loss = sum((o * o).sum() for o in out if o.requires_grad)
        loss.backward()
        return
    loss = 0
    for i, o in enumerate(out):
        if o.requires_grad:
            r = torch.from_numpy(detgen.normal(f'{name}.r{i}', tuple(o.shape), seed=meta['seed']))
            loss = loss + (o * r).sum()
    assert abs(float(loss.detach()) - float(z['loss'])) < 1e-3 * max(1.0, abs(float(z['loss'])))
    loss.backward()
    none_ref = set(z['none_grads'].tolist())
    for pname, p in m.named_parameters():
        if pname in none_ref:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, pname
            continue
        assert p.grad is not None, pname
        g_ref = z['grad_' + pname]
        g = sample_grad(p.grad)
        scale = max(np.abs(g_ref).max(), 1e-6)
        assert np.abs(g - g_ref).max() < 3e-4 * scale + 2e-6, (pname, float(np.abs(g - g_ref).max()), float(scale))


def test_select_model_and_errors(fake_backend):
    assert select_model('2G-GCN') is TGGCN
    with pytest.raises(KeyError):
        select_model('nope')
    with pytest.raises(ValueError):
        TGGCN(input_size=(2152, 2048), num_classes=(13, None), hidden_size=8, discrete_optimization_strategy='bogus',
              message_type='v2', message_granularity='v1', attention_style='v3')
    # a bare TGGCN(input_size, num_classes) -- the reference's constructor defaults (relational messages, frame level
    # only) -- runs; the golden case c2_ctor_defaults pins its numbers
    m = TGGCN(input_size=(2152, 2048), num_classes=(13, None), hidden_size=8)
    out = m(torch.rand(1, 2, 2, 2152), torch.rand(1, 2, 4, 2048), torch.ones(1, 4))
    assert len(out) == 6 and all(torch.isfinite(o).all() for o in out)
    m = TGGCN(input_size=(2152, 2048), num_classes=(13, None), hidden_size=8, message_segment=True)
    out = m(torch.rand(1, 2, 2, 2152), torch.rand(1, 2, 4, 2048), torch.ones(1, 4))   # ... with segment-level messages
    assert len(out) == 6 and all(torch.isfinite(o).all() for o in out)
    # what the path cannot run says so loudly (no silent fallback)
    with pytest.raises(NotImplementedError):
        TGGCN(input_size=(2152, 2048), num_classes=(13, None), hidden_size=7, add_time_position=True,
              positional_encoding_style='p')(torch.zeros(1, 2, 2, 2152), torch.zeros(1, 2, 4, 2048), torch.ones(1, 4))
    with pytest.raises(AttributeError):   # 'same_as_human' with two humans: the reference never built that MLP either
        TGGCN(input_size=(2152, 2048), num_classes=(13, None), hidden_size=8, message_type='v2', message_granularity='v1',
              attention_style='v3', object_segment_update_strategy='sah')(
            torch.zeros(1, 2, 2, 2152), torch.zeros(1, 2, 4, 2048), torch.ones(1, 4))


def test_eval_mode_uses_running_stats(fake_backend):
    z, meta = load_g4('c2_stage1')
    m = build_model(meta)
    m.eval()
    m._gumbel_noise_override = torch.from_numpy(z['gumbel_noise'])
    bn = m.geometry_embedding_gcn.joint_embed.cnn[0].bn
    rm = bn.running_mean.clone()
    with torch.no_grad():
        out = m(**g4_inputs(z))
    assert torch.equal(bn.running_mean, rm)
    assert len(out) == 6


def test_in_place_gradient_route_equals_autograd_accumulation(fake_backend):
    """Parameters that already own a .grad buffer get `grad += g` from the producing kernels (ops._Grads sinks); the
    result must equal autograd's own accumulation, including across two backward passes."""
    z, meta = load_g4('c2_stage1')
    noise = torch.from_numpy(z['gumbel_noise'])

    def run(prealloc, passes):
        m = build_model(meta)
        m.train()
        m._gumbel_noise_override = noise if len(noise) else None
        if prealloc:
            for p in m.parameters():
                p.grad = torch.zeros_like(p)
            ops.enable_grad_sinks(m.parameters())
        for _ in range(passes):
            out = m(**g4_inputs(z))
            sum((o * o).sum() for o in out if o.requires_grad).backward()
        return {n: (None if p.grad is None else p.grad.clone()) for n, p in m.named_parameters()}

    ref1, got1, got2 = run(False, 1), run(True, 1), run(True, 2)
    for n, g in ref1.items():
        if g is None:
            assert float(got1[n].abs().max()) == 0.0, n      # dead parameter: the preallocated buffer stays zero
            continue
        scale = float(g.abs().max()) + 1e-12
        assert float((got1[n] - g).abs().max()) <= 1e-6 * scale, n
        assert float((got2[n] - 2 * g).abs().max()) <= 1e-5 * scale, n


@pytest.mark.parametrize('sinks', [False, True])
def test_bias_gradient_taken_by_the_held_weight_gradient_gemm(sinks, fake_backend, monkeypatch):
    """ops._Grads holds a tall dW = dY^T X problem for one call; the colsum(dY) that follows attaches to it (twog_gemm_t::
    a_colsum: the GEMM that streams dY also returns its column sums) -- with and without gradient sinks the parameter
    gradients must equal the route with separate column-sum launches (TWOG_DW_COLSUM=0), and the pairing must actually
    happen (the double counts the GEMM problems that carry a request)."""
    z, meta = load_g4('c2_stage1')
    noise = torch.from_numpy(z['gumbel_noise'])

    def run(fused):
        monkeypatch.setenv('TWOG_DW_COLSUM', '1' if fused else '0')
        fk = FakeKernels()
        twog_kernels._set_backend_for_tests(fk)
        n_req = [0]
        inner = fk.gemm

        def gemm(problems, **kw):
            n_req[0] += sum(1 for p in problems if p.get('colsum') is not None)
            return inner(problems, **kw)
        fk.gemm = gemm
        m = build_model(meta)
        m.train()
        m._gumbel_noise_override = noise if len(noise) else None
        if sinks:
            for p in m.parameters():
                p.grad = torch.zeros_like(p)
            ops.enable_grad_sinks(m.parameters())
        out = m(**g4_inputs(z))
        sum((o * o).sum() for o in out if o.requires_grad).backward()
        return {n: (None if p.grad is None else p.grad.clone()) for n, p in m.named_parameters()}, n_req[0]

    ref, n0 = run(False)
    got, n1 = run(True)
    assert n0 == 0 and n1 >= 10, (n0, n1)
    for n, g in ref.items():
        if g is None:
            assert got[n] is None, n
            continue
        scale = float(g.abs().max()) + 1e-12
        assert float((got[n] - g).abs().max()) <= 2e-6 * scale, n


@pytest.mark.parametrize('case', ['c2_stage1', 'c1_stage2', 'c5_stage1'])
def test_operands_of_deferred_gradient_launches_do_not_change_before_they_are_issued(case, fake_backend, monkeypatch):
    """ops._Grads defers weight-gradient GEMMs, column sums, copies and `grad += g` additions until flush(); the contract is
    that no operand changes in between. TWOG_VERIFY_DEFERRED=1 snapshots every deferred operand and compares
    it at issue time: a whole backward pass of each dataset layout runs clean under it, with and without gradient sinks --
    and a violation is caught (an operand overwritten between the call and the flush)."""
    monkeypatch.setenv('TWOG_VERIFY_DEFERRED', '1')
    z, meta = load_g4(case)
    noise = torch.from_numpy(z['gumbel_noise'])
    for sinks in (False, True):
        m = build_model(meta)
        m.train()
        m._gumbel_noise_override = noise if len(noise) else None
        if sinks:
            for p in m.parameters():
                p.grad = torch.zeros_like(p)
            ops.enable_grad_sinks(m.parameters())
        out = m(**g4_inputs(z))
        sum((o * o).sum() for o in out if o.requires_grad).backward()
        assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    G = ops._Grads(twog_kernels.get_kernels())
    x = torch.randn(64, 8)
    G.colsum(x)
    x[3, 3] += 1.0                                  # the violation: the operand changes before the launch is issued
    with pytest.raises(RuntimeError, match='changed between the call and flush'):
        G.flush()


@pytest.mark.parametrize('case', ['c2_stage1', 'c1_stage2', 'c5_stage1'])
def test_gradient_stages_are_final_when_the_hook_fires(case, fake_backend):
    """distributed.DataParallel starts a stage's all-reduce from ops' stage hook: at that moment every gradient of the
    stage must already have its final value (nothing may be added to it later in the backward pass)."""
    from twog_gcn_amd import ops
    from twog_gcn_amd.distributed import FlatParameters
    z, meta = load_g4(case)
    m = build_model(meta)
    m.train()
    noise = torch.from_numpy(z['gumbel_noise'])
    m._gumbel_noise_override = noise if len(noise) else None
    flat = FlatParameters(m, stage_of=ops.grad_ready_stage)
    assert set(flat.stage_ranges) == {0, 1, 2}
    assert sum(e - b for b, e in flat.stage_ranges.values()) == flat.numel
    snaps = {}
    ops.set_grad_stage_hook(m, lambda st: snaps.setdefault(st, flat.grad[slice(*flat.stage_ranges[st])].clone()))
    try:
        out = m(**g4_inputs(z))
        sum((o * o).sum() for o in out if o.requires_grad).backward()
    finally:
        ops.set_grad_stage_hook(m, None)
    assert set(snaps) == {0, 1}
    for st, snap in snaps.items():
        final = flat.grad[slice(*flat.stage_ranges[st])]
        assert torch.equal(snap, final), f'stage {st} gradients changed after the hook'
        assert float(final.abs().max()) > 0
    # and the stage-2 block (embeddings, GCN) is what remains
    b2, e2 = flat.stage_ranges[2]
    assert float(flat.grad[b2:e2].abs().max()) > 0
    # the hook is scoped to its model: the backward of ANOTHER model in the process never calls it
    calls = []
    ops.set_grad_stage_hook(m, calls.append)
    other = build_model(meta)
    other.train()
    other._gumbel_noise_override = m._gumbel_noise_override
    sum((o * o).sum() for o in other(**g4_inputs(z)) if o.requires_grad).backward()
    assert calls == []
    ops.set_grad_stage_hook(m, None)


def test_randomised_layouts_vs_oracle(fake_backend):
    """Host composition (kernel test double) against the oracle over random layouts / gate semantics / message switches
    (the generator of tools/parity_fuzz.py; the same sweep runs on the HIP kernels in tests/test_parity_gpu.py).
    Covers e.g. a GIVEN segmentation combined with filter_discrete_updates (the filter applies to it too)."""
    import random
    from tools.parity_fuzz import one_case
    rng = random.Random(2)
    seen, axes = set(), set()
    for i in range(16):
        d = one_case(rng, i, dev='cpu')
        assert d['worst_output_rel'] < 1e-4, d     # (gradients: asserted per tensor inside one_case)
        seen.add((d['given_seg'], d['filt']))
        axes.update([d['strat'], d['att'], d['agg'], ('shared heads', d['share'] and not d['cat'], d['training']),
                     ('cat levels', d['cat'], d['training'])])
    assert len(seen) == 4 and {'gs', 'st', 'v2', 'v3', 'att', 'mp', ('shared heads', True, True),
                               ('cat levels', True, True)} <= axes


def test_inspect_model_returns_the_reference_attention_scores(fake_backend):
    """predict.py --inspect_model: (outputs, [a_frame, a_segment_forward, a_segment_backward]), each (bs, H, T, O)
    (vhoi/models.py:928-933), against the oracle."""
    from oracle import cpu_ref
    z, meta = load_g4('c2_stage1')
    m = build_model(meta).eval()
    noise = torch.from_numpy(z['gumbel_noise'])
    m._gumbel_noise_override = noise
    kw = g4_inputs(z)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        out, att = m(**kw, inspect_model=True)
        ref_out, ref_att = cpu_ref.tggcn_forward(sd, dict(m.cfg), kw['x_human'], kw['x_objects'], kw['objects_mask'],
                                                 human_segmentation=kw.get('human_segmentation'),
                                                 objects_segmentation=kw.get('objects_segmentation'), training=False,
                                                 gumbel_noise=noise, inspect_model=True)
    assert len(out) == len(ref_out) == 6 and len(att) == len(ref_att) == 3
    bs, T, H = z['x_human'].shape[:3]
    O = z['x_objects'].shape[2]
    for a, r in zip(att, ref_att):
        assert tuple(a.shape) == tuple(r.shape) == (bs, H, T, O)
        assert (a - r).abs().max().item() < 1e-5
        assert not a.requires_grad
    for o, r in zip(out, ref_out):
        assert (o - r).abs().max().item() < 1e-4 * max(1.0, r.abs().max().item())
    # the slice predict.py takes (human 0) sums to one over the real objects wherever a clip has any
    s = att[0][:, 0].sum(-1)
    assert torch.all(((s - 1).abs() < 1e-5) | (s.abs() < 1e-6))


def test_cat_level_states_in_place_gradient_route(fake_backend):
    """cat_level_states writes the head weight gradient as two column blocks: the in-place route (existing .grad buffers,
    the DataParallel case) must add exactly what the autograd route returns."""
    z, meta = load_g4('c2_stage1')
    cfg = dict(meta['cfg'], cat_level_states=True)
    N = meta['N']
    torch.manual_seed(4)
    m = TGGCN(input_size=(2048 + 4 * N, 2048), num_classes=tuple(meta['classes']), **cfg).train()
    assert m.human_recognition_mlp[0].weight.shape[1] == 4 * cfg['hidden_size']
    m._gumbel_noise_override = torch.from_numpy(z['gumbel_noise'])
    kw = g4_inputs(z)

    def run():
        out = m(**kw)
        sum((o * (i + 1)).sum() for i, o in enumerate(out) if o.requires_grad).backward()

    run()                                                   # autograd route: .grad tensors are created
    first = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    ops.enable_grad_sinks(m.parameters())
    run()                                                   # in-place route: kernels add into the existing buffers
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        scale = max(first[n].abs().max().item(), 1e-6)
        assert (p.grad - 2 * first[n]).abs().max().item() < 1e-5 * scale, n


def test_gradient_sink_route_is_opt_in(fake_backend):
    """Without enable_grad_sinks an existing .grad buffer changes nothing about the autograd contract:
    torch.autograd.grad returns every gradient (and leaves .grad alone), post-accumulate-grad hooks fire; a tagged
    parameter that carries such a hook keeps the autograd route too."""
    z, meta = load_g4('c2_stage1')
    noise = torch.from_numpy(z['gumbel_noise'])
    m = build_model(meta)
    m.train()
    m._gumbel_noise_override = noise if len(noise) else None
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    out = m(**g4_inputs(z))
    loss = sum((o * o).sum() for o in out if o.requires_grad)
    used = [p for n, p in m.named_parameters() if 'att_mlp' not in n and 'geometry_to_object_segment' not in n]
    gs = torch.autograd.grad(loss, used, allow_unused=True, retain_graph=True)
    assert sum(g is not None for g in gs) > 50
    assert all(float(p.grad.abs().max()) == 0.0 for p in m.parameters())     # .grad untouched by autograd.grad
    fired = []
    w = m.human_embedding_mlp[0].weight
    w.register_post_accumulate_grad_hook(lambda p: fired.append(float(p.grad.abs().sum())))
    ops.enable_grad_sinks(m.parameters())                                     # tagged, but w has a hook
    loss.backward()
    assert len(fired) == 1 and fired[0] > 0
    ref = dict(zip([id(p) for p in used], gs))
    for p in used:
        g = ref[id(p)]
        if g is not None:
            assert float((p.grad - g).abs().max()) <= 1e-5 * (float(g.abs().max()) + 1e-12)


def test_no_reference_cycle_keeps_saved_buffers_alive(fake_backend):
    """The autograd node must not sit in a reference cycle with its saved state: after the step's tensors go out of
    scope every saved buffer is freed by reference counting alone (no cyclic-GC pass) -- otherwise each training step
    leaves gigabytes of buffers behind until the collector happens to run."""
    import gc
    import weakref
    z, meta = load_g4('c2_stage1')
    m = build_model(meta)
    m.train()
    noise = torch.from_numpy(z['gumbel_noise'])
    m._gumbel_noise_override = noise if len(noise) else None
    gc.disable()
    try:
        out = m(**g4_inputs(z))
        node = next(o.grad_fn for o in out if o.grad_fn is not None)
        probe = weakref.ref(ops.saved_state(node)['HUM'])
        sum((o * o).sum() for o in out if o.requires_grad).backward()
        del out, node
        assert probe() is None, 'saved buffers survived without a GC pass: reference cycle through the autograd node'
    finally:
        gc.enable()


@pytest.mark.parametrize('strategy', ['sah', 'coh', 'ind'])
@pytest.mark.parametrize('filt', [False, True])
def test_callers_inputs_are_never_written(strategy, filt, fake_backend):
    """One human, one object, one frame: every expanded (bs, T, O) view of the human's decisions has the shape of the
    tensor itself. A given human segmentation must come back untouched (the forced end of the objects' copy of it was
    once written through such a view) and equal output 0, as in the reference (vhoi/models.py:738-745)."""
    torch.manual_seed(3)
    N, bs, T, H, O = 19, 4, 1, 1, 1
    m = TGGCN(input_size=(2048 + 4 * N, 2048), num_classes=(13, None), hidden_size=16, gcn_node=N,
              object_segment_update_strategy=strategy, filter_discrete_updates=filt, update_segment_threshold=0.3)
    m.train()
    x_h, x_o = torch.rand(bs, T, H, 2048 + 4 * N), torch.rand(bs, T, O, 2048)
    mask = torch.ones(bs, O)
    seg = torch.tensor([1.0, 0.0, 1.0, 0.0]).view(bs, T, H)
    keep = [t.clone() for t in (x_h, x_o, mask, seg)]
    out = m(x_h, x_o, mask, human_segmentation=seg)
    sum((o * o).sum() for o in out if o.requires_grad).backward()
    for t, k in zip((x_h, x_o, mask, seg), keep):
        assert torch.equal(t, k)
    if not filt:
        assert torch.equal(out[0], keep[3])


def test_saved_state_is_released_by_backward_not_by_the_loss(fake_backend):
    """Lifetime of the saved state follows torch's rule for saved tensors: gone when backward has run (without
    retain_graph) even though the caller still holds the outputs, reachable again and again with retain_graph."""
    import gc
    import weakref
    z, meta = load_g4('c2_stage1')
    m = build_model(meta)
    m.train()
    noise = torch.from_numpy(z['gumbel_noise'])
    m._gumbel_noise_override = noise if len(noise) else None
    gc.disable()
    try:
        out = m(**g4_inputs(z))
        node = next(o.grad_fn for o in out if o.grad_fn is not None)
        probe = weakref.ref(ops.saved_state(node)['HUM'])
        loss = sum((o * o).sum() for o in out if o.requires_grad)
        loss.backward(retain_graph=True)
        g1 = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        assert probe() is not None
        m.zero_grad()
        loss.backward()                      # second run on the retained state, then the engine lets go of it
        for n, p in m.named_parameters():
            if p.grad is not None:
                assert torch.equal(p.grad, g1[n]), n
        assert probe() is None, 'saved buffers outlived backward while the caller still holds loss and outputs'
        with pytest.raises(RuntimeError):
            ops.saved_state(node)
        assert out[0].shape[0] > 0 and loss.item() == loss.item()   # outputs and loss stay valid
    finally:
        gc.enable()


def test_x3_operand_split_is_exact():
    """The 128x128 GEMM class multiplies fp32 operands on the bf16 matrix cores after splitting every element into three
    bf16 values by truncation (csrc/gemm_f32.hip, split3): h = top 8 significant bits, m = the next 8, l = the rest. The
    arithmetic restated in numpy: the three chunks are exactly representable in bf16 (low 16 bits zero) and add up to the
    element EXACTLY -- so the six products the kernel keeps differ from the exact product only by the three dropped
    cross terms (m l, l m, l l: <= 2^-21 relative in the worst case, ~2^-24 on average)."""
    import numpy as np
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.randn(20000).astype(np.float32) * 10.0 ** rng.randint(-20, 20, 20000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.1754944e-38, 1e-30, 0.1, 1.0 + 2.0 ** -23,
                                  np.nextafter(np.float32(2.0), np.float32(0.0))], dtype=np.float32)])
    mask = np.uint32(0xffff0000)
    h = (x.view(np.uint32) & mask).view(np.float32)
    r1 = x - h
    m = (r1.view(np.uint32) & mask).view(np.float32)
    r2 = r1 - m
    l = (r2.view(np.uint32) & mask).view(np.float32)
    assert np.array_equal(l, r2), 'the third chunk needs no truncation: at most 8 significant bits are left'
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64)) + l.astype(np.float64), x.astype(np.float64))
    for c in (h, m, l):
        assert not np.any(c.view(np.uint32) & np.uint32(0xffff)), 'every chunk is a bf16 value'
    # the product from the six kept chunk products against the exact product
    y = rng.randn(x.size).astype(np.float32)
    hy = (y.view(np.uint32) & mask).view(np.float32); ry = y - hy
    my = (ry.view(np.uint32) & mask).view(np.float32); ly = ry - my
    f = np.float64
    six = f(h) * f(hy) + f(h) * f(my) + f(m) * f(hy) + f(m) * f(my) + f(h) * f(ly) + f(l) * f(hy)
    exact = f(x) * f(y)
    ok = np.isfinite(exact) & (np.abs(exact) > 1e-290)
    rel = np.abs(six[ok] - exact[ok]) / np.abs(exact[ok])
    assert np.max(rel) <= 2.0 ** -21 and np.mean(rel) <= 2.0 ** -23


GENERAL_FORMS = {
    'relational': dict(message_type='v1'),
    'specific': dict(message_type='v2', message_granularity='v2', attention_style='v3'),
    'concat': dict(message_type='v2', message_granularity='v1', attention_style='v1'),
    'bilinear': dict(message_type='v2', message_granularity='v1', attention_style='v4'),
    'specific_mean_pool': dict(message_type='v2', message_granularity='v2', message_aggregation='mp'),
    'distance': dict(message_type='v2', message_granularity='v1', attention_style='v3'),
}


@pytest.mark.parametrize('form', sorted(GENERAL_FORMS))
def test_general_segment_loop_replay_logic_equals_the_composed_loop_and_the_oracle(form, fake_backend, monkeypatch):
    """The HOST side of the replayed general segment loop (ops.segment_recurrence_general_*: which steps are composed,
    which descriptors the backward pass finds, per-step slots, deferred parameter gradients) with the test double's
    recorder -- every tensor argument of a replayed step is the recorded view moved by a constant -- at T = 10: replayed
    and composed loops bit-identical, both against the oracle. (The library's replay of descriptor bytes is checked on the
    GPU: tests/test_parity_gpu.py.)"""
    from oracle import cpu_ref
    bs, T, H, O, N, h = 2, 10, 2, 3, 26, 16
    torch.manual_seed(5)
    m = TGGCN(input_size=(2048 + 4 * N, 2048), num_classes=(13, None), hidden_size=h, gcn_node=N, message_segment=True,
              **GENERAL_FORMS[form])
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    x_human, x_objects = torch.rand(bs, T, H, 2048 + 4 * N, generator=g), torch.rand(bs, T, O, 2048, generator=g)
    mask = torch.ones(bs, O)
    mask[1, O - 1] = 0.0
    x_objects = x_objects * mask[:, None, :, None]
    kw = dict(human_segmentation=(torch.rand(bs, T, H, generator=g) < 0.6).float())
    if form == 'distance':
        def dd(*shape):
            d = torch.rand(*shape, generator=g) * 2 + 0.05
            d[torch.rand(*shape, generator=g) < 0.15] = 0.0
            return d
        kw.update(human_human_distances=dd(bs, T, H, H), human_object_distances=dd(bs, T, H, O),
                  object_object_distances=dd(bs, T, O, O))
    noise = torch.distributions.gumbel.Gumbel(0.0, 1.0).sample((T * O, bs, 2))
    m._gumbel_noise_override = noise
    K = twog_kernels.get_kernels()
    runs = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('TWOG_GENERAL_TAPE', mode)
        m.zero_grad(set_to_none=True)
        replays = []
        real = K.tape_run
        K.tape_run = lambda a, b, k0, k1, dev=None: (replays.append((k0, k1)), real(a, b, k0, k1, dev))[1]
        try:
            out = m(x_human, x_objects, mask, **kw)
            rs = [torch.randn(o.shape, generator=torch.Generator().manual_seed(i)) for i, o in enumerate(out)]
            sum((o * r).sum() for o, r in zip(out, rs) if o.requires_grad).backward()
        finally:
            del K.tape_run
        assert replays == ([(0, T - 1), (0, T - 2)] if mode == '1' else []), replays
        runs[mode] = ([o.detach().clone() for o in out],
                      {n: (None if p.grad is None else p.grad.clone()) for n, p in m.named_parameters()})
    for a, b in zip(runs['1'][0], runs['0'][0]):
        assert torch.equal(a, b)
    for n, ga in runs['1'][1].items():
        gb = runs['0'][1][n]
        assert (ga is None) == (gb is None) and (ga is None or torch.equal(ga, gb)), n
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone()) for k, v in sd.items()}
    ref = cpu_ref.tggcn_forward(osd, dict(m.cfg), x_human, x_objects, mask, training=True, gumbel_noise=noise, **kw)
    sum((o * r).sum() for o, r in zip(ref, rs) if o.requires_grad).backward()
    for o, r in zip(runs['1'][0], ref):
        assert float((o - r.detach()).abs().max()) <= 1e-4 * max(1.0, float(r.detach().abs().max()))
    for n, ga in runs['1'][1].items():
        g_ref = osd[n].grad
        if g_ref is None:
            assert ga is None or float(ga.abs().max()) == 0.0, n
            continue
        assert float((ga - g_ref).abs().max()) <= 5e-4 * max(float(g_ref.abs().max()), 1e-6) + 5e-6, n
