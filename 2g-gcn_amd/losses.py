"""Multi-task criterion of the 2G-GCN training step on the HIP library.

Mirror of the reference's loss interface -- `vhoi/losses.py:8-70` (`select_loss`) and `pyrutils/torch/losses.py:7-51`
(`multi_task_loss`, `binary_cross_entropy_loss`, `budget_loss`) -- with the same names, argument meaning and return
values (a list of weighted scalar losses, one per output of the model), so `train.py` keeps calling
`criterion(output, target)` / `sum(losses).backward()`. All terms run in ONE forward and ONE backward kernel launch
(`twog_multitask_loss_fwd/bwd`), with no host synchronisation (the reference calls `mask.sum().item()` per term).
Terms whose weight is 0 (stage-1: budget, BCE and the frame-level NLL terms) get no backward work at all.
"""
from functools import partial

import torch
import torch.nn.functional as F

from .kernels import get_kernels

_NLL, _BCE, _BUDGET = 0, 1, 2


def nll_loss(input, target, ignore_index=-1, reduction='mean'):
    """F.nll_loss(input, target, ignore_index, reduction='mean') on the HIP path."""
    return _run([input], [target], [_NLL], [1.0], ignore_index, reduction)[0]


def binary_cross_entropy_loss(input, target, positive_class_weight=1, ignore_value=-1, reduction='mean'):
    """pyrutils/torch/losses.py:7-21."""
    if positive_class_weight != 1:
        raise NotImplementedError('positive_class_weight != 1 is not used by any reference configuration')
    return _run([input], [target], [_BCE], [1.0], ignore_value, reduction)[0]


def budget_loss(input, target, ignore_value=-1, reduction='mean'):
    """pyrutils/torch/losses.py:24-36."""
    return _run([input], [target], [_BUDGET], [1.0], ignore_value, 'mean')[0]


_KIND = {nll_loss: _NLL, F.nll_loss: _NLL, binary_cross_entropy_loss: _BCE, budget_loss: _BUDGET}

# Data-parallel normalisation: every term is a sum over the valid (non-ignored) targets divided by their
# number (pyrutils/torch/losses.py:13-21, :30-36; F.nll_loss reduction='mean', :47). With ragged clips the ranks hold
# different numbers of valid targets, and the average of per-rank means is not the global mean. A registered reducer
# (distributed.DataParallel(count_weighted_loss=True)) maps this rank's per-term counts [terms] (fp64, on the device) to
# (global count) / W: the term becomes sum_rank / (count_global / W), whose average over the W ranks -- and so the
# averaged gradient -- is the global-batch value. No reducer: the reference's single-process arithmetic.
#
# The reducer is a COLLECTIVE, so it is never installed process-wide: it is active only inside
# `with dp.loss_scope():` (count_reducer_scope below), which the training step enters around its criterion call, and only
# while autograd is recording -- a validation pass on one rank, a second model's criterion or a no_grad evaluation issue
# no collective and cannot pair with another rank's all-reduce. Every rank must make the same number of criterion calls
# inside the scope.
_count_reducer = None


class count_reducer_scope:
    """Context manager: `fn` (or None) reduces the per-term counts of every criterion call made inside the scope."""

    def __init__(self, fn):
        self.fn, self.prev = fn, None

    def __enter__(self):
        global _count_reducer
        self.prev, _count_reducer = _count_reducer, self.fn
        return self

    def __exit__(self, *exc):
        global _count_reducer
        _count_reducer = self.prev
        return False


def get_count_reducer():
    return _count_reducer


class _MultiTaskLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, *inputs):
        kinds, targets, weights, ignore, reducer = spec
        K = get_kernels()
        terms = []
        for x, y, k, w in zip(inputs, targets, kinds, weights):
            x = x.detach().contiguous()
            y = (y.to(torch.int64) if k == _NLL else y.to(torch.float32)).contiguous()
            terms.append(dict(kind=k, input=x, target=y, weight=w, ignore=ignore))
        losses, stats = K.multitask_loss_fwd(terms)
        if reducer is not None:
            # six to twelve scalars: sum_rank / (count_global / W) per term, 0 / 0 -> NaN for NLL like torch's mean over an
            # empty selection, 0 for the BCE / budget terms (pyrutils/torch/losses.py:15-16, :32-33). Same arithmetic as
            # loss_final_kernel (csrc/loss.hip): the fp64 quotient rounded to fp32, then the fp32 product with the weight
            # -- with a reducer that returns the counts unchanged (one rank) the losses are bit-identical to the plain path.
            eff = reducer(stats[:, 1].clone())
            val = stats[:, 0] / eff
            soft = torch.tensor([k != _NLL for k in kinds], device=val.device)
            val = torch.where(soft & (eff <= 0), torch.zeros_like(val), val)
            w = torch.tensor([float(x) for x in weights], dtype=torch.float32, device=val.device)
            losses = w * val.to(torch.float32)
            stats = torch.stack([stats[:, 0], eff], 1).contiguous()
        ctx.terms, ctx.stats = terms, stats
        return losses

    @staticmethod
    def backward(ctx, dlosses):
        need = [ctx.needs_input_grad[i + 1] and t['weight'] != 0 for i, t in enumerate(ctx.terms)]
        if not any(need):
            return (None,) + tuple(None for _ in ctx.terms)
        dins = get_kernels().multitask_loss_bwd(ctx.terms, ctx.stats, dlosses, need)
        return (None,) + tuple(dins)


def _run(inputs, targets, kinds, weights, ignore_value, reduction):
    if reduction != 'mean':
        raise NotImplementedError("only reduction='mean' (the reference's setting) is implemented")
    # the scope's reducer, and only while autograd records (read here: grad mode is off inside Function.forward)
    reducer = _count_reducer if torch.is_grad_enabled() else None
    losses = _MultiTaskLoss.apply((list(kinds), list(targets), [float(w) for w in weights], ignore_value, reducer), *inputs)
    return list(losses.unbind(0))


def multi_task_loss(input: list, target: list, loss_functions: list, weight: list = None, ignore_value=-1,
                    reduction: str = 'mean'):
    """pyrutils/torch/losses.py:39-51: the list of weighted losses, one per (input, target, loss function)."""
    if weight is None:
        weight = [1.0] * len(input)
    n = min(len(input), len(target), len(loss_functions), len(weight))  # zip semantics
    kinds = []
    for fn in loss_functions[:n]:
        if fn not in _KIND:
            raise NotImplementedError(f'loss function {fn} is not part of the 2G-GCN criterion')
        kinds.append(_KIND[fn])
    return _run(list(input[:n]), list(target[:n]), kinds, list(weight[:n]), ignore_value, reduction)


def select_loss(model_name: str, model_input_type: str, dataset_name: str, cfg):
    """vhoi/losses.py:8-70 for model '2G-GCN': (criterion, loss_names). cfg needs `.get('misc', default_value={})`."""
    if model_name != '2G-GCN':
        raise NotImplementedError(f'{model_name}: only the 2G-GCN hot path is built (SURVEY section 8)')
    try:
        misc = cfg.get('misc', default_value={})
    except TypeError:  # plain dict
        misc = cfg.get('misc', {})
    hb_weight = ob_weight = 0.0
    if misc.get('budget_loss', {}).get('add', False):
        hb_weight = misc.get('budget_loss', {}).get('human_weight', 1.0)
        ob_weight = misc.get('budget_loss', {}).get('object_weight', 1.0)
    cad = dataset_name == 'cad120'
    weight = [hb_weight, ob_weight] if cad else [hb_weight]
    hs_weight = os_weight = 0.0
    seg = misc.get('segmentation_loss', {})
    s_weight, add_seg = seg.get('weight', 1.0), seg.get('add', False)
    if add_seg and not misc.get('input_human_segmentation', False):
        hs_weight = s_weight
    if add_seg and not misc.get('input_object_segmentation', False):
        os_weight = s_weight
    weight += [hs_weight, os_weight] if cad else [hs_weight]
    weight_val = 0.0 if (add_seg and seg.get('pretrain', False)) else 1.0
    ant_weight = misc.get('anticipation_loss_weight', 1.0)
    fl_weight = misc.get('first_level_loss_weight', 0.0)
    if cad:
        weight += [fl_weight] * 4 + [weight_val, ant_weight, weight_val, ant_weight]
        fns = (budget_loss, budget_loss, binary_cross_entropy_loss, binary_cross_entropy_loss) + (nll_loss,) * 8
        names = ['B_HS', 'B_OS', 'BCE_HS', 'BCE_OS', 'NLL_SAR_F', 'NLL_SAP_F', 'NLL_OAR_F', 'NLL_OAP_F',
                 'NLL_SAR', 'NLL_SAP', 'NLL_OAR', 'NLL_OAP']
    else:
        weight += [fl_weight] * 2 + [weight_val, ant_weight]
        fns = (budget_loss, binary_cross_entropy_loss) + (nll_loss,) * 4
        names = ['B_HS', 'BCE_HS', 'NLL_SAR_F', 'NLL_SAP_F', 'NLL_SAR', 'NLL_SAP']
    return partial(multi_task_loss, loss_functions=fns, weight=weight), names


def select_loss_types(model_name: str, dataset_name: str, cfg):
    """vhoi/losses.py:72-80."""
    if model_name != '2G-GCN':
        raise ValueError(f'Multi-task learning option not implemented for {model_name}')
    return ['budget'] * 2 + ['bce'] * 2 + ['softmax'] * 8 if dataset_name == 'cad120' else ['budget', 'bce'] + ['softmax'] * 4


def select_loss_learning_mask(model_name: str, dataset_name: str, cfg):
    """vhoi/losses.py:83-91."""
    if model_name != '2G-GCN':
        raise ValueError(f'Multi-task learning option not implemented for {model_name}')
    return [False] * 4 + [True] * 8 if dataset_name == 'cad120' else [False] * 2 + [True] * 4
