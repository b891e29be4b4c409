"""Inference post-processing on the device.

Mirror of the steps `predict.py` runs after the model: `match_shape` (:95-116), the ``repeat_interleave`` by the
downsampling factor (:64-70), `process_output`'s argmax (:186-202) and `pyrutils.metrics.f1_at_k` (:68-81) as used by
`evaluate_f1_at_k` (predict.py:229-246). The reference copies every (bs, C, T, E) log-probability tensor to the host and
does these in numpy; here the labels and the per-sequence F1@k are computed by HIP kernels and only the final scalar
(or the int64 labels, C x smaller than the log-probabilities) crosses PCIe.
"""
import numpy as np
import torch

from .kernels import get_kernels


def match_shape(out: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """predict.py:95-116 (kept for callers that still want the resized log-probabilities)."""
    if out.ndim == 3:
        o, t = out.shape[-1], tgt.shape[-1]
        if o >= t:
            return out[..., :t]
        return torch.cat([out, out[..., -1:].expand(*out.shape[:-1], t - o)], dim=-1)
    if out.ndim == 4:
        o, t = out.shape[-2], tgt.shape[-2]
        if o >= t:
            return out[:, :, :t]
        return torch.cat([out, out[:, :, -1:].expand(out.shape[0], out.shape[1], t - o, out.shape[3])], dim=-2)
    return out


def predict_labels(output: torch.Tensor, target: torch.Tensor = None, downsampling: int = 1) -> torch.Tensor:
    """Labels of one model output: ``argmax(match_shape(repeat_interleave(output, downsampling, -2), target), 1)`` in one
    kernel, without materialising the upsampled tensor. output (bs, C, T, E); returns int64 (bs, T_target, E)."""
    if output.ndim != 4:
        raise RuntimeError(f'Number of dimensions for output is {output.ndim}')  # predict.py:66-67
    steps = output.shape[-2] if (downsampling <= 1 or target is None) else target.shape[-2]
    return get_kernels().predict_labels(output, max(1, int(downsampling)), steps)


def process_output(outputs, downsampling: int = 1, targets=None, index_to_name=None):
    """predict.py:186-202 for predictions: list over batches of lists of (bs, C, T, E) outputs -> {index: int64 labels
    (N, T, E) on the device} (the reference returns numpy arrays of the same values)."""
    per_index = {}
    for bi, output in enumerate(outputs):
        for i, tensor in enumerate(output):
            tgt = None if targets is None else targets[bi][i]
            key = index_to_name[i] if index_to_name is not None else i
            per_index.setdefault(key, []).append(predict_labels(tensor, tgt, downsampling))
    return {k: torch.cat(v, 0) for k, v in per_index.items()}


def f1_at_k(y_true, y_pred, num_classes: int, *, overlap: float, ignore_value: float = None) -> float:
    """pyrutils/metrics.py:68-81 for (n_seq, n_steps) label matrices (device tensors or anything torch.as_tensor takes)."""
    K = get_kernels()
    dev = y_pred.device if isinstance(y_pred, torch.Tensor) else (y_true.device if isinstance(y_true, torch.Tensor) else 'cuda')
    yt = torch.as_tensor(np.asarray(y_true) if not isinstance(y_true, torch.Tensor) else y_true).to(dev)
    yp = torch.as_tensor(np.asarray(y_pred) if not isinstance(y_pred, torch.Tensor) else y_pred).to(dev)
    f1, valid = K.f1_at_k(yt.reshape(-1, yt.shape[-1]), yp.reshape(-1, yp.shape[-1]), num_classes, overlap, ignore_value)
    return float(f1.sum() / valid.sum())  # ZeroDivisionError-equivalent: nan when nothing is valid (reference raises)


def evaluate_f1_at_k(targets: dict, outputs: dict, num_subactivities, num_affordances, overlap: float = 0.25):
    """predict.py:229-246: {index: labels (N, T) or (N, T, E)} -> {index: F1@overlap}."""
    results = {}
    for index, target in sorted(targets.items()):
        output = outputs[index]
        if target.ndim == 3:
            target, output = target.transpose(1, 2), output.transpose(1, 2)
        steps = output.shape[-1]
        num_classes = num_affordances if 'affordance' in str(index) else num_subactivities
        results[index] = f1_at_k(target.reshape(-1, steps), output.reshape(-1, steps), num_classes, overlap=overlap,
                                 ignore_value=-1.0)
    return results
