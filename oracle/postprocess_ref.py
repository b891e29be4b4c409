"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's inference post-processing for the 2G-GCN outputs, pinned to outputs of the
reference itself (golden G8, tools/make_golden.py):
  * predict_labels: predict.py:64-70 (repeat_interleave by the downsampling factor along time, match_shape :95-116)
    followed by process_output's np.argmax over classes (:195-201);
  * f1_at_k / f1_at_k_single_example: pyrutils/metrics.py:7-81 with run-length encoding from
    pyrutils/itertools.py:15-18 and pyrutils/utils.py:38-42.
"""
from itertools import groupby

import numpy as np


def predict_labels(logp: np.ndarray, downsampling: int, target_steps: int) -> np.ndarray:
    """logp (bs, C, T, E) -> labels (bs, target_steps, E)."""
    out = logp
    if downsampling > 1:
        out = np.repeat(out, downsampling, axis=-2)                 # predict.py:68
        steps = out.shape[-2]
        if steps >= target_steps:                                   # match_shape, predict.py:107-110
            out = out[:, :, :target_steps]
        else:                                                       # :111-115 pad with the last step
            pad = np.repeat(out[:, :, -1:], target_steps - steps, axis=-2)
            out = np.concatenate([out, pad], axis=-2)
    return np.argmax(out, axis=1)                                   # predict.py:199


def _rle(seq):
    ids, lengths = [], []
    for k, v in groupby(seq):                                       # pyrutils/itertools.py:17-18
        ids.append(k)
        lengths.append(len(list(v)))
    starts = np.concatenate([[0], np.cumsum(lengths)])              # pyrutils/utils.py:40-42
    return np.array(ids), np.stack([starts[:-1], starts[1:]], 1)


def f1_at_k_single_example(y_true, y_pred, num_classes: int, overlap: float) -> float:
    """pyrutils/metrics.py:7-65."""
    tgt_ids, tgt_iv = _rle(list(y_true))
    out_ids, out_iv = _rle(list(y_pred))
    tp = fp = 0.0
    used = np.zeros(len(tgt_ids))
    for (o0, o1), oid in zip(out_iv, out_ids):
        inter = np.minimum(o1, tgt_iv[:, 1]) - np.maximum(o0, tgt_iv[:, 0])
        union = np.maximum(o1, tgt_iv[:, 1]) - np.minimum(o0, tgt_iv[:, 0])
        iou = (inter / union) * (oid == tgt_ids)
        idx = int(np.argmax(iou))
        if oid >= num_classes:
            continue
        if iou[idx] >= overlap and not used[idx]:
            tp += 1
            used[idx] = 1
        else:
            fp += 1
    fn = len(used) - used.sum()
    precision = tp / (tp + fp) if tp + fp > 0 else 0.0
    recall = tp / (tp + fn) if tp + fn > 0 else 0.0
    return 2 * precision * recall / (precision + recall) if precision + recall > 0 else 0.0


def f1_at_k(y_true, y_pred, num_classes: int, overlap: float, ignore_value=None) -> float:
    """pyrutils/metrics.py:68-81."""
    total, n = 0.0, 0.0
    for y_t, y_p in zip(y_true, y_pred):
        y_t, y_p = np.array(y_t), np.array(y_p)
        if ignore_value is not None:
            keep = y_t != ignore_value
            y_t, y_p = y_t[keep], y_p[keep]
        if y_t.size == 0:
            continue
        total += f1_at_k_single_example(y_t, y_p, num_classes, overlap)
        n += 1
    return total / n
