"""Per-clip batching for the 2G-GCN path: counterpart of the batching half of the reference's ``vhoi/data_loading.py``
(``create_data_loader`` :362-379, ``assemble_tensors`` :436-471, ``assemble_bimanual_tensors`` :480-501,
``assemble_mphoi_tensors`` :504-522, the per-dataset human/object assemblers :562-982, distances :985-1203,
``assemble_num_steps`` :1206, fetcher/feeder :1215-1315, ``determine_num_classes`` :1318, ``input_size_from_data_loader``
:1332), and of the MPHOI-72 / Bimanual-Actions dataset readers (``load_mphoi_training_data`` :118-160,
``load_mphoi_testing_data`` :285-309, ``load_bimanual_training_data`` :63-115, ``load_bimanual_testing_data`` :234-282,
``split_train_test`` :353-359) over the zarr feature stores, read natively by ``featstore`` (no zarr dependency). The
CAD-120 readers (:23-60, :201-231: pickled ``CAD120Video`` objects) are out of scope: start from the in-memory ``data``
list they produce.

Same outputs as the reference (tuple slot order of SURVEY.md Appendix B, NaN padding to the split maximum then
``nan_to_num``, float32 / int64 dtypes), but assembled with whole-video numpy operations instead of the reference's
per-frame Python loops, and with a double-buffered pinned / HBM-resident input pipeline (``DevicePrefetcher``,
SURVEY.md section 8f row 2) in place of the reference's synchronous pageable copies.
"""
import json
import random
from functools import partial
from itertools import groupby
from typing import Optional

import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset

from . import featstore

# layout constants per dataset (reference lines in comments)
_SPECS = {
    'mphoi': dict(scale=1000.0, keypoints=[1, 2, 4, 6, 7, 11, 13, 14, 27], max_objects=4,  # :784-809
                  keys=('Human1', 'Human2'), dims=(3840.0, 2160.0)),
    'bimanual': dict(scale=100.0, keypoints=[0, 4, 8, 12, 16, 20], max_objects=9,  # :668-693
                     keys=('left_hand', 'right_hand'), dims=(640.0, 480.0)),
}


def _ds(x, d):
    return x[d - 1::d]


def _pos_vel(points):
    """(L, K, 2) positions -> (L, K*4): per point (x, y, vx, vy) with v = (next - cur)*100, 0 at the last frame."""
    points = np.asarray(points, dtype=np.float64)  # the reference's intermediates are float64 (np.zeros padding)
    vel = np.zeros_like(points)
    vel[:-1] = (points[1:] - points[:-1]) * 100
    return np.concatenate([points, vel], axis=-1).reshape(points.shape[0], -1)


def _boxes_as_points(obb, max_objects):
    """(L, n, 4) boxes -> (L, 2*max_objects, 2) corner points, zero rows for missing objects (:790-803)."""
    L, n = obb.shape[0], obb.shape[1]
    b = np.zeros((L, max_objects, 4), dtype=obb.dtype)
    b[:, :n] = obb
    return b.reshape(L, 2 * max_objects, 2)


def run_length_encoding(seq):
    for k, v in groupby(seq):
        yield k, len(list(v))


def _next_labels(y):
    rle = list(run_length_encoding(y))
    out = []
    for (_, prev_len), (nxt, _) in zip(rle[:-1], rle[1:]):
        out += [nxt] * prev_len
    return out


def segmentation_from_output_class(y, segmentation_type='input'):
    """End-of-segment flags from per-frame labels (reference behaviour: vhoi/data_loading.py:885-896): a frame ends a
    segment when the next frame carries another label -- a missing (-1) neighbour counts as another label -- and the last
    frame of every row always does. Missing frames themselves read 1.0 ('input' style) or -1.0 ('output' style)."""
    labels = np.asarray(y)
    missing = labels == -1
    ends = np.ones(labels.shape, dtype=bool)                       # last column: always an end
    ends[:, :-1] = np.diff(labels.astype(np.float64), axis=1) != 0
    ends[:, :-1] |= missing[:, 1:] | missing[:, :-1]
    seg = ends.astype(np.float32)
    seg[missing] = -1.0 if segmentation_type == 'output' else 1.0
    return seg


def ignore_last_step_end_flag(x):
    """:524-533: clears the last end flag of every example. x (num_examples, num_steps)."""
    for m in range(x.shape[0]):
        idx = np.nonzero(x[m] == 1.0)[0]
        if len(idx):
            x[m, idx[-1]] = 0.0
    return x


def ignore_last_step_end_flag_general(x):
    for e in range(x.shape[-1]):
        x[:, :, e] = ignore_last_step_end_flag(x[:, :, e])
    return x


def smooth_segmentation(x, sigma: float):
    """Budget targets (reference behaviour: vhoi/data_loading.py:544-559): the 0/1 end flags blurred along time with a
    unit-mass Gaussian (zero beyond the clip ends), rescaled by 2.5 sigma and clipped to [0, 1]; missing (-1) frames
    contribute nothing and stay -1 in the result. sigma == 0: unchanged. Bit-compatible with the reference: the product is
    evaluated as (blurred * 2.5) * sigma in the array's precision, and -- as there -- the missing entries of the CALLER'S
    array are zeroed in place (a caller that reuses its array sees 0, not -1, at those frames)."""
    if not sigma:
        return x
    from scipy.ndimage import gaussian_filter1d
    missing = x == -1.0
    x[missing] = 0.0
    out = np.clip(gaussian_filter1d(x, sigma=sigma, axis=1, mode='constant') * 2.5 * sigma, 0.0, 1.0)
    out[missing] = -1.0
    return out


def compute_centroid(bb):
    return np.concatenate([(bb[..., :1] + bb[..., 2:3]) / 2, (bb[..., 1:2] + bb[..., 3:4]) / 2], axis=-1)


def _pad_stack(arrays, shape_tail, fill=np.nan, dtype=np.float32):
    out = np.full([len(arrays)] + list(shape_tail), fill_value=fill, dtype=dtype)
    for m, a in enumerate(arrays):
        out[(m,) + tuple(slice(0, s) for s in a.shape)] = a
    return out


# ---------------------------------------------------------------------------------------------------------------
# two-human datasets (MPHOI :769-882, Bimanual :653-766); data item:
#   [h1_feat (L,2048), h2_feat, objects (L,n,F), ground_truth dict, h1_bb, h2_bb, objects_bb (L,n,4), h1_pose (L,J,2), h2_pose]
# ---------------------------------------------------------------------------------------------------------------
def assemble_two_human_frame_level_recurrent_human(data, dataset: str, downsampling: int = 1, test_data: bool = False):
    sp = _SPECS[dataset]
    kp, mo = sp['keypoints'], sp['max_objects']
    xs, max_len = [], 0
    for h1, h2, _, _, _, _, obb, h1p, h2p in data:
        max_len = max(max_len, h1.shape[0])
        h1, h2 = _ds(h1, downsampling), _ds(h2, downsampling)
        p1 = _ds(h1p, downsampling)[:, kp] / sp['scale']
        p2 = _ds(h2p, downsampling)[:, kp] / sp['scale']
        ob = _boxes_as_points(_ds(obb, downsampling) / sp['scale'], mo)
        context = np.concatenate([_pos_vel(p1), _pos_vel(p2), _pos_vel(ob)], axis=-1)  # (L, 4N)
        xs.append(np.stack([np.concatenate([h1, context], -1), np.concatenate([h2, context], -1)], axis=1))
    T = max(x.shape[0] for x in xs)
    x_hs = _pad_stack(xs, [T, 2, xs[0].shape[-1]])
    y_rec = np.full([len(xs), max_len, 2], fill_value=-1, dtype=np.int64)
    y_pred = np.full_like(y_rec, fill_value=-1)
    for m, item in enumerate(data):
        gt = item[3]
        for e, key in enumerate(sp['keys']):
            y = gt[key]
            y_rec[m, :len(y), e] = y
            yp = _next_labels(y)
            y_pred[m, :len(yp), e] = yp
    x_seg = segmentation_from_output_class(y_rec[:, downsampling - 1::downsampling], segmentation_type='input')
    if not test_data:
        y_rec, y_pred = y_rec[:, downsampling - 1::downsampling], y_pred[:, downsampling - 1::downsampling]
    y_seg = segmentation_from_output_class(y_rec, segmentation_type='output')
    return [x_hs, x_seg], [y_rec, y_pred, y_seg]


def assemble_two_human_frame_level_recurrent_objects(data, downsampling: int = 1):
    objs = [_ds(item[2], downsampling) for item in data]
    T = max(o.shape[0] for o in objs)
    O = max(o.shape[1] for o in objs)
    x_objects = _pad_stack(objs, [T, O, objs[-1].shape[-1]])
    mask = np.zeros([len(objs), O], dtype=np.float32)
    for m, o in enumerate(objs):
        mask[m, :o.shape[1]] = 1.0
    return [x_objects, mask]


def _two_human_distances(data, dataset, downsampling):
    dims = np.array(_SPECS[dataset]['dims'], dtype=np.float32)
    hh, ho1, ho2, oo = [], [], [], []
    for item in data:
        c1 = compute_centroid(_ds(item[4], downsampling)) / dims
        c2 = compute_centroid(_ds(item[5], downsampling)) / dims
        co = compute_centroid(_ds(item[6], downsampling)) / dims  # (L, n, 2)
        hh.append(np.linalg.norm(c1 - c2, ord=2, axis=-1))
        ho1.append(np.linalg.norm(co - c1[:, None], ord=2, axis=-1))
        ho2.append(np.linalg.norm(co - c2[:, None], ord=2, axis=-1))
        oo.append(np.linalg.norm(co[:, None, :, :] - co[:, :, None, :], ord=2, axis=-1))
    T = max(len(d) for d in hh)
    O = max(d.shape[1] for d in ho1)
    x_hh = np.full([len(hh), T, 2, 2], np.nan, dtype=np.float32)
    x_ho = np.full([len(hh), T, 2, O], np.nan, dtype=np.float32)
    for m in range(len(hh)):
        L = len(hh[m])
        x_hh[m, :L, 0, 1] = hh[m]
        x_hh[m, :L, 1, 0] = hh[m]
        x_hh[m, :L, 0, 0] = 0.0
        x_hh[m, :L, 1, 1] = 0.0
        x_ho[m, :L, 0, :ho1[m].shape[1]] = ho1[m]
        x_ho[m, :L, 1, :ho2[m].shape[1]] = ho2[m]
    x_oo = _pad_stack(oo, [T, O, O])
    return x_hh, x_ho, x_oo


def assemble_num_steps(data, downsampling: int = 1):
    return np.array([len(_ds(item[0], downsampling)) for item in data], dtype=np.float32)


def _assemble_two_human_tensors(data, dataset, model_name, sigma=0.0, downsampling=1, test_data=False):
    if model_name != '2G-GCN':
        raise ValueError(f'{dataset} code not implemented for {model_name} yet.')
    xs, ys = assemble_two_human_frame_level_recurrent_human(data, dataset, downsampling, test_data)
    xs_objects = assemble_two_human_frame_level_recurrent_objects(data, downsampling)
    if sigma:
        ys[2] = ignore_last_step_end_flag_general(ys[2])
    ys[2] = smooth_segmentation(ys[2], sigma)
    ys_budget = ys[2]
    x_hh, x_ho, x_oo = _two_human_distances(data, dataset, downsampling)
    xs = xs[:1] + xs_objects + xs[1:] + [x_hh, x_ho, x_oo, assemble_num_steps(data, downsampling)]
    ys = [ys_budget] + ys[2:] + ys[:2]
    ys += ys[-2:]
    return xs, ys


def assemble_mphoi_tensors(data, model_name, sigma=0.0, downsampling=1, test_data=False):
    return _assemble_two_human_tensors(data, 'mphoi', model_name, sigma, downsampling, test_data)


def assemble_bimanual_tensors(data, model_name, sigma=0.0, downsampling=1, test_data=False):
    return _assemble_two_human_tensors(data, 'bimanual', model_name, sigma, downsampling, test_data)


# ---------------------------------------------------------------------------------------------------------------
# CAD-120 (:562-650, :899-941); data item:
#   [human_feat (L,2048), object_feat (L,n,F), skeleton_bb (L,4), objects_bb (L,n,4), skeleton_pose (L,9,2), segments]
# ---------------------------------------------------------------------------------------------------------------
def assemble_tensors(data, model_name, model_input_type='multiple', sigma=0.0, downsampling=1, test_data=False):
    if model_name != '2G-GCN':
        raise ValueError(f'{model_name} is not an option for model name.')
    xs_h, max_len = [], 0
    for hf, _, _, obb, pose, _ in data:
        max_len = max(max_len, hf.shape[0])
        p = _ds(pose, downsampling) / 100
        ob = _boxes_as_points(_ds(obb, downsampling) / 100, 5)
        xs_h.append(np.concatenate([_ds(hf, downsampling), _pos_vel(p), _pos_vel(ob)], axis=-1))
    T = max(x.shape[0] for x in xs_h)
    x_human = _pad_stack(xs_h, [T, xs_h[-1].shape[-1]])
    objs = [_ds(item[1], downsampling) for item in data]
    O = max(o.shape[1] for o in objs)
    x_objects = _pad_stack(objs, [T, O, objs[-1].shape[-1]])
    mask = np.zeros([len(objs), O], dtype=np.float32)
    for m, o in enumerate(objs):
        mask[m, :o.shape[1]] = 1.0
    M = len(data)
    y_rec_h = np.full([M, max_len], -1, dtype=np.int64)
    y_pred_h = np.full_like(y_rec_h, -1)
    y_rec_o = np.full([M, max_len, O], -1, dtype=np.int64)
    y_pred_o = np.full_like(y_rec_o, -1)
    for m, item in enumerate(data):
        for seg in item[5]:
            if seg.start_frame is None or seg.end_frame is None:
                continue
            s, e = seg.start_frame - 1, seg.end_frame - 1
            y_rec_h[m, s:e + 1] = seg.subactivity - 1
            y_pred_h[m, s:e + 1] = seg.next_subactivity - 1 if seg.next_subactivity is not None else -1
            for oid, aff in seg.object_affordance.items():
                y_rec_o[m, s:e + 1, oid - 1] = aff - 1
            for oid, aff in seg.next_object_affordance.items():
                y_pred_o[m, s:e + 1, oid - 1] = aff - 1
    d = downsampling
    x_seg_h = segmentation_from_output_class(y_rec_h[:, d - 1::d], 'input')
    x_seg_o = segmentation_from_output_class(y_rec_o[:, d - 1::d], 'input')
    if not test_data:
        y_rec_h, y_pred_h, y_rec_o, y_pred_o = (y[:, d - 1::d] for y in (y_rec_h, y_pred_h, y_rec_o, y_pred_o))
    y_seg_h = segmentation_from_output_class(y_rec_h, 'output')
    y_seg_o = segmentation_from_output_class(y_rec_o, 'output')
    if sigma:
        y_seg_h = ignore_last_step_end_flag(y_seg_h)
        y_seg_o = ignore_last_step_end_flag_general(y_seg_o)
    y_seg_h = smooth_segmentation(y_seg_h, sigma)
    y_seg_o = smooth_segmentation(y_seg_o, sigma)
    # distances (:1019-1040, :1132-1153): objects are NOT divided by the image size there, the skeleton is
    cad_dims = np.array([640, 480], dtype=np.float32)
    ho, oo = [], []
    for _, _, sbb, obb, _, _ in data:
        co = compute_centroid(_ds(obb, d))
        cs = compute_centroid(_ds(sbb, d)) / cad_dims
        ho.append(np.linalg.norm(co - cs[:, None], ord=2, axis=-1)[:, None, :])
        oo.append(np.linalg.norm(co[:, None, :, :] - co[:, :, None, :], ord=2, axis=-1))
    x_ho = _pad_stack(ho, [T, 1, O])
    x_oo = _pad_stack(oo, [T, O, O])
    ex = lambda a: np.expand_dims(a, axis=2)  # "fake" human dimension (:474-477)
    xs = [ex(x_human), x_objects, mask, ex(x_seg_h), x_seg_o, x_ho, x_oo, assemble_num_steps(data, d)]
    ys = [ex(y_seg_h), y_seg_o, ex(y_seg_h), y_seg_o, ex(y_rec_h), ex(y_pred_h), y_rec_o, y_pred_o,
          ex(y_rec_h), ex(y_pred_h), y_rec_o, y_pred_o]
    return xs, ys


def assemble_cad120_segmentations_from_frame_level_features(data):
    """(start, end) frame pairs per video (:389-400; the reference forgets the return statement there)."""
    out = []
    for item in data:
        out.append([(s.start_frame - 1, s.end_frame - 1) for s in item[5]
                    if s.start_frame is not None and s.end_frame is not None])
    return out


# ---------------------------------------------------------------------------------------------------------------
def scale_array(x, scaler=None, scaling_strategy='standard'):
    from sklearn.preprocessing import StandardScaler
    shape = x.shape
    x = x.reshape(-1, shape[-1])
    if scaler is None:
        if scaling_strategy != 'standard':
            raise ValueError(f'scaling_strategy must be standard and not {scaling_strategy}.')
        scaler = StandardScaler().fit(x)
    return scaler.transform(x).reshape(*shape), scaler


def maybe_scale_input_tensors(x, model_name, scaling_strategy=None, scalers=None):
    if not scalers:
        scalers = {}
        if scaling_strategy is None:
            return x, scalers
    xh, hs = scale_array(x[0], scaler=scalers.get('human_scaler'), scaling_strategy=scaling_strategy)
    xo, os_ = scale_array(x[1], scaler=scalers.get('object_scaler'), scaling_strategy=scaling_strategy)
    return [xh, xo] + x[2:], {'human_scaler': hs, 'object_scaler': os_}


class LengthBucketedBatchSampler:
    """Opt-in batch sampler: clips are ordered by their number of valid steps and cut into
    consecutive batches, so a batch holds clips of similar length and can be trimmed to ITS longest clip instead of the
    split's (`trim_to_batch_length`). With `shuffle` the ORDER of the batches is permuted per epoch (the membership of a
    batch is fixed by the lengths). This is NOT the reference's batching (a plain shuffled DataLoader over tensors
    padded to the split maximum, vhoi/data_loading.py:362-379) and it changes the numbers: the model has no length
    masking, its backward-direction GRUs start on the padding and its forced last-step segment end sits on the padded
    T -- hence off by default."""

    def __init__(self, num_steps, batch_size: int, shuffle: bool = False, generator=None, drop_last: bool = False):
        steps = torch.as_tensor(num_steps).flatten()
        order = torch.argsort(steps, stable=True).tolist()
        self.batches = [order[i:i + batch_size] for i in range(0, len(order), batch_size)]
        if drop_last and self.batches and len(self.batches[-1]) < batch_size:
            self.batches.pop()
        self.shuffle, self.generator = shuffle, generator

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        if self.shuffle:
            for i in torch.randperm(len(self.batches), generator=self.generator).tolist():
                yield self.batches[i]
        else:
            yield from self.batches


def trim_to_batch_length(tensors, steps_slot: int = 7):
    """Cuts the time axis (dim 1) of every tensor that runs over the input frames down to the longest clip of the batch
    (slot `steps_slot` = steps_per_example, Appendix B). Tensors with another time length (the full-rate targets of
    test_data=True) are left alone; predict.py's match_shape copes with them as before."""
    t_in = tensors[0].shape[1]
    t_b = max(1, min(t_in, int(tensors[steps_slot].max().item())))
    return [t[:, :t_b].contiguous() if (t.dim() >= 2 and t.shape[1] == t_in) else t for t in tensors]


def _trim_collate(samples):
    return trim_to_batch_length([torch.stack(col, 0) for col in zip(*samples)])


def create_data_loader(data, model_name: str, model_input_type: str, dataset_name: str, batch_size: int, shuffle: bool,
                       scaling_strategy: Optional[str] = None, scalers: Optional[dict] = None, sigma: float = 0.0,
                       downsampling: int = 1, test_data: bool = False, pin_memory: bool = False,
                       length_bucketing: bool = False):
    """vhoi/data_loading.py:362-379 (plus an opt-in pinned-memory dataset for asynchronous H2D copies, and the opt-in
    `length_bucketing`: batches of similar-length clips trimmed to their own longest clip -- see
    LengthBucketedBatchSampler for why it is not the default)."""
    name = dataset_name.lower()
    if name == 'cad120':
        x, y = assemble_tensors(data, model_name, model_input_type, sigma, downsampling, test_data)
    elif name == 'mphoi':
        x, y = assemble_mphoi_tensors(data, model_name, sigma, downsampling, test_data)
    else:
        x, y = assemble_bimanual_tensors(data, model_name, sigma, downsampling, test_data)
    x, scalers = maybe_scale_input_tensors(x, model_name, scaling_strategy=scaling_strategy, scalers=scalers)
    x = [np.nan_to_num(ix, copy=False, nan=0.0) for ix in x]
    tensors = [torch.from_numpy(np.ascontiguousarray(a)) for a in x + y]
    if pin_memory and torch.cuda.is_available():
        tensors = [t.pin_memory() for t in tensors]
    if length_bucketing:
        sampler = LengthBucketedBatchSampler(tensors[7], batch_size, shuffle=shuffle)
        loader = DataLoader(TensorDataset(*tensors), batch_sampler=sampler, num_workers=0, pin_memory=False,
                            collate_fn=_trim_collate)
    else:
        loader = DataLoader(TensorDataset(*tensors), batch_size=batch_size, shuffle=shuffle, num_workers=0,
                            pin_memory=False, drop_last=False)
    segmentations = assemble_cad120_segmentations_from_frame_level_features(data) if name == 'cad120' else None
    return loader, scalers, segmentations


def split_train_test(training_data: list, test_fraction: float = 0.2, seed: int = 42):
    """:353-359 (seeds and uses the global ``random`` generator exactly as the reference does)."""
    random.seed(seed)
    random.shuffle(training_data)
    n_test = round(len(training_data) * test_fraction)
    return training_data[n_test:], training_data[:n_test]


# per-video record layout of the two-entity datasets: [entity-1 features, entity-2 features, object features,
# ground truth, entity-1 boxes, entity-2 boxes, object boxes, entity-1 pose, entity-2 pose]; (store, array) per slot
_STORE_SLOTS = {
    'mphoi': (('feat', 'Human1'), ('feat', 'Human2'), ('feat', 'objects'), None, ('hbb', 'Human1'), ('hbb', 'Human2'),
              ('obb', 'objects'), ('hps', 'Human1'), ('hps', 'Human2')),                              # :134-143
    'bimanual': (('feat', 'left_hand'), ('feat', 'right_hand'), ('feat', 'objects'), None, ('bbs', 'left_hand'),
                 ('bbs', 'right_hand'), ('bbs', 'objects'), ('hps', 'left_hand'), ('hps', 'right_hand')),  # :80-87
}


def _read_videos(dataset, data_path, store_paths, keep, video_id_to_video_fps=None):
    """Ground-truth JSON + feature stores -> list of per-video records (and their ids), in the JSON's order."""
    with open(data_path, mode='rb') as f:
        ground_truth = json.load(f)
    roots = {k: featstore.open(p, mode='r') for k, p in store_paths.items()}
    records, ids = [], []
    for video_id, gt in ground_truth.items():
        if not keep(video_id):
            continue
        rec = [gt if slot is None else roots[slot[0]][video_id][slot[1]][:] for slot in _STORE_SLOTS[dataset]]
        if video_id_to_video_fps is not None and video_id_to_video_fps[video_id] == 15:
            # some Bimanual videos were recorded at 15 FPS: every frame (and label) is doubled (:89-99)
            rec = [r if i == 3 else np.repeat(r, repeats=2, axis=0) for i, r in enumerate(rec)]
            for k in ('left_hand', 'right_hand'):
                gt[k] = np.repeat(gt[k], repeats=2, axis=0)
        records.append(rec)
        ids.append(video_id)
    return records, ids


def _training_loaders(records, dataset, model_name, model_input_type, batch_size, val_fraction, seed, debug,
                      scaling_strategy, sigma, downsampling):
    training_data, val_data = split_train_test(records, test_fraction=val_fraction, seed=seed)
    if debug:
        training_data, val_data = training_data[:4], val_data[:1]
    train_loader, scalers, _ = create_data_loader(training_data, model_name, model_input_type, dataset,
                                                  batch_size=batch_size, shuffle=True,
                                                  scaling_strategy=scaling_strategy, sigma=sigma,
                                                  downsampling=downsampling, test_data=False)
    val_loader, _, _ = create_data_loader(val_data, model_name, model_input_type, dataset, batch_size=len(val_data),
                                          shuffle=False, scalers=scalers, sigma=sigma, downsampling=downsampling,
                                          test_data=False)
    data_info = {'input_size': input_size_from_data_loader(train_loader, model_name, model_input_type)}
    return train_loader, val_loader, data_info, scalers


def _testing_loader(records, ids, dataset, model_name, model_input_type, batch_size, scalers, downsampling):
    test_loader, _, segmentations = create_data_loader(records, model_name, model_input_type, dataset,
                                                       batch_size=batch_size, shuffle=False, scalers=scalers,
                                                       downsampling=downsampling, test_data=True)
    data_info = {'input_size': input_size_from_data_loader(test_loader, model_name, model_input_type)}
    return test_loader, data_info, segmentations, ids


def _mphoi_stores(data_path_zarr, data_path_obbs_zarr, data_path_hbbs_zarr, data_path_hps_zarr):
    return {'feat': data_path_zarr, 'obb': data_path_obbs_zarr, 'hbb': data_path_hbbs_zarr, 'hps': data_path_hps_zarr}


def load_mphoi_training_data(data_path, data_path_zarr, data_path_obbs_zarr, data_path_hbbs_zarr, data_path_hps_zarr,
                             model_name: str, model_input_type: str, test_subject_id, batch_size: int = 8,
                             val_fraction: float = 0.2, seed: int = 42, debug: bool = False, scaling_strategy=None,
                             sigma: float = 0.0, downsampling: int = 1):
    """:118-160. A video trains unless one of its two subjects is one of the two held-out subjects (the last two
    digits of 'SubjectAB' name the pair)."""
    test_pair = {int(test_subject_id[-2]), int(test_subject_id[-1])}

    def keep(video_id):
        subject = video_id.split(sep='-')[0]
        return not ({int(subject[-2]), int(subject[-1])} & test_pair)

    records, _ = _read_videos('mphoi', data_path, _mphoi_stores(data_path_zarr, data_path_obbs_zarr,
                                                                data_path_hbbs_zarr, data_path_hps_zarr), keep)
    return _training_loaders(records, 'mphoi', model_name, model_input_type, batch_size, val_fraction, seed, debug,
                             scaling_strategy, sigma, downsampling)


def load_mphoi_testing_data(data_path, data_path_zarr, data_path_obbs_zarr, data_path_hbbs_zarr, data_path_hps_zarr,
                            model_name: str, model_input_type: str, test_subject_id, batch_size: int,
                            scalers: Optional[dict] = None, downsampling: int = 1):
    """:285-309: the videos whose subject field equals ``test_subject_id``."""
    records, ids = _read_videos('mphoi', data_path, _mphoi_stores(data_path_zarr, data_path_obbs_zarr,
                                                                  data_path_hbbs_zarr, data_path_hps_zarr),
                                lambda video_id: video_id.split(sep='-')[0] == test_subject_id)
    return _testing_loader(records, ids, 'mphoi', model_name, model_input_type, batch_size, scalers, downsampling)


def _bimanual_subject(video_id):
    return int(video_id.split(sep='-')[0].split(sep='_')[1])


def load_bimanual_training_data(data_path, data_path_zarr, data_path_bbs_zarr, data_path_hps_zarr, model_name: str,
                                model_input_type: str, test_subject_id: int, video_id_to_video_fps: dict,
                                batch_size: int = 8, val_fraction: float = 0.2, seed: int = 42, debug: bool = False,
                                scaling_strategy=None, sigma: float = 0.0, downsampling: int = 1):
    """:63-115."""
    stores = {'feat': data_path_zarr, 'bbs': data_path_bbs_zarr, 'hps': data_path_hps_zarr}
    records, _ = _read_videos('bimanual', data_path, stores, lambda v: _bimanual_subject(v) != test_subject_id,
                              video_id_to_video_fps)
    return _training_loaders(records, 'bimanual', model_name, model_input_type, batch_size, val_fraction, seed, debug,
                             scaling_strategy, sigma, downsampling)


def load_bimanual_testing_data(data_path, data_path_zarr, data_path_bbs_zarr, data_path_hps_zarr, model_name: str,
                               model_input_type: str, test_subject_id: int, video_id_to_video_fps: dict,
                               batch_size: int, scalers: Optional[dict] = None, downsampling: int = 1):
    """:234-282."""
    stores = {'feat': data_path_zarr, 'bbs': data_path_bbs_zarr, 'hps': data_path_hps_zarr}
    records, ids = _read_videos('bimanual', data_path, stores, lambda v: _bimanual_subject(v) == test_subject_id,
                                video_id_to_video_fps)
    return _testing_loader(records, ids, 'bimanual', model_name, model_input_type, batch_size, scalers, downsampling)


def gcn_fetcher(dataset, device, non_blocking: bool = False, **kwargs):
    """:1282-1315: moves only the tensors the configuration needs; the others stay on the host."""
    to = lambda t: t.to(device, non_blocking=non_blocking)
    data = [to(dataset[0]), to(dataset[1]), to(dataset[2])]
    data.append(to(dataset[3]) if kwargs.get('input_human_segmentation', False) else dataset[3])
    dist = kwargs.get('make_attention_distance_based', False)
    if kwargs.get('dataset_name', 'cad120') == 'cad120':
        data.append(to(dataset[4]) if kwargs.get('input_object_segmentation', False) else dataset[4])
        data += [to(dataset[5]), to(dataset[6])] if dist else [dataset[5], dataset[6]]
    else:
        data += [to(dataset[4]), to(dataset[5]), to(dataset[6])] if dist else [dataset[4], dataset[5], dataset[6]]
    targets = [to(t) for t in dataset[8:]]
    data.append(to(dataset[7]))
    return data, targets


class DevicePrefetcher:
    """Double-buffered host -> HBM input pipeline around a DataLoader and a fetcher.

    The reference copies every batch synchronously from pageable memory inside the step (vhoi/data_loading.py:376,
    :1284-1314: ``tensor.to(device)`` per slot). Here batch i+1 is staged while batch i computes: each host tensor
    the fetcher would move is first copied into one of two persistent PINNED staging slots, then sent with a
    non-blocking copy on a side HIP stream; the consumer's stream waits on the copy's event only when it receives the
    batch, and the device tensors are tied to the consumer stream (``record_stream``) so the allocator cannot recycle
    them early. With ``resident=True`` the whole split is moved to HBM once (the real datasets are a few GB; 288 GB
    of HBM3E) and batches become device-side index selections: no per-step PCIe traffic at all.

    Iterating yields exactly what ``fetch(batch, device=...)`` yields for every batch of ``loader``, in loader order.
    On a CPU device it degenerates to that plain loop (used by the CPU tests of the host logic)."""

    def __init__(self, loader, fetch, device, resident: bool = False, **fetch_kwargs):
        self.loader, self.fetch, self.device, self.kw = loader, fetch, torch.device(device), fetch_kwargs
        self.on_gpu = self.device.type == 'cuda'
        self.copy_stream = torch.cuda.Stream(self.device) if self.on_gpu else None
        self._slots = [{}, {}]  # pinned staging buffers of the two in-flight batches, keyed by tuple position
        self._slot_events = [None, None]
        self._resident = None
        if resident:
            ds = loader.dataset
            if not isinstance(ds, TensorDataset):
                raise TypeError('resident=True needs a TensorDataset (create_data_loader builds one)')
            self._resident = TensorDataset(*[t.to(self.device) for t in ds.tensors])

    def __len__(self):
        return len(self.loader)

    def _slot_view(self, slot, i, shape, dtype):
        n = 1
        for d in shape:
            n *= d
        buf = self._slots[slot].get(i)
        if buf is None or buf.dtype != dtype or buf.numel() < n:
            buf = torch.empty(max(n, 1), dtype=dtype).pin_memory()
            self._slots[slot][i] = buf
        return buf[:n].view(shape)

    def _host_batches(self):
        """Host-side batches. For a TensorDataset the rows of a batch are gathered straight into the pinned staging
        slot (one host copy; DataLoader's collate + pin would make two); otherwise the loader's own batches are
        copied into the slot."""
        ds = self.loader.dataset
        if self.on_gpu and isinstance(ds, TensorDataset) and self.loader.batch_sampler is not None and \
                self.loader.collate_fn is not _trim_collate:
            for k, idx in enumerate(self._batch_indices()):
                slot = k & 1
                self._wait_slot_free(slot)
                idx = torch.as_tensor(idx, dtype=torch.int64)
                yield [torch.index_select(t, 0, idx, out=self._slot_view(slot, i, (len(idx),) + tuple(t.shape[1:]), t.dtype))
                       for i, t in enumerate(ds.tensors)]
        else:
            for k, batch in enumerate(self.loader):
                if self.on_gpu:
                    slot = k & 1
                    self._wait_slot_free(slot)
                    staged = []
                    for i, t in enumerate(batch):
                        if isinstance(t, torch.Tensor) and not t.is_cuda and not t.is_pinned():
                            v = self._slot_view(slot, i, tuple(t.shape), t.dtype)
                            v.copy_(t)
                            t = v
                        staged.append(t)
                    batch = staged
                yield batch

    def _batch_indices(self):
        """The loader's own batch order (shuffled or sequential, drop_last honoured). A DataLoader iterator draws its
        worker base seed from the loader's generator before the sampler draws the permutation; the same draw is made
        here so that a seeded shuffling loader visits the clips in the same order with and without the prefetcher."""
        torch.empty((), dtype=torch.int64).random_(generator=self.loader.generator)
        return iter(self.loader.batch_sampler)

    def _wait_slot_free(self, slot):
        """A pinned slot is rewritten two batches after the copy that read it was issued: wait for that copy."""
        ev = self._slot_events[slot]
        if ev is not None:
            ev.synchronize()

    def _stage(self, batch, k):
        if not self.on_gpu:
            return self.fetch(batch, device=self.device, **self.kw), None
        with torch.cuda.stream(self.copy_stream):
            out = self.fetch(batch, device=self.device, non_blocking=True, **self.kw)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._slot_events[k & 1] = ev
        # tensors the fetcher leaves on the host are still views of this batch's pinned staging slot, which is rewritten
        # two batches later: hand out private copies (these are the small ones: segmentations, distances, targets), so a
        # consumer may keep them across iterations -- "exactly what fetch yields" includes their lifetime
        slots = {b.untyped_storage().data_ptr() for b in self._slots[k & 1].values()}
        out = type(out)(type(group)(t.clone() if (isinstance(t, torch.Tensor) and not t.is_cuda and
                                                   t.untyped_storage().data_ptr() in slots) else t for t in group)
                        if isinstance(group, (list, tuple)) else group for group in out)
        return out, ev

    def _hand_over(self, staged):
        out, ev = staged
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for group in out:
                for t in group:
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(cur)
        return out

    def _resident_batches(self):
        ds, bs = self._resident, self.loader.batch_size
        for idx in self._batch_indices():
            idx = torch.as_tensor(idx, dtype=torch.int64, device=self.device)
            batch = [t.index_select(0, idx) for t in ds.tensors]
            if self.loader.collate_fn is _trim_collate:   # length-bucketed loader: trim to the batch's longest clip
                batch = trim_to_batch_length(batch)
            yield self.fetch(batch, device=self.device, **self.kw)

    def __iter__(self):
        if self._resident is not None:
            yield from self._resident_batches()
            return
        self._slot_events = [None, None]
        staged = None
        for k, batch in enumerate(self._host_batches()):
            upcoming = self._stage(batch, k)  # batch k is on its way while the consumer still works on batch k-1
            if staged is not None:
                yield self._hand_over(staged)
            staged = upcoming
        if staged is not None:
            yield self._hand_over(staged)


def gcn_forward(model, data, **kwargs):
    """:1233-1279: maps tuple slots to forward() keywords."""
    pattern = kwargs.get('impose_segmentation_pattern', 0)
    if pattern not in (0, 1):
        raise ValueError(f'Segmentation pattern can only be 1, not {pattern}')
    if pattern:
        human_seg = torch.ones(data[0].size()[:-1], dtype=data[0].dtype, device=data[0].device)
    elif kwargs.get('input_human_segmentation', False):
        human_seg = data[3]
    else:
        human_seg = None
    mk = dict(x_human=data[0], x_objects=data[1], objects_mask=data[2], human_segmentation=human_seg)
    hh = ho = oo = None
    if kwargs.get('dataset_name', 'cad120') == 'cad120':
        if pattern:
            obj_seg = torch.ones(data[1].size()[:-1], dtype=data[1].dtype, device=data[1].device)
        elif kwargs.get('input_object_segmentation', False):
            obj_seg = data[4]
        else:
            obj_seg = None
        mk['objects_segmentation'] = obj_seg
        if kwargs.get('make_attention_distance_based', False):
            ho, oo = data[5], data[6]
    elif kwargs.get('make_attention_distance_based', False):
        hh, ho, oo = data[4], data[5], data[6]
    mk.update(human_human_distances=hh, human_object_distances=ho, object_object_distances=oo,
              steps_per_example=data[7], inspect_model=kwargs.get('inspect_model', False))
    return model(**mk)


def select_model_data_fetcher(model_name: str, model_input_type: str, **kwargs):
    return {'2G-GCN': partial(gcn_fetcher, **kwargs)}[model_name]


def select_model_data_feeder(model_name: str, model_input_type: str, **kwargs):
    return {'2G-GCN': partial(gcn_forward, **kwargs)}[model_name]


def determine_num_classes(model_name: str, model_input_type: str, dataset_name: str):
    if dataset_name.lower() == 'bimanual':
        return 14, None
    if dataset_name.lower() == 'mphoi':
        return 13, None
    return 10, 12


def input_size_from_data_loader(data_loader: DataLoader, model_name: str, model_input_type: str):
    if model_name != '2G-GCN':
        raise ValueError(f'{model_name} is not an option for model name.')
    return data_loader.dataset[0][0].size(-1), data_loader.dataset[0][1].size(-1)
