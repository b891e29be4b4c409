"""Drop-in counterpart of the reference's ``vhoi.models`` for the 2G-GCN path (vhoi/models.py:178-1595).

``TGGCN`` takes the same constructor keywords (vhoi/models.py:179-190), registers parameters and buffers under the same
``state_dict`` names and shapes -- including the parameters the reference constructs but never uses (SURVEY.md
Appendix A6), so checkpoints move both ways -- and has the same ``forward`` signature and output list
(vhoi/models.py:584-586, :919-933). The computation itself is ``ops.TGGCNFunction``: hand-written gfx950 kernels behind
one autograd node. torch.nn modules are used here purely as parameter containers (they are never called), built in the
reference's construction order so that a given ``torch.manual_seed`` yields the reference's initial weights.
"""
import math

import torch
import torch.nn as nn

from . import ops

_ACT = {'identity': nn.Identity, 'logsigmoid': nn.LogSigmoid, 'logsoftmax': nn.LogSoftmax, 'relu': nn.ReLU,
        'sigmoid': nn.Sigmoid, 'softmax': nn.Softmax, 'softplus': nn.Softplus, 'tanh': nn.Tanh}


def build_mlp(dims, activations=None, dropout: float = 0.0, bias: bool = True):
    """Parameter container with the layout of the reference's build_mlp (pyrutils/torch/models.py:8-36):
    nn.Sequential(Linear, activation[, Dropout], ...), so Linear i is registered as '<2i>.weight' / '<2i>.bias'."""
    if activations is None:
        activations = ['identity'] * (len(dims) - 1)
    if len(dims) - 1 != len(activations):
        raise ValueError('Number of activations must be the same as the number of dimensions - 1.')
    layers = []
    for d_in, d_out, act in zip(dims[:-1], dims[1:], activations):
        layers.append(nn.Linear(d_in, d_out, bias=bias))
        if isinstance(act, dict):
            kw = dict(act)
            layers.append(_ACT[kw.pop('name').lower()](**kw))
        else:
            layers.append(_ACT[act.lower()]())
        if dropout:
            layers.append(nn.Dropout(p=dropout))
    return nn.Sequential(*layers)


class _Conv1x1(nn.Module):  # cnn1x1, models_gcn.py:77-84
    def __init__(self, d1, d2, bias=True):
        super().__init__()
        self.cnn = nn.Conv2d(d1, d2, kernel_size=1, bias=bias)


class _NormData(nn.Module):  # norm_data, models_gcn.py:39-50
    def __init__(self, dim, node_n):
        super().__init__()
        self.bn = nn.BatchNorm1d(dim * node_n)


class _Embed(nn.Module):  # embed(norm=True), models_gcn.py:52-74
    def __init__(self, dim, dim1, node_n, bias):
        super().__init__()
        self.cnn = nn.Sequential(_NormData(dim, node_n), _Conv1x1(dim, 64, bias=bias), nn.ReLU(),
                                 _Conv1x1(64, dim1, bias=bias), nn.ReLU())


class _Similarity(nn.Module):  # compute_similarity, models_gcn.py:86-100
    def __init__(self, d1, d2, bias):
        super().__init__()
        self.s1 = _Conv1x1(d1, d2, bias=bias)
        self.s2 = _Conv1x1(d1, d2, bias=bias)


class GeoGcnParams(nn.Module):
    """Parameter container of Geo_gcn(node_n, 4, 128) (models_gcn.py:6-28)."""

    def __init__(self, node_n, in_channels, out_channels):
        super().__init__()
        self.joint_embed = _Embed(in_channels, 64, node_n, bias=True)
        self.get_s = _Similarity(64, 128, bias=True)
        self.weight = nn.Parameter(torch.empty(64, out_channels))
        stdv = 1. / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)


class TGGCN(nn.Module):
    def __init__(self, input_size: tuple, num_classes: tuple, hidden_size: int = 128,
                 discrete_networks_num_layers: int = 1, discrete_optimization_strategy: str = 'gumbel-sigmoid',
                 filter_discrete_updates: bool = False, gcn_node: int = 26,
                 message_humans_to_human: bool = True, message_human_to_objects: bool = True,
                 message_objects_to_human: bool = True, message_objects_to_object: bool = True,
                 message_geometry_to_objects: bool = True, message_geometry_to_human: bool = False,
                 message_segment: bool = False, message_type: str = 'relational', message_granularity: str = 'specific',
                 message_aggregation: str = 'attention', attention_style: str = 'concat',
                 object_segment_update_strategy: str = 'independent', update_segment_threshold: float = 0.5,
                 add_segment_length: bool = False, add_time_position: bool = False, time_position_strategy: str = 's',
                 positional_encoding_style: str = 'embedding', cat_level_states: bool = False,
                 share_level_mlps: bool = False, bias: bool = True):
        super().__init__()
        human_input_size, object_input_size = input_size
        num_subactivities, num_affordances = num_classes
        self.cfg = dict(hidden_size=hidden_size, discrete_networks_num_layers=discrete_networks_num_layers,
                        discrete_optimization_strategy=discrete_optimization_strategy,
                        filter_discrete_updates=filter_discrete_updates, gcn_node=gcn_node,
                        message_humans_to_human=message_humans_to_human,
                        message_human_to_objects=message_human_to_objects,
                        message_objects_to_human=message_objects_to_human,
                        message_objects_to_object=message_objects_to_object,
                        message_geometry_to_objects=message_geometry_to_objects,
                        message_geometry_to_human=message_geometry_to_human, message_segment=message_segment,
                        message_type=message_type, message_granularity=message_granularity,
                        message_aggregation=message_aggregation, attention_style=attention_style,
                        object_segment_update_strategy=object_segment_update_strategy,
                        update_segment_threshold=update_segment_threshold, add_segment_length=add_segment_length,
                        add_time_position=add_time_position, time_position_strategy=time_position_strategy,
                        positional_encoding_style=positional_encoding_style, cat_level_states=cat_level_states,
                        share_level_mlps=share_level_mlps, bias=bias)
        for k, v in self.cfg.items():
            if k not in ('hidden_size', 'discrete_networks_num_layers', 'bias', 'share_level_mlps'):
                setattr(self, k, v)
        self.num_classes = (num_subactivities, num_affordances)
        self.object_input_size = object_input_size
        h = hidden_size
        # ---- construction order follows vhoi/models.py:258-580 (same RNG stream => same initial weights)
        if add_time_position and positional_encoding_style in {'e', 'embedding'}:
            self.time_position_mlp = build_mlp([1, h], ['relu'], bias=bias)
        if add_segment_length and positional_encoding_style in {'e', 'embedding'}:
            self.segment_length_mlp = build_mlp([1, h], ['relu'], bias=bias)
        self.geometry_embedding_gcn = GeoGcnParams(gcn_node, 4, 128)
        self.geometry_embedding_mlp = build_mlp([gcn_node * 128, 2048, h], ['relu', 'relu'], bias=bias)
        self.geometry_bd_rnn = nn.GRU(h, h, num_layers=1, bias=bias, batch_first=True, bidirectional=True)
        self.geometry_bd_embedding_mlp = build_mlp([2 * h, h], ['relu'], bias=bias)
        self.human_embedding_mlp = build_mlp([2048, h], ['relu'], bias=bias)
        self.human_bd_rnn = nn.GRU(h, h, num_layers=1, bias=bias, batch_first=True, bidirectional=True)
        self.human_bd_embedding_mlp = build_mlp([2 * h, h], ['relu'], bias=bias)
        hs_in = h
        if message_humans_to_human:
            hs_in += h + (h if message_segment else 0)
        if message_geometry_to_human:
            hs_in += h
        if message_objects_to_human:
            hs_in += h + (h if message_segment else 0)
        if add_time_position and time_position_strategy == 's':
            hs_in += h
        if add_segment_length:
            hs_in += h
        self.human_segment_rnn_fcell = nn.GRUCell(hs_in, h, bias=bias)
        self.human_segment_rnn_bcell = nn.GRUCell(hs_in, h, bias=bias)
        self.object_embedding_mlp = build_mlp([object_input_size, h], ['relu'], bias=bias)
        self.object_bd_rnn = nn.GRU(h, h, num_layers=1, bias=bias, batch_first=True, bidirectional=True)
        self.object_bd_embedding_mlp = build_mlp([2 * h, h], ['relu'], bias=bias)
        os_in = h
        if message_geometry_to_objects:
            os_in += h
        if message_human_to_objects:
            os_in += h + (h if message_segment else 0)
        if message_objects_to_object:
            os_in += h + (h if message_segment else 0)
        if add_time_position and time_position_strategy == 's':
            os_in += h
        if add_segment_length:
            os_in += h
        self.object_segment_rnn_fcell = nn.GRUCell(os_in, h, bias=bias)
        self.object_segment_rnn_bcell = nn.GRUCell(os_in, h, bias=bias)
        relational = message_type in {'v1', 'relational'}
        generic = message_granularity in {'v1', 'generic'}
        attention = message_aggregation in {'att', 'attention'}
        general = attention_style in {'v4', 'general'}
        # (enabled, relational prefix, message-mlp stem, attention-mlp stem)  -- names as in vhoi/models.py:323-520
        relations = [
            (message_humans_to_human, 'human_human', 'humans_to_human', 'humans_to_human'),
            (message_human_to_objects, 'object_human', 'human_to_object', 'humans_to_object'),
            (message_objects_to_human, 'human_object', 'objects_to_human', 'objects_to_human'),
            (message_objects_to_object, 'object_object', 'objects_to_object', 'objects_to_object'),
            (message_geometry_to_human, 'human_geometry', 'geometry_to_human', 'geometry_to_human'),
            (message_geometry_to_objects, 'object_geometry', 'geometry_to_object', 'geometry_to_object'),
        ]
        for on, rel, msg, att in relations:
            if not on:
                continue
            if relational:
                setattr(self, f'{rel}_pairwise_relation_mlp', build_mlp([4 * h, h], ['relu'], bias=bias))
                setattr(self, f'{rel}_full_relation_mlp', build_mlp([h, h], ['relu'], bias=bias))
                if message_segment:
                    setattr(self, f'{rel}_segment_pairwise_relation_mlp', build_mlp([2 * h, h], ['relu'], bias=bias))
                    setattr(self, f'{rel}_segment_full_relation_mlp', build_mlp([h, h], ['relu'], bias=bias))
            else:
                k = 2 if generic else 4
                setattr(self, f'{msg}_message_mlp', build_mlp([k * h, h], ['relu'], bias=bias))
                if message_segment:
                    setattr(self, f'{msg}_segment_message_mlp', build_mlp([(k // 2) * h, h], ['relu'], bias=bias))
                if attention:
                    if general:
                        setattr(self, f'{att}_message_att_mlp', nn.Bilinear(2 * h, 2 * h, 1, bias=bias))
                        if message_segment:
                            setattr(self, f'{att}_segment_message_att_mlp', nn.Bilinear(h, h, 1, bias=bias))
                    else:
                        setattr(self, f'{att}_message_att_mlp', build_mlp([4 * h, 1], ['relu'], bias=bias))
                        if message_segment:
                            setattr(self, f'{att}_segment_message_att_mlp', build_mlp([2 * h, 1], ['relu'], bias=bias))
        uh_in = 2 * h + (h if message_humans_to_human else 0) + (h if message_objects_to_human else 0) + \
            (h if message_geometry_to_human else 0) + (h if (add_time_position and time_position_strategy == 'u') else 0)
        n_hidden = discrete_networks_num_layers - 1
        acts = ['relu'] * n_hidden + ['sigmoid']
        self.update_human_segment_mlp = build_mlp([uh_in] + [h] * n_hidden + [1], acts, bias=bias)
        if object_segment_update_strategy not in {'same_as_human', 'sah'}:
            uo_in = 2 * h + (h if message_human_to_objects else 0) + (h if message_objects_to_object else 0) + \
                (h if message_geometry_to_objects else 0) + \
                (h if (add_time_position and time_position_strategy == 'u') else 0)
            self.update_object_segment_mlp = build_mlp([uo_in] + [h] * n_hidden + [1], acts, bias=bias)
        lab_in = 2 * h + (2 * h if cat_level_states else 0)
        lsm = [{'name': 'logsoftmax', 'dim': -1}]
        self.human_recognition_mlp = build_mlp([lab_in, num_subactivities], lsm, bias=bias)
        self.human_prediction_mlp = build_mlp([lab_in, num_subactivities], lsm, bias=bias)
        if num_affordances is not None:
            self.object_recognition_mlp = build_mlp([lab_in, num_affordances], lsm, bias=bias)
            self.object_prediction_mlp = build_mlp([lab_in, num_affordances], lsm, bias=bias)
        if share_level_mlps and not cat_level_states:
            self.human_frame_recognition_mlp = self.human_recognition_mlp
            self.human_frame_prediction_mlp = self.human_prediction_mlp
            if num_affordances is not None:
                self.object_frame_recognition_mlp = self.object_recognition_mlp
                self.object_frame_prediction_mlp = self.object_prediction_mlp
        else:
            self.human_frame_recognition_mlp = build_mlp([2 * h, num_subactivities], lsm, bias=bias)
            self.human_frame_prediction_mlp = build_mlp([2 * h, num_subactivities], lsm, bias=bias)
            if num_affordances is not None:
                self.object_frame_recognition_mlp = build_mlp([2 * h, num_affordances], lsm, bias=bias)
                self.object_frame_prediction_mlp = build_mlp([2 * h, num_affordances], lsm, bias=bias)
        self._gumbel_noise_override = None  # tests: replay the exact noise the oracle / reference drew
        self._check_supported()

    # ------------------------------------------------------------------------------------------------------------
    def _check_supported(self):
        c = self.cfg
        bad = []
        if c['discrete_networks_num_layers'] < 1:
            bad.append('discrete_networks_num_layers < 1')
        if c['discrete_optimization_strategy'] not in {'gumbel-sigmoid', 'gs', 'straight-through', 'st'}:
            raise ValueError('strategy must be either straight-through or gumbel-sigmoid, not '
                             f"{c['discrete_optimization_strategy']}.")
        if (c['add_time_position'] or c['add_segment_length']) and c['hidden_size'] % 2 and \
                c['positional_encoding_style'] not in {'e', 'embedding'}:
            bad.append('periodic position embedding with an odd hidden_size')   # the reference asserts (models.py:1787)
        self._unsupported = bad

    def forward(self, x_human, x_objects, objects_mask, human_segmentation=None, objects_segmentation=None,
                human_human_distances=None, human_object_distances=None, object_object_distances=None,
                steps_per_example=None, inspect_model=False):
        """Same contract as the reference forward (vhoi/models.py:584-933): returns the list of 6 tensors
        [y_hs, y_hss, frame_rec, frame_pred, rec, pred] (12 with affordance heads); with ``inspect_model=True`` the pair
        (that list, [a_frame, a_segment_forward, a_segment_backward]) of :928-933."""
        if self._unsupported:
            raise NotImplementedError('configuration not implemented by the gfx950 path: ' + ', '.join(self._unsupported)
                                      + '; ' + ops.SUPPORTED_NOTE)
        dists = dict(hh=human_human_distances, ho=human_object_distances, oo=object_object_distances)
        dists = {k: v.to(device=x_human.device, dtype=torch.float32).contiguous() for k, v in dists.items()
                 if v is not None}
        bs, T, H, F_h = x_human.shape
        O = x_objects.shape[2]
        vw = F_h - 2048  # generalises the reference's hard-coded 76 / 120 / 104 split (vhoi/models.py:631-639)
        if vw <= 0 or vw % 4 or vw // 4 != self.gcn_node:
            raise ValueError(f'x_human feature size {F_h} does not match gcn_node={self.gcn_node} '
                             f'(expected 2048 + 4*gcn_node)')
        x_human = x_human.contiguous().float()
        x_objects = x_objects.contiguous().float()
        objects_mask = objects_mask.contiguous().float()
        n_sub, n_aff = self.num_classes
        plan = ops.Plan(self.cfg, bs, T, H, O, self.gcn_node, x_objects.shape[-1], n_sub, n_aff,
                        human_segmentation is not None, objects_segmentation is not None)
        plan.stage_hook = ops.get_model_extra(self, 'stage_hook')   # data-parallel overlap (ops.set_grad_stage_hook)
        plan.dists = dists or None
        plan.steps = None
        if steps_per_example is not None and (plan.time_s or plan.time_u or plan.seglen):
            plan.steps = steps_per_example.to(device=x_human.device, dtype=torch.float32).contiguous()
        noise = None
        n_gated = (H if plan.learn_h else 0) + (O if plan.own_o else 0)
        if plan.gs and n_gated:
            if self._gumbel_noise_override is not None:
                noise = self._gumbel_noise_override
            elif ops.get_model_extra(self, 'noise_shard') is not None:
                # data-parallel equivalence mode (distributed.DataParallel(global_noise=True)): every rank draws the noise
                # of the GLOBAL batch from an identically seeded generator and keeps its own clips
                rank, world, gen = ops.get_model_extra(self, 'noise_shard')
                u = torch.rand(T * n_gated, world * bs, 2, generator=gen).clamp_(1e-10, 1.0 - 1e-7)
                noise = (-torch.log(-torch.log(u)))[:, rank * bs:(rank + 1) * bs]
            else:
                # drawn on the CPU default generator like the reference (pyrutils/torch/distributions.py:16);
                # one (T*n_gated, bs, 2) draw equals the reference's T*n_gated sequential (bs, 2) draws
                noise = torch.distributions.gumbel.Gumbel(0.0, 1.0).sample((T * n_gated, bs, 2))
            noise = noise.to(device=x_human.device, dtype=torch.float32, non_blocking=True).contiguous()
            assert noise.numel() == T * n_gated * bs * 2, (tuple(noise.shape), T, n_gated, bs)
        names = ops.used_parameter_names(plan)
        sd = dict(self.named_parameters())
        params = [sd[n] for n in names]
        bn = self.geometry_embedding_gcn.joint_embed.cnn[0].bn
        bn_bufs = dict(running_mean=bn.running_mean, running_var=bn.running_var,
                       num_batches_tracked=bn.num_batches_tracked, stats_reduce=ops.get_model_extra(self, 'bn_stats_reduce'))
        hs = human_segmentation.float() if human_segmentation is not None else None
        osg = objects_segmentation.float() if objects_segmentation is not None else None
        out = ops.TGGCNFunction.apply(plan, names, self.training, bn_bufs, x_human, x_objects, objects_mask, hs, osg,
                                      noise, bool(inspect_model), *params)
        out = list(out)
        if inspect_model:   # (outputs, [frame-level, segment forward, segment backward] objects->human attention weights)
            return out[:-3], out[-3:]
        return out


def select_model(model_name: str):
    """vhoi/models.py:1589-1595. Only the 2G-GCN model is on the hot path; the two baselines are out of scope
     and raise KeyError like any unknown name."""
    return {'2G-GCN': TGGCN}[model_name]
