"""CPU: the opt-in length-bucketed batching. It must (1) visit every clip exactly once in batches of
neighbouring lengths, trimmed to the batch's own longest clip, dropping nothing but padding; (2) be a no-op on the
numbers when no clip of a batch is padded; and (3) change them exactly the way Appendix A9 predicts when padding is
removed: the model has no length masking, so the time-forward recurrence of the valid frames is untouched while the
time-backward recurrence (which starts on the padding) moves."""
import numpy as np
import torch

import twog_gcn_amd  # noqa: F401
from twog_gcn_amd import data_loading as dl
from oracle import cpu_ref
from tests.test_batching_cpu import raw_videos


def _loaders(kind='mphoi', bs=2):
    vids = raw_videos(kind, 60) + raw_videos(kind, 61)          # six clips, raw lengths 20 / 14 / 17 twice
    plain, _, _ = dl.create_data_loader(vids, '2G-GCN', 'multiple', kind, batch_size=bs, shuffle=False, downsampling=3)
    buck, _, _ = dl.create_data_loader(vids, '2G-GCN', 'multiple', kind, batch_size=bs, shuffle=False, downsampling=3,
                                       length_bucketing=True)
    return plain, buck


def test_default_is_the_reference_batching():
    plain, _ = _loaders()
    assert not isinstance(plain.batch_sampler, dl.LengthBucketedBatchSampler)
    t_max = plain.dataset.tensors[0].shape[1]
    assert all(b[0].shape[1] == t_max for b in plain)           # padded to the split maximum, every batch


def test_bucketed_batches_cover_every_clip_once_and_trim_only_padding():
    plain, buck = _loaders()
    full = plain.dataset.tensors
    steps = full[7]
    seen = []
    for batch_idx, batch in zip(buck.batch_sampler, buck):
        seen += batch_idx
        t_b = int(steps[batch_idx].max())
        lens = steps[batch_idx]
        assert float(lens.max() - lens.min()) <= float(steps.max() - steps.min())
        for slot, (t, f) in enumerate(zip(batch, full)):
            ref = f[batch_idx]
            if f.dim() >= 2 and f.shape[1] == full[0].shape[1]:
                assert t.shape[1] == t_b, (slot, t.shape, t_b)
                assert torch.equal(t, ref[:, :t_b])
                cut = ref[:, t_b:]       # what was trimmed is padding: zeros, ignore labels (-1), or -- slot 3, the
                pad = (cut == 0) | (cut == -1) | ((cut == 1) if slot == 3 else False)   # input-style flags -- ones
                assert cut.numel() == 0 or bool(pad.all()), slot
            else:
                assert torch.equal(t, ref)
    assert sorted(seen) == list(range(len(steps)))
    order = [float(steps[i]) for i in seen]
    assert order == sorted(order)                               # neighbouring lengths share a batch


def test_shuffle_permutes_batches_not_membership():
    steps = torch.tensor([5., 9., 5., 7., 9., 6.])
    g = torch.Generator().manual_seed(3)
    s = dl.LengthBucketedBatchSampler(steps, 2, shuffle=True, generator=g)
    a, b = list(s), list(s)
    assert sorted(map(tuple, a)) == sorted(map(tuple, b)) == sorted(map(tuple, s.batches))
    assert len(s) == 3 and {i for bt in a for i in bt} == set(range(6))


def test_trimming_changes_only_what_appendix_a9_predicts():
    """Frame-level BiGRU of one short clip, padded to the split maximum vs trimmed to its own length (oracle = the
    reference's arithmetic): forward-direction states of the valid frames are bit-identical, backward-direction states
    differ (they have run over the zero padding first)."""
    torch.manual_seed(0)
    h, T_pad, T_own = 8, 12, 7
    gru = torch.nn.GRU(h, h, num_layers=1, batch_first=True, bidirectional=True)
    sd = {'human_bd_rnn.' + k: v.detach() for k, v in gru.state_dict().items()}
    x = torch.zeros(2, T_pad, h)
    x[:, :T_own] = torch.randn(2, T_own, h)
    padded = cpu_ref._bigru(sd, 'human_bd_rnn', x)
    trimmed = cpu_ref._bigru(sd, 'human_bd_rnn', x[:, :T_own])
    assert torch.equal(padded[:, :T_own, :h], trimmed[..., :h])            # time-forward half: untouched
    assert (padded[:, :T_own, h:] - trimmed[..., h:]).abs().max() > 1e-4   # time-backward half: starts on padding
    # and with no padding in the batch the trimmed batch IS the plain batch
    full = cpu_ref._bigru(sd, 'human_bd_rnn', x[:, :T_own])
    assert torch.equal(full, trimmed)


def test_bucketed_loader_feeds_the_prefetcher_and_the_feeder():
    _, buck = _loaders()
    pf = dl.DevicePrefetcher(buck, dl.gcn_fetcher, 'cpu', dataset_name='mphoi')
    n = 0
    for (data, target), ref in zip(pf, buck):
        assert data[0].shape[1] == ref[0].shape[1] == int(ref[7].max())
        assert all(t.shape[1] == data[0].shape[1] for t in target if t.dim() >= 2)
        n += 1
    assert n == len(buck)
