"""TEST DOUBLE of the kernel interface (2g-gcn_amd/kernels.py::HipKernels) written in plain torch.

Test infrastructure only (lives under tests/, injected explicitly through kernels._set_backend_for_tests):
  * `-m "not gpu"` tests run the host logic (ops.py / models.py: buffer layouts, GEMM operand forms, the hand-derived
    backward pass) on CPU against the oracle;
  * `-m gpu` tests use each method here as the executable specification of the corresponding HIP kernel.
It follows the *kernel contracts* of include/twog_gcn.h, not the product code.
"""
import math

import torch


def _mat(t):
    return t.reshape(-1, t.shape[-1])


class FakeKernels:
    name = 'fake-torch'

    def __init__(self):
        self.calls = []
        self._tape = None

    def version(self):
        return 'fake'

    def empty(self, *shape, like=None, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=like.device)

    def zeros(self, *shape, like=None, dtype=torch.float32, device=None):
        return torch.zeros(*shape, dtype=dtype, device=like.device if like is not None else device)

    # ------------------------------------------------------------------ recorded steps of an affine loop
    # The double of HipKernels.tape_* (twog_tape_run): the six recordable calls are kept as (method, arguments) instead
    # of descriptor bytes; "affine in the step index" becomes: every tensor argument of step a + k is the same view of
    # the same storage as in step a, moved by k x (its offset in step b - its offset in step a) elements.
    def copy_blocks(self, pairs):
        with torch.no_grad():
            for s, d in pairs:
                d.view(-1).copy_(s.detach().reshape(-1))

    def tape_begin(self):
        assert self._tape is None
        self._tape = []

    def tape_end(self):
        t, self._tape = self._tape, None
        return t

    @staticmethod
    def _walk(x, y, fn):
        """fn(tensor_a, tensor_b) on every pair of tensors at the same place of two argument structures -> new structure."""
        if torch.is_tensor(x):
            assert torch.is_tensor(y) and x.shape == y.shape and x.stride() == y.stride(), 'not the same program'
            return fn(x, y)
        if isinstance(x, dict):
            assert isinstance(y, dict) and x.keys() == y.keys(), 'not the same program'
            return {k: FakeKernels._walk(x[k], y[k], fn) for k in x}
        if isinstance(x, (list, tuple)):
            assert type(x) is type(y) and len(x) == len(y), 'not the same program'
            return type(x)(FakeKernels._walk(a, b, fn) for a, b in zip(x, y))
        assert x == y or (x != x and y != y), ('not the same program', x, y)
        return x

    @staticmethod
    def _moved(k):
        def fn(ta, tb):
            if ta.data_ptr() == tb.data_ptr():
                return ta
            assert ta.untyped_storage().data_ptr() == tb.untyped_storage().data_ptr(), 'operands of consecutive steps in different buffers'
            off = ta.storage_offset() + k * (tb.storage_offset() - ta.storage_offset())
            return torch.as_strided(ta, ta.shape, ta.stride(), off)
        return fn

    def tape_matches(self, a, b, c, k):
        if not (len(a) == len(b) == len(c)) or any(x[0] != y[0] or x[0] != z[0] for x, y, z in zip(a, b, c)):
            return False
        ok = [True]

        def check(pred, tc):
            ok[0] = ok[0] and pred.data_ptr() == tc.data_ptr()
            return pred
        try:
            for (_, xa, ka), (_, xb, kb), (_, xc, kc) in zip(a, b, c):
                if ka != kb or ka != kc:
                    return False
                pred = self._walk(xa, xb, self._moved(k))
                self._walk(pred, xc, check)
        except AssertionError:
            return False
        return ok[0]

    def tape_run(self, a, b, k_begin, k_end, device=None):
        assert self._tape is None and len(a) == len(b)
        for k in range(k_begin, k_end):
            for (name, xa, kw), (nb, xb, _) in zip(a, b):
                assert name == nb
                getattr(self, name)(self._walk(xa, xb, self._moved(k)), **kw)

    # ------------------------------------------------------------------ GEMM
    def gemm(self, problems, a_kmajor=False, b_kmajor=False, split_k_workspace=True):
        if self._tape is not None:
            self._tape.append(('gemm', problems, dict(a_kmajor=a_kmajor, b_kmajor=b_kmajor)))
            return
        self.calls.append(('gemm', len(problems)))
        for p in problems:
            batch = p.get('batch') or (1, 0, 0, 0)
            nb, sa, sb, sc = batch
            for i in range(nb):
                A = self._shift(p['A'], i * sa)
                B = self._shift(p['B'], i * sb)
                Cm = self._shift(p['C'], i * sc)
                Am = _mat(A).t() if a_kmajor else _mat(A)
                Bm = _mat(B).t() if b_kmajor else _mat(B)  # (N, K)
                out = Am @ Bm.t()
                if p.get('bias') is not None:
                    out = out + p['bias']
                if p.get('accumulate'):
                    out = out + _mat(Cm)
                if p.get('act', 0) == 1:
                    out = torch.relu(out)
                Cm.copy_(out.reshape(Cm.shape))
            cs = p.get('colsum')
            if cs is not None:   # twog_gemm_t::a_colsum: column sums of the k-major A from the same pass
                assert a_kmajor and nb == 1
                v = _mat(p['A']).sum(0)
                cs.copy_(cs + v if p.get('colsum_accumulate') else v)

    def gemm_colsum_ok(self, problem):
        """The HIP library serves the fused column sums for aligned operands of the 128x128 class; the double says yes for
        every plain 2-D problem so that the CPU suite exercises the pairing logic of ops._Grads."""
        return problem['A'].dim() == 2 and problem['B'].dim() == 2

    @staticmethod
    def _shift(t, off):
        if off == 0:
            return t
        return torch.as_strided(t, t.shape, t.stride(), t.storage_offset() + off)

    # ------------------------------------------------------------------ GCN
    @staticmethod
    def _geo(x_human, N):
        bs, T = x_human.shape[:2]
        return x_human[:, :, 0, 2048:].reshape(bs * T, N, 4)

    def bn_fold(self, x_human, n_nodes, gamma, beta, running_mean, running_var, num_batches_tracked, training,
                stats_reduce=None, fold=None):
        if fold is not None:
            wq, wk, bq = fold
            ab, mi = self.bn_fold(x_human, n_nodes, gamma, beta, running_mean, running_var, num_batches_tracked, training,
                                  stats_reduce=stats_reduce)
            return ab, mi, torch.cat([wk.t() @ wq, (wk.t() @ bq).view(1, 64)], 0)
        N = n_nodes
        x = self._geo(x_human, N).double()  # (F, N, 4)
        xc = x.permute(2, 1, 0).reshape(4 * N, -1)  # channel c*N+n
        nf = xc.shape[1]
        if training:
            sums = torch.cat([xc.sum(1), (xc * xc).sum(1)])
            if stats_reduce is not None:
                sums, nf = stats_reduce(sums, nf)
            mean = sums[:4 * N] / nf
            var = sums[4 * N:] / nf - mean * mean
            running_mean.mul_(0.9).add_(0.1 * mean.float())
            running_var.mul_(0.9).add_(0.1 * (var * nf / max(nf - 1, 1)).float())
            if num_batches_tracked is not None:
                num_batches_tracked.add_(1)
            mean, var = mean.float(), var.float()
        else:
            mean, var = running_mean.clone(), running_var.clone()
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        a = gamma * invstd
        return torch.stack([a, beta - mean * a]), torch.stack([mean, invstd])

    def _xhat(self, x_human, N, ab):
        x = self._geo(x_human, N)  # (F, N, 4)
        a = ab[0].view(4, N).t()
        b = ab[1].view(4, N).t()
        return x * a + b  # (F, N, 4)

    def gcn_embed1_fwd(self, x_human, n_nodes, ab, w1, b1):
        xh = self._xhat(x_human, n_nodes, ab)
        return torch.relu(xh.reshape(-1, 4) @ w1.t() + b1)

    def gcn_fused_fwd(self, x_human, n_nodes, ab, w1, b1, w2, b2, md, save_x=True):
        """Executable spec of geo_fused.hip: embed -> X -> folded similarity -> softmax -> aggregation."""
        e1 = self.gcn_embed1_fwd(x_human, n_nodes, ab, w1, b1)
        X = torch.relu(e1 @ w2.t() + b2)
        nf = X.shape[0] // n_nodes
        adj, Z = self.gcn_attn2_fwd(X, md, nf, n_nodes)
        return (X if save_x else None), adj, Z

    def gcn_embed1_bwd(self, x_human, n_nodes, ab, mean_invstd, w1, de1):
        N = n_nodes
        x = self._geo(x_human, N)
        xh = self._xhat(x_human, N, ab).reshape(-1, 4)
        dw1 = de1.t() @ xh
        db1 = de1.sum(0)
        dxh = (de1 @ w1).view(-1, N, 4)  # (F, N, 4)
        da = (dxh * x).sum(0).t().reshape(-1)  # channel c*N+n
        db = dxh.sum(0).t().reshape(-1)
        mean, invstd = mean_invstd[0], mean_invstd[1]
        return dw1, db1, invstd * (da - mean * db), db

    def gcn_attn_fwd(self, qk, x, n_frames, n_nodes):
        q = qk[:, :128].view(n_frames, n_nodes, 128)
        k = qk[:, 128:].view(n_frames, n_nodes, 128)
        xs = x.view(n_frames, n_nodes, 64)
        s = torch.softmax(q @ k.transpose(1, 2), dim=-1)
        return s.contiguous(), (s @ xs).reshape(-1, 64)

    def gcn_attn_bwd(self, qk, x, s, dz, n_frames, n_nodes):
        q = qk[:, :128].view(n_frames, n_nodes, 128)
        k = qk[:, 128:].view(n_frames, n_nodes, 128)
        xs = x.view(n_frames, n_nodes, 64)
        dzs = dz.view(n_frames, n_nodes, 64)
        dS = dzs @ xs.transpose(1, 2)
        dx = s.transpose(1, 2) @ dzs
        dP = s * (dS - (dS * s).sum(-1, keepdim=True))
        dq = dP @ k
        dk = dP.transpose(1, 2) @ q
        return dx.reshape(-1, 64), torch.cat([dq, dk], -1).reshape(-1, 256)

    def gcn_attn2_fwd(self, x, md, n_frames, n_nodes):
        """Executable spec of geo_attn_mfma.hip: P = X M + d, adj = softmax(P X^T), Z = adj X (md = [Mt | d])."""
        xs = x.view(n_frames, n_nodes, 64)
        p = xs @ md[:64].t() + md[64]
        s = torch.softmax(p @ xs.transpose(1, 2), dim=-1)
        return s.contiguous(), (s @ xs).reshape(-1, 64)

    def gcn_attn2_bwd(self, x, md, s, dz, n_frames, n_nodes):
        xs = x.view(n_frames, n_nodes, 64)
        dzs = dz.view(n_frames, n_nodes, 64)
        Mt, d = md[:64], md[64]
        p = xs @ Mt.t() + d
        dA = dzs @ xs.transpose(1, 2)
        dS = s * (dA - (dA * s).sum(-1, keepdim=True))
        dP = dS @ xs
        dx = s.transpose(1, 2) @ dzs + dS.transpose(1, 2) @ p + dP @ Mt
        dMt = (dP.transpose(1, 2) @ xs).sum(0)
        dd = dP.sum((0, 1))
        return dx.reshape(-1, 64), torch.cat([dMt, dd.view(1, 64)], 0)

    # ------------------------------------------------------------------ GRU
    @staticmethod
    def _gates(gi, gh, hp, h):
        r = torch.sigmoid(gi[..., :h] + gh[..., :h])
        z = torch.sigmoid(gi[..., h:2 * h] + gh[..., h:2 * h])
        hn = gh[..., 2 * h:]
        n = torch.tanh(gi[..., 2 * h:] + r * hn)
        return r, z, n, hn, (1 - z) * n + z * hp

    @staticmethod
    def _gates_bwd(d, save, hp, h, u=None):
        r, z, n, hn = save[..., :h], save[..., h:2 * h], save[..., 2 * h:3 * h], save[..., 3 * h:]
        g = (1 - z) * n + z * hp
        du = (d * (g - hp)).sum(-1)
        dg = d if u is None else u.unsqueeze(-1) * d
        dprev = torch.zeros_like(d) if u is None else (1 - u.unsqueeze(-1)) * d
        dn = dg * (1 - z)
        dz = dg * (hp - n)
        dprev = dprev + dg * z
        dn_pre = dn * (1 - n * n)
        dr_pre = dn_pre * hn * r * (1 - r)
        dz_pre = dz * z * (1 - z)
        dgi = torch.cat([dr_pre, dz_pre, dn_pre], -1)
        dgh = torch.cat([dr_pre, dz_pre, dn_pre * r], -1)
        return dgi, dgh, dprev, du

    def gru_step_fwd(self, steps):
        if self._tape is not None:
            self._tape.append(('gru_step_fwd', steps, {}))
            return
        for d in steps:
            h = d['hidden']
            gi = d['gi'] if d.get('gi2') is None else d['gi'] + d['gi2'].reshape(d['gi'].shape)
            hp = d['h_prev'] if d.get('h_prev') is not None else torch.zeros(*d['h_out'].shape)
            r, z, n, hn, g = self._gates(gi, d['gh'].reshape(gi.shape), hp.reshape(d['h_out'].shape), h)
            u = d.get('u')
            new = g if u is None else u.unsqueeze(-1) * g + (1 - u.unsqueeze(-1)) * hp.reshape(g.shape)
            d['h_out'].copy_(new.reshape(d['h_out'].shape))
            if d.get('save') is not None:
                d['save'].copy_(torch.cat([r, z, n, hn], -1).reshape(d['save'].shape))

    def gru_step_bwd(self, steps):
        if self._tape is not None:
            self._tape.append(('gru_step_bwd', steps, {}))
            return
        for d in steps:
            h = d['hidden']
            dh = d['dh'] if d.get('dh2') is None else d['dh'] + d['dh2'].reshape(d['dh'].shape)
            hp = d['h_prev'] if d.get('h_prev') is not None else torch.zeros(*d['dh'].shape)
            save = d['save'].reshape(*dh.shape[:-1], 4 * h)
            dgi, dgh, dprev, du = self._gates_bwd(dh, save, hp.reshape(dh.shape), h, d.get('u'))
            d['dgi'].copy_(dgi.reshape(d['dgi'].shape))
            d['dgh'].copy_(dgh.reshape(d['dgh'].shape))
            if d.get('dh_prev_accumulate'):
                d['dh_prev'].add_(dprev.reshape(d['dh_prev'].shape))
            else:
                d['dh_prev'].copy_(dprev.reshape(d['dh_prev'].shape))
            if d.get('du') is not None:
                d['du'].add_(du.reshape(d['du'].shape))

    def bigru_fwd(self, types, bs, T, h):
        outs = []
        for y in types:
            gi = y['gi']
            E = gi.shape[2]
            out = torch.zeros(bs, T, E, 2 * h, device=gi.device)
            save = torch.zeros(2, bs, T, E, 4 * h, device=gi.device)
            for d, (w, b) in enumerate(((y['w_hh_f'], y.get('b_hh_f')), (y['w_hh_r'], y.get('b_hh_r')))):
                hp = torch.zeros(bs, E, h, device=gi.device)
                order = range(T) if d == 0 else range(T - 1, -1, -1)
                for t in order:
                    gh = hp @ w.t() + (b if b is not None else 0.0)
                    r, z, n, hn, g = self._gates(gi[:, t, :, d * 3 * h:(d + 1) * 3 * h], gh, hp, h)
                    out[:, t, :, d * h:(d + 1) * h] = g
                    save[d, :, t] = torch.cat([r, z, n, hn], -1)
                    hp = g
            outs.append((out, save))
        return outs

    def bigru_bwd_would_persist(self, Es, bs, h):
        return False

    def bigru_bwd(self, types, bs, T, h, allow_persistent=True):
        outs = []
        for y in types:
            d_out, save, out = y['d_out'], y['save'], y['out']
            E = d_out.shape[2]
            d_gi = torch.zeros(bs, T, E, 6 * h, device=d_out.device)
            d_gh = torch.zeros_like(d_gi)
            for d, w in enumerate((y['w_hh_f'], y['w_hh_r'])):
                carry = torch.zeros(bs, E, h, device=d_out.device)
                order = range(T - 1, -1, -1) if d == 0 else range(T)
                for t in order:
                    tp = t - 1 if d == 0 else t + 1
                    hp = out[:, tp, :, d * h:(d + 1) * h] if 0 <= tp < T else torch.zeros(bs, E, h, device=d_out.device)
                    dh = d_out[:, t, :, d * h:(d + 1) * h] + carry
                    dgi, dgh, dprev, _ = self._gates_bwd(dh, save[d, :, t], hp, h)
                    d_gi[:, t, :, d * 3 * h:(d + 1) * 3 * h] = dgi
                    d_gh[:, t, :, d * 3 * h:(d + 1) * 3 * h] = dgh
                    carry = dprev + dgh @ w
            outs.append((d_gi, d_gh))
        return outs

    # ------------------------------------------------------------------ entity attention
    @staticmethod
    def _rows(t, inst, E):
        return None if t is None else t.reshape(inst, E, t.shape[-1])

    def _attn_weights(self, d):
        n, H, O = d['n_inst'], d['H'], d['O']
        fh = self._rows(d['feat_h'], n, H)
        fo = self._rows(d['feat_o'], n, O)
        sc = d['scale']
        mask = d.get('obj_mask')
        dev = fh.device
        if mask is None:
            m = torch.ones(n, O, device=dev)
        else:
            m = mask.repeat_interleave(d['inst_per_clip'], dim=0)[:n]

        def sm(scores, ok):
            s = torch.where(ok, scores, torch.full_like(scores, float('-inf')))
            w = torch.softmax(s, dim=-1)
            return torch.where(torch.isnan(w), torch.zeros_like(w), w)

        eyeH = torch.eye(H, dtype=torch.bool, device=dev)
        eyeO = torch.eye(O, dtype=torch.bool, device=dev)
        w = {}
        w['hh'] = sm(fh @ fh.transpose(1, 2) * sc, (~eyeH).expand(n, H, H))
        w['oh'] = sm(fh @ fo.transpose(1, 2) * sc, (m != 0).unsqueeze(1).expand(n, H, O))
        w['ho'] = sm(fo @ fh.transpose(1, 2) * sc, torch.ones(n, O, H, dtype=torch.bool, device=dev))
        w['oo'] = sm(fo @ fo.transpose(1, 2) * sc, (~eyeO).unsqueeze(0) & (m != 0).unsqueeze(1))
        for r in ('hh', 'oh', 'ho', 'oo'):
            if d.get('msg_' + r) is None:
                w[r] = torch.zeros_like(w[r])
        return w, m, fh, fo

    def attn_fwd(self, descs):
        for d in descs:
            n, H, O = d['n_inst'], d['H'], d['O']
            w, m, fh, fo = self._attn_weights(d)
            if d.get('att') is not None:
                d['att'].copy_(torch.cat([w['hh'].reshape(n, -1), w['oh'].reshape(n, -1), w['ho'].reshape(n, -1),
                                          w['oo'].reshape(n, -1)], -1).reshape(d['att'].shape))
            rm = m if d['recv_mask_ho'] else torch.ones_like(m)
            if d.get('msg_hh') is not None:
                d['out_hh'].copy_((w['hh'] @ self._rows(d['msg_hh'], n, H)).reshape(d['out_hh'].shape))
            if d.get('msg_oh') is not None:
                d['out_oh'].copy_((w['oh'] @ self._rows(d['msg_oh'], n, O)).reshape(d['out_oh'].shape))
            if d.get('msg_sh') is not None:
                d['out_sh'].copy_(d['msg_sh'].reshape(n, 1, -1).expand(n, H, -1).reshape(d['out_sh'].shape))
            if d.get('msg_ho') is not None:
                d['out_ho'].copy_(((w['ho'] @ self._rows(d['msg_ho'], n, H)) * rm.unsqueeze(-1)).reshape(d['out_ho'].shape))
            if d.get('msg_so') is not None:
                d['out_so'].copy_((d['msg_so'].reshape(n, 1, -1) * rm.unsqueeze(-1)).reshape(d['out_so'].shape))
            if d.get('msg_oo') is not None:
                d['out_oo'].copy_((w['oo'] @ self._rows(d['msg_oo'], n, O)).reshape(d['out_oo'].shape))

    def attn_bwd(self, descs):
        for b in descs:
            d = b['f']
            n, H, O = d['n_inst'], d['H'], d['O']
            att = d['att'].reshape(n, -1)
            o = 0
            w = {}
            for r, (R, S_) in (('hh', (H, H)), ('oh', (H, O)), ('ho', (O, H)), ('oo', (O, O))):
                w[r] = att[:, o:o + R * S_].reshape(n, R, S_)
                o += R * S_
            _, m, fh, fo = self._attn_weights(d)
            rm = m if d['recv_mask_ho'] else torch.ones_like(m)
            sc = d['scale']
            dfh = torch.zeros_like(fh)
            dfo = torch.zeros_like(fo)
            dwx, ox = {}, 0
            for r, (R, S_) in (('hh', (H, H)), ('oh', (H, O)), ('ho', (O, H)), ('oo', (O, O))):
                if b.get('dw_extra') is not None:
                    dwx[r] = b['dw_extra'].reshape(n, -1)[:, ox:ox + R * S_].reshape(n, R, S_)
                ox += R * S_

            def one(rel, R, S_, recv_scale, fq, fk, dfq, dfk):
                msg = d.get('msg_' + rel)
                if msg is None:
                    return
                msgs = self._rows(msg, n, S_)
                g = self._rows(b['dout_' + rel], n, R)
                if recv_scale is not None:
                    g = g * recv_scale.unsqueeze(-1)
                dmsg = w[rel].transpose(1, 2) @ g
                if b.get('relu_mask_dmsg'):
                    dmsg = dmsg * (msgs > 0)
                b['dmsg_' + rel].copy_(dmsg.reshape(b['dmsg_' + rel].shape))
                dw = g @ msgs.transpose(1, 2)
                if rel in dwx:
                    dw = dw + dwx[rel]
                ds = w[rel] * (dw - (w[rel] * dw).sum(-1, keepdim=True)) * sc
                dfq += ds @ fk
                dfk += ds.transpose(1, 2) @ fq

            one('hh', H, H, None, fh, fh, dfh, dfh)
            one('oh', H, O, None, fh, fo, dfh, dfo)
            one('ho', O, H, rm, fo, fh, dfo, dfh)
            one('oo', O, O, None, fo, fo, dfo, dfo)
            if d.get('msg_sh') is not None:
                dm = self._rows(b['dout_sh'], n, H).sum(1)
                if b.get('relu_mask_dmsg'):
                    dm = dm * (d['msg_sh'].reshape(n, -1) > 0)
                b['dmsg_sh'].copy_(dm.reshape(b['dmsg_sh'].shape))
            if d.get('msg_so') is not None:
                dm = (self._rows(b['dout_so'], n, O) * rm.unsqueeze(-1)).sum(1)
                if b.get('relu_mask_dmsg'):
                    dm = dm * (d['msg_so'].reshape(n, -1) > 0)
                b['dmsg_so'].copy_(dm.reshape(b['dmsg_so'].shape))
            for key, val in (('dfeat_h', dfh), ('dfeat_o', dfo)):
                dst = b[key]
                if b.get('dfeat_accumulate'):
                    dst.add_(val.reshape(dst.shape))
                else:
                    dst.copy_(val.reshape(dst.shape))

    # ------------------------------------------------------------------ segment-level recurrence
    @staticmethod
    def _seg_dims(p):
        return (int(p['rel_hh']) + int(p['rel_ho']), int(p['rel_oh']) + int(p['rel_oo']),
                int(p['rel_hh']) + int(p['rel_oh']), int(p['rel_ho']) + int(p['rel_oo']))

    def _seg_attn_desc(self, p, bufs, d, t, prev_h, prev_o):
        bs, H, O, h = p['bs'], p['H'], p['O'], p['hidden']
        nsh, nso, nmh, nmo = self._seg_dims(p)
        msh, mso = bufs['msrc_h'][d, :, t], bufs['msrc_o'][d, :, t]  # (bs, E, ns*h)
        mgh, mgo = bufs['mg_h'][d, :, t], bufs['mg_o'][d, :, t]
        desc = dict(feat_h=prev_h, feat_o=prev_o, obj_mask=p['obj_mask'], att=bufs['att'][d, t], n_inst=bs,
                    inst_per_clip=1, H=H, O=O, D=h, hidden=h, scale=p['att_scale'], recv_mask_ho=0)
        if p['rel_hh']:
            desc['msg_hh'], desc['out_hh'] = msh[..., :h], mgh[..., :h]
        if p['rel_ho']:
            o = h if p['rel_hh'] else 0
            desc['msg_ho'], desc['out_ho'] = msh[..., o:o + h], mgo[..., :h]
        if p['rel_oh']:
            o = h if p['rel_hh'] else 0
            desc['msg_oh'], desc['out_oh'] = mso[..., :h], mgh[..., o:o + h]
        if p['rel_oo']:
            o1 = h if p['rel_oh'] else 0
            o2 = h if p['rel_ho'] else 0
            desc['msg_oo'], desc['out_oo'] = mso[..., o1:o1 + h], mgo[..., o2:o2 + h]
        return desc

    def segrnn_fwd(self, p):
        bs, T, H, O, h = p['bs'], p['T'], p['H'], p['O'], p['hidden']
        dev = p['gi_h'].device
        nsh, nso, nmh, nmo = self._seg_dims(p)
        natt = H * H + 2 * H * O + O * O
        z = lambda *s: torch.zeros(*s, device=dev)
        bufs = dict(hs_h=z(bs, T, H, 2 * h), hs_o=z(bs, T, O, 2 * h), save_h=z(2, bs, T, H, 4 * h),
                    save_o=z(2, bs, T, O, 4 * h), msrc_h=z(2, bs, T, H, nsh * h), msrc_o=z(2, bs, T, O, nso * h),
                    mg_h=z(2, bs, T, H, nmh * h), mg_o=z(2, bs, T, O, nmo * h), att=z(2, T, bs, natt))
        msg = p['msg_segment'] and (nmh + nmo) > 0
        for d in range(2):
            prev_h, prev_o = z(bs, H, h), z(bs, O, h)
            order = range(T) if d == 0 else range(T - 1, -1, -1)
            for t in order:
                if msg:
                    if nsh:
                        bufs['msrc_h'][d, :, t] = torch.relu(prev_h @ p['w_smsg_h'].t() + (p['b_smsg_h'] if p.get('b_smsg_h') is not None else 0.0))
                    if nso:
                        bufs['msrc_o'][d, :, t] = torch.relu(prev_o @ p['w_smsg_o'].t() + (p['b_smsg_o'] if p.get('b_smsg_o') is not None else 0.0))
                    self.attn_fwd([self._seg_attn_desc(p, bufs, d, t, prev_h, prev_o)])
                new = []
                for kind, E, prev in (('h', H, prev_h), ('o', O, prev_o)):
                    gi = p['gi_' + kind][:, t, :, d * 3 * h:(d + 1) * 3 * h]
                    nm = nmh if kind == 'h' else nmo
                    if msg and nm:
                        gi = gi + bufs['mg_' + kind][d, :, t] @ p['w_ihm_' + kind][d].t()
                    bhh = p['b_hh_' + kind][d]
                    gh = prev @ p['w_hh_' + kind][d].t() + (bhh if bhh is not None else 0.0)
                    r, zz, n, hn, g = self._gates(gi, gh, prev, h)
                    u = p['u_' + kind][:, t].unsqueeze(-1)
                    hnew = u * g + (1 - u) * prev
                    bufs['hs_' + kind][:, t, :, d * h:(d + 1) * h] = hnew
                    bufs['save_' + kind][d, :, t] = torch.cat([r, zz, n, hn], -1)
                    new.append(hnew)
                prev_h, prev_o = new
        return bufs

    def segrnn_bwd(self, p, bufs, d_hs_h, d_hs_o):
        bs, T, H, O, h = p['bs'], p['T'], p['H'], p['O'], p['hidden']
        dev = bufs['hs_h'].device
        nsh, nso, nmh, nmo = self._seg_dims(p)
        z = lambda *s: torch.zeros(*s, device=dev)
        out = dict(d_gi_h=z(bs, T, H, 6 * h), d_gi_o=z(bs, T, O, 6 * h), d_gh_h=z(bs, T, H, 6 * h),
                   d_gh_o=z(bs, T, O, 6 * h), d_u_h=z(bs, T, H), d_u_o=z(bs, T, O),
                   d_pre_h=z(2, bs, T, H, nsh * h), d_pre_o=z(2, bs, T, O, nso * h))
        msg = p['msg_segment'] and (nmh + nmo) > 0
        dhs = {'h': d_hs_h, 'o': d_hs_o}
        for d in range(2):
            carry = {'h': z(bs, H, h), 'o': z(bs, O, h)}
            order = range(T - 1, -1, -1) if d == 0 else range(T)
            for t in order:
                tp = t - 1 if d == 0 else t + 1
                has_prev = 0 <= tp < T
                prev = {k: (bufs['hs_' + k][:, tp, :, d * h:(d + 1) * h] if has_prev else z(bs, E, h))
                        for k, E in (('h', H), ('o', O))}
                newc = {}
                dmg = {}
                for kind, E in (('h', H), ('o', O)):
                    dh = dhs[kind][:, t, :, d * h:(d + 1) * h] + carry[kind]
                    u = p['u_' + kind][:, t]
                    dgi, dgh, dprev, du = self._gates_bwd(dh, bufs['save_' + kind][d, :, t], prev[kind], h, u)
                    out['d_gi_' + kind][:, t, :, d * 3 * h:(d + 1) * 3 * h] = dgi
                    out['d_gh_' + kind][:, t, :, d * 3 * h:(d + 1) * 3 * h] = dgh
                    out['d_u_' + kind][:, t] += du
                    newc[kind] = dprev + dgh @ p['w_hh_' + kind][d]
                    nm = nmh if kind == 'h' else nmo
                    if msg and nm:
                        dmg[kind] = dgi @ p['w_ihm_' + kind][d]
                if msg:
                    desc = self._seg_attn_desc(p, bufs, d, t, prev['h'], prev['o'])
                    b = dict(f=desc, relu_mask_dmsg=1, dfeat_accumulate=1, dfeat_h=newc['h'], dfeat_o=newc['o'])
                    dph, dpo = out['d_pre_h'][d, :, t], out['d_pre_o'][d, :, t]
                    if p['rel_hh']:
                        b['dout_hh'], b['dmsg_hh'] = dmg['h'][..., :h], dph[..., :h]
                    if p['rel_ho']:
                        o = h if p['rel_hh'] else 0
                        b['dout_ho'], b['dmsg_ho'] = dmg['o'][..., :h], dph[..., o:o + h]
                    if p['rel_oh']:
                        o = h if p['rel_hh'] else 0
                        b['dout_oh'], b['dmsg_oh'] = dmg['h'][..., o:o + h], dpo[..., :h]
                    if p['rel_oo']:
                        o1 = h if p['rel_oh'] else 0
                        o2 = h if p['rel_ho'] else 0
                        b['dout_oo'], b['dmsg_oo'] = dmg['o'][..., o2:o2 + h], dpo[..., o1:o1 + h]
                    self.attn_bwd([b])
                    if nsh:
                        newc['h'] = newc['h'] + dph @ p['w_smsg_h']
                    if nso:
                        newc['o'] = newc['o'] + dpo @ p['w_smsg_o']
                carry = newc
        return out

    # ------------------------------------------------------------------ gates
    def gate_fwd(self, d):
        bs, T, E, h = d['bs'], d['T'], d['E'], d['hidden']
        x = d['x']
        w = d['w'].reshape(-1)
        logit = 0
        for i, c in enumerate(d['seg_col']):
            logit = logit + x[:, c:c + h] @ w[i * h:(i + 1) * h]
        if d.get('b') is not None:
            logit = logit + d['b'][0]
        p = torch.sigmoid(logit).view(bs, T, E)
        if d.get('noise') is not None:
            g = d['noise'].reshape(T, d['noise_entities'], bs, 2)[:, d['noise_offset']:d['noise_offset'] + E]
            g = g.permute(2, 0, 1, 3)  # (bs, T, E, 2)
            a = torch.stack([torch.log(p + 1e-20) + g[..., 0], torch.log((1 - p) + 1e-20) + g[..., 1]], -1)
            y = torch.softmax(a, dim=-1)[..., 0]
        else:
            y = p
        hard = (y > d['threshold']).float()
        if d['force_last']:
            hard[:, T - 1] = 1.0
        d['hard'], d['soft'], d['p_save'] = hard, y.contiguous(), p.contiguous()
        return hard, d['soft']

    def gate_bwd(self, d, d_hard, d_soft, st_mask):
        bs, T, E = d['bs'], d['T'], d['E']
        tot = torch.zeros(bs, T, E, device=d['x'].device)
        if d_soft is not None:
            tot = tot + d_soft.view(bs, T, E)
        if d_hard is not None:
            m = st_mask.view(bs, T, E).clone() if st_mask is not None else torch.ones(bs, T, E, device=tot.device)
            if d['force_last']:
                m[:, T - 1] = 0.0
            tot = tot + d_hard.view(bs, T, E) * m
        p = d['p_save']
        if d.get('noise') is not None:
            y = d['soft']
            tot = tot * y * (1 - y) * (1 / (p + 1e-20) + 1 / ((1 - p) + 1e-20))
        return (tot * p * (1 - p)).reshape(-1)

    def rank1_update(self, dst, s, v):
        dst.add_((s.reshape(-1, 1) * v.reshape(1, -1)).reshape(dst.shape))

    def colsum(self, x, rowscale=None, out=None, accumulate=False):
        m = _mat(x)
        r = (m * rowscale.reshape(-1, 1)).sum(0) if rowscale is not None else m.sum(0)
        if out is None:
            return r
        if accumulate:
            out.add_(r)
        else:
            out.copy_(r)
        return out

    def colsum_many(self, ops):
        for x, rs, out, acc in ops:
            self.colsum(x, rowscale=rs, out=out, accumulate=acc)

    def filter_fwd(self, soft, threshold):
        um1 = torch.cat([torch.zeros_like(soft[:, :1]), soft[:, :-1]], 1)
        up1 = torch.cat([soft[:, 1:], torch.zeros_like(soft[:, :1])], 1)
        cond = (soft > um1) & (soft > up1) & (soft >= threshold)
        return cond.float(), (cond | ~(soft >= threshold)).float()

    # ------------------------------------------------------------------ reorder / heads / elementwise
    @staticmethod
    def _reorder_idx(gate):
        bs, T, E = gate.shape
        idx = torch.arange(T, device=gate.device).view(1, T, 1).repeat(bs, 1, E)
        nxt = torch.full((bs, E), -1, dtype=torch.long, device=gate.device)
        for t in range(T - 1, -1, -1):
            nz = gate[:, t] != 0
            nxt = torch.where(nz, torch.full_like(nxt, t), nxt)
            idx[:, t] = torch.where(nxt >= 0, nxt, torch.full_like(nxt, t))
        return idx

    def reorder_fwd(self, hx, gate):
        idx = self._reorder_idx(gate)
        return torch.gather(hx, 1, idx.unsqueeze(-1).expand_as(hx))

    def reorder_bwd(self, dout, gate):
        idx = self._reorder_idx(gate)
        return torch.zeros_like(dout).scatter_add_(1, idx.unsqueeze(-1).expand_as(dout), dout)

    def logsoftmax_permute_fwd(self, logits, bs, T, E, Cn):
        return torch.log_softmax(logits.view(bs, T, E, Cn), -1).permute(0, 3, 1, 2).contiguous()

    def logsoftmax_permute_bwd(self, out, dout):
        o = out.permute(0, 2, 3, 1)
        g = dout.permute(0, 2, 3, 1)
        return (g - torch.exp(o) * g.sum(-1, keepdim=True)).reshape(-1, out.shape[1]).contiguous()

    def relu_bwd(self, dy, y, dx=None):
        r = _mat(dy) * (_mat(y) > 0)
        if dx is None:
            return r.contiguous()
        dx.copy_(r.reshape(dx.shape))
        return dx

    def add_rows(self, src, dst):
        dst.add_(src.reshape(dst.shape))

    # ------------------------------------------------------------------ sender-side projection glue
    def ssp_fwd(self, gi, ph, ps, att, mask, n_inst, inst_per_clip, H, O, att_off):
        cols = gi.shape[-1]
        m = (mask.repeat_interleave(inst_per_clip, 0) if mask is not None else torch.ones(n_inst, O)).view(n_inst, O, 1)
        add = torch.zeros(n_inst, O, cols)
        if ph is not None:
            w = att[:, att_off:att_off + O * H].view(n_inst, O, H)
            add = add + torch.einsum('nkh,nhc->nkc', w, ph.view(n_inst, H, cols))
        if ps is not None:
            add = add + ps.view(n_inst, 1, cols)
        gi.add_((m * add).view(gi.shape))

    def ssp_gather(self, dgi, att, att_ld_clip, att_ld_frame, att_off, n_inst, inst_per_clip, H, O):
        cols = dgi.shape[1]
        flat = att.reshape(-1)
        qh = torch.zeros(n_inst, H, cols)
        g = dgi.reshape(n_inst, O, cols)
        for inst in range(n_inst):
            c, f = divmod(inst, inst_per_clip)
            base = c * att_ld_clip + f * att_ld_frame + att_off
            w = flat[base:base + O * H].view(O, H)
            qh[inst] = w.t() @ g[inst]
        return qh.reshape(n_inst * H, cols)

    def ssp_bwd(self, dgi, ph, att, mask, n_inst, inst_per_clip, H, O, att_off, want_qs, dw=None):
        cols = dgi.shape[-1]
        m = (mask.repeat_interleave(inst_per_clip, 0) if mask is not None else torch.ones(n_inst, O)).view(n_inst, O, 1)
        g = dgi.view(n_inst, O, cols) * (m != 0).float()
        qh = qs = None
        if ph is not None:
            w = att[:, att_off:att_off + O * H].view(n_inst, O, H)
            qh = torch.einsum('nkh,nkc->nhc', w, g).reshape(n_inst * H, cols)
            if dw is not None:
                dw[:, att_off:att_off + O * H] = torch.einsum('nkc,nhc->nkh', g, ph.view(n_inst, H, cols)).reshape(n_inst, O * H)
        if want_qs:
            qs = g.sum(1)
        return qh, qs

    # ------------------------------------------------------------------ general single-relation message passing
    REL_SUM, REL_DOT, REL_ADDITIVE, REL_DISTANCE, REL_MEAN = 0, 1, 2, 3, 4
    REL_MSG_SENDER, REL_MSG_PAIR = 0, 1

    @staticmethod
    def _rel_rows(t, n_inst, n):
        """(n_inst*n, w) 2-D view or (n_inst, n, w) 3-D view -> (n_inst, n, w)"""
        return None if t is None else (t if t.dim() == 3 else t.reshape(n_inst, n, t.shape[-1]))

    def _rel_forward(self, d, leaf=None):
        """Specification of twog_relation_fwd as differentiable torch code (leaf: dict of tensors to differentiate)."""
        nI, ipc, R, S = d['n_inst'], d['inst_per_clip'], d['R'], d['S']
        g = lambda k: (leaf or {}).get(k, d.get(k))
        valid = torch.ones(nI, R, S)
        mval = torch.ones(nI, 1, S)
        if d.get('send_mask') is not None:
            mval = d['send_mask'].repeat_interleave(ipc, 0).view(nI, 1, S)
            valid = valid * (mval != 0).float()
        if d.get('exclude_self'):
            valid = valid * (1 - torch.eye(R, S)).view(1, R, S)
        mode = d['score_mode']
        if mode == self.REL_DOT:
            q, k = self._rel_rows(g('q'), nI, R), self._rel_rows(g('k'), nI, S)
            sb = g('score_bias')
            sc = torch.einsum('nrd,nsd->nrs', q, k) * d.get('scale', 1.0) + (sb if sb is not None else 0.0)
            if d.get('relu_scores'):
                sc = torch.relu(sc)
        elif mode == self.REL_ADDITIVE:
            sc = torch.relu(g('a_r').view(nI, R, 1) + g('c_s').view(nI, 1, S))
        elif mode == self.REL_DISTANCE:
            dist = d['dist']
            valid = valid * (dist != 0).float()
            sc = 1.0 / (dist + 1e-7)
        else:
            sc = None
        if mode == self.REL_SUM:
            w = valid * mval
        elif mode == self.REL_MEAN:
            w = valid / valid.sum(-1, keepdim=True).clamp(min=1.0)
        else:
            sc = torch.where(valid.bool(), sc, torch.full_like(sc, float('-inf')))
            w = torch.softmax(sc, dim=-1)
            w = torch.where(torch.isnan(w), torch.zeros_like(w), w)
        if d['msg_mode'] == self.REL_MSG_SENDER:
            out = torch.einsum('nrs,nsh->nrh', w, self._rel_rows(g('msg'), nI, S))
        else:
            pr, ps = self._rel_rows(g('p_r'), nI, R), self._rel_rows(g('p_s'), nI, S)
            out = (w.unsqueeze(-1) * torch.relu(pr.unsqueeze(2) + ps.unsqueeze(1))).sum(2)
        if d.get('recv_mask') is not None:
            out = out * d['recv_mask'].repeat_interleave(ipc, 0).view(nI, R, 1)
        return out, w

    def relation_fwd(self, d):
        out, w = self._rel_forward(d)
        d['out'].copy_(out.reshape(d['out'].shape))
        if d.get('att') is not None:
            d['att'].copy_(w.reshape(d['att'].shape))

    def relation_fwd_many(self, ds):
        if self._tape is not None:
            self._tape.append(('relation_fwd_many', ds, {}))
            return
        for d in ds:
            self.relation_fwd(d)

    def relation_bwd_many(self, bs):
        if self._tape is not None:
            self._tape.append(('relation_bwd_many', bs, {}))
            return
        for b in bs:
            self.relation_bwd(b)

    def rowops(self, ops):
        if self._tape is not None:
            self._tape.append(('rowops', ops, {}))
            return
        """ops: ('relu_bwd', dy, y, dx) | ('add', src, dst) | ('rank1', dst, s, v) -- one launch on the device."""
        for op in ops:
            getattr(self, {'relu_bwd': 'relu_bwd', 'add': 'add_rows', 'rank1': 'rank1_update'}[op[0]])(*op[1:])

    def relation_bwd(self, b):
        d = b['f']
        keys = [k for k in ('q', 'k', 'msg', 'p_r', 'p_s', 'a_r', 'c_s', 'score_bias') if d.get(k) is not None]
        leaf = {k: d[k].detach().clone().requires_grad_(True) for k in keys}
        if d.get('q') is not None and d.get('k') is not None and d['q'].data_ptr() == d['k'].data_ptr() and \
                d['q'].shape == d['k'].shape:
            pass   # self relation: q and k stay separate leaves; their gradients go to dq / dk separately
        nI, R = d['n_inst'], d['R']
        with torch.enable_grad():   # called from inside an autograd backward pass, where grad mode is off
            out, _ = self._rel_forward(d, leaf)
            out.backward(self._rel_rows(b['dout'], nI, R).reshape(out.shape).detach())
        grad = lambda k: (leaf[k].grad if leaf[k].grad is not None else torch.zeros_like(leaf[k])) if k in leaf else None
        if b.get('dmsg') is not None:
            gm = grad('msg')
            if b.get('relu_mask_dmsg'):
                gm = gm * (d['msg'] > 0).float()
            b['dmsg'].copy_(gm.reshape(b['dmsg'].shape))
        for src, dst in (('p_r', 'dp_r'), ('p_s', 'dp_s'), ('a_r', 'da_r'), ('c_s', 'dc_s')):
            if b.get(dst) is not None and src in leaf:
                b[dst].copy_(grad(src).reshape(b[dst].shape))
        if b.get('dscore_sum') is not None:
            # per-instance sums are an implementation detail of the kernel; the double puts the total into element 0
            b['dscore_sum'].zero_()
            if 'score_bias' in leaf and leaf['score_bias'].grad is not None:
                b['dscore_sum'][0] = leaf['score_bias'].grad.reshape(-1)[0]
        for src, dst, acc in (('q', 'dq', 'dq_accumulate'), ('k', 'dk', 'dk_accumulate')):
            if b.get(dst) is not None:
                gq = grad(src) if src in leaf else torch.zeros(b[dst].shape)
                if b.get(acc):
                    b[dst].add_(gq.reshape(b[dst].shape))
                else:
                    b[dst].copy_(gq.reshape(b[dst].shape))

    # ------------------------------------------------------------------ position features / rare gate strategies
    def pos_embed_fwd(self, out, bs, T, E, hidden, w=None, b=None, periodic=False, s=None, steps=None, divide=False):
        if s is None:
            t = torch.arange(1, T + 1, dtype=torch.float32).view(1, T, 1).expand(bs, T, E)
            if divide:
                t = t / steps.view(bs, 1, 1)
            s = t.reshape(-1)
        s = s.reshape(-1).clone()
        x = s.view(-1, 1)
        if periodic:
            wk = torch.tensor([1e4]) ** torch.linspace(0, 1, hidden // 2)
            v = torch.cat([torch.sin(x / wk), torch.cos(x / wk)], dim=-1)
        else:
            v = torch.relu(x * w.view(1, -1) + (b.view(1, -1) if b is not None else 0.0))
        out.copy_(v.reshape(out.shape))
        return s

    def periodic_embed_bwd(self, dout, s):
        hidden = dout.shape[-1]
        half = hidden // 2
        wk = torch.tensor([1e4]) ** torch.linspace(0, 1, half)
        x = s.view(-1, 1)
        d = _mat(dout)
        return ((d[:, :half] * torch.cos(x / wk) - d[:, half:] * torch.sin(x / wk)) / wk).sum(-1)

    def seglen_fwd(self, u, steps, divide):
        bs, T, E = u.shape
        out = torch.zeros_like(u)
        acc = torch.zeros(bs, E)
        for t in range(T):
            xt = torch.full((bs, 1), float(t + 1))
            if divide:
                xt = xt / steps.view(bs, 1)
            rel = u[:, t] * xt
            rel = torch.where(rel != 0, rel - acc, rel)
            acc = acc + rel
            out[:, t] = rel
        return out

    def seglen_bwd(self, u, steps, divide, ds, du):
        bs, T, E = u.shape
        da = torch.zeros(bs, E)
        for t in range(T - 1, -1, -1):
            xt = torch.full((bs, 1), float(t + 1))
            if divide:
                xt = xt / steps.view(bs, 1)
            dr = ds[:, t] + da
            du[:, t] += dr * xt
            da = torch.where(u[:, t] * xt != 0, da - dr, da)

    def mul(self, a, b, out=None, accumulate=False):
        if out is None:
            return a * b
        if accumulate:
            out += a * b
        else:
            out.copy_(a * b)
        return out

    def scale_rows(self, x, s):
        x.mul_(s.view(*([-1] + [1] * (x.dim() - 1))) if x.dim() == 2 else s.view(x.shape[0], x.shape[1], 1))

    def adam_step(self, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
        grad = grad * grad_scale
        g = grad + weight_decay * param if weight_decay else grad
        exp_avg.mul_(beta1).add_(g, alpha=1 - beta1)
        exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
        param.addcdiv_(exp_avg, exp_avg_sq.sqrt() / math.sqrt(bc2) + eps, value=-lr / bc1)

    # ---------------------------------------------------------------- multi-task loss (executable spec of loss.hip)
    @staticmethod
    def _loss_stats(t):
        x, y, ig = t['input'], t['target'], t['ignore']
        if t['kind'] == 0:
            valid = (y != int(ig)) & (y >= 0) & (y < x.shape[1])
            picked = torch.gather(x, 1, y.clamp(0, x.shape[1] - 1).unsqueeze(1)).squeeze(1)
            return -(picked.double() * valid).sum(), valid.double().sum()
        m = y != ig
        if t['kind'] == 1:
            el = -(y * torch.log(x).clamp(min=-100.0) + (1 - y) * torch.log(1 - x).clamp(min=-100.0))
        else:
            el = x
        return torch.where(m, el, torch.zeros_like(el)).double().sum(), m.double().sum()

    def multitask_loss_fwd(self, terms):
        stats = torch.stack([torch.stack(self._loss_stats(t)) for t in terms])
        vals = []
        for t, (s, c) in zip(terms, stats):
            v = s / c if t['kind'] == 0 else (s / c if c > 0 else torch.zeros((), dtype=torch.float64))
            vals.append(t['weight'] * v.float())
        return torch.stack(vals).float(), stats

    def multitask_loss_bwd(self, terms, stats, dlosses, need):
        out = []
        for i, (t, nd) in enumerate(zip(terms, need)):
            if not nd:
                out.append(None)
                continue
            x, y, ig, c = t['input'], t['target'], t['ignore'], stats[i, 1]
            g = (t['weight'] * dlosses[i] / c.float()) if c > 0 else torch.zeros(())
            if t['kind'] == 0:
                valid = (y != int(ig)) & (y >= 0) & (y < x.shape[1])
                d = torch.zeros_like(x)
                d.scatter_(1, y.clamp(0, x.shape[1] - 1).unsqueeze(1), (-g * valid.float()).unsqueeze(1))
            elif t['kind'] == 1:
                d = torch.where(y != ig, g * (x - y) / ((1 - x) * x).clamp(min=1e-12), torch.zeros_like(x))
            else:
                d = torch.where(y != ig, g.expand_as(x), torch.zeros_like(x)).clone()
            out.append(d)
        return out

    # ---------------------------------------------------------------- inference post-processing (spec of postprocess.hip)
    def predict_labels(self, logp, downsampling, target_steps):
        bs, Cn, T, E = logp.shape
        t = torch.clamp(torch.arange(target_steps) // max(1, int(downsampling)), max=T - 1)
        return torch.from_numpy(logp[:, :, t].numpy().argmax(1).astype('int64'))

    def f1_at_k(self, y_true, y_pred, num_classes, overlap, ignore_value=None):
        from oracle import postprocess_ref as R  # the test double may lean on the oracle; the product never does
        f1, valid = [], []
        for yt, yp in zip(y_true.numpy(), y_pred.numpy()):
            if ignore_value is not None:
                keep = yt != ignore_value
                yt, yp = yt[keep], yp[keep]
            valid.append(float(yt.size > 0))
            f1.append(R.f1_at_k_single_example(yt, yp, num_classes, overlap) if yt.size else 0.0)
        return torch.tensor(f1, dtype=torch.float32), torch.tensor(valid, dtype=torch.float32)

