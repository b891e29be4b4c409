#!/usr/bin/env python3
"""Headline benchmark of the 2G-GCN hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]
  (N > 1: one rank per GPU over RCCL -- either launched by torch.distributed.run (WORLD_SIZE set), or, when started
  bare, bench.py starts the N ranks itself as child processes before touching the GPU and exits with their code)

Workload (BASELINE.json configs[2]): synthetic clips T=120, N=34 geometry nodes (H=2 humans, O=8 objects), C=h=512,
64 clips per GPU per step (weak scaling: global batch = 64*N; --scaling strong: global batch 64, 64/N per GPU), fp32. One "step" = forward + the reference's stage-1
loss list (NLL on the two segment-level heads, vhoi/losses.py:53-61) + backward + gradient all-reduce (N > 1) + fused
Adam; inputs are resident in HBM before the timed region. Prints ONE JSON line on rank 0.

Also measured live: the dominant kernel's roofline (fp32-MFMA GEMM, HIP events around every GEMM launch made from the
host composition during the timed steps) and, on rank 0 at N=1, the CPU baseline (the oracle = port of the reference's
PyTorch-CPU path) on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import sys
import time


def _cpu_quota():
    """Cores the cgroup really grants (the GPU boxes show 256 cores under a 16-core quota); see 2g-gcn_amd/hostcpu.py."""
    n = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


# before torch creates its thread pools: oversubscribing the quota gets the whole cgroup throttled in 100 ms periods,
# which stalls the thread that feeds the GPU (measured: sporadic +90 ms steps)
_OMP_FROM_USER = 'OMP_NUM_THREADS' in os.environ
os.environ.setdefault('OMP_NUM_THREADS', str(max(1, _cpu_quota() // int(os.environ.get('LOCAL_WORLD_SIZE', '1')) - 4)))

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = dict(hidden_size=512, gcn_node=34, attention_style='v3', bias=True, discrete_networks_num_layers=1,
           discrete_optimization_strategy='gs', filter_discrete_updates=False, message_humans_to_human=True,
           message_human_to_objects=True, message_objects_to_human=True, message_objects_to_object=True,
           message_geometry_to_objects=True, message_geometry_to_human=False, message_segment=True, message_type='v2',
           message_granularity='v1', message_aggregation='att', object_segment_update_strategy='ind',
           update_segment_threshold=0.5)  # conf/models/2G-GCN_stage1.yaml:5-29 with gcn_node=34
T, H, O, N_NODES, BS, N_CLASSES = 120, 2, 8, 34, 64, 13
# the other single-GPU shapes of BASELINE.json.configs, selectable with --workload (informational: NOT the headline metric)
WORKLOADS = {
    'c3': dict(H=2, O=8, N=34, h=512, bs=64, classes=13, name='Synthetic T=120 N=34 C=512 (BASELINE configs[2])'),
    'c2': dict(H=2, O=4, N=26, h=512, bs=8, classes=13, name='MPHOI-72 layout hs512 bs8 (BASELINE configs[1])'),
    'c5': dict(H=2, O=9, N=30, h=64, bs=16, classes=14, name='Bimanual layout h=64, 16 clips per GPU (BASELINE configs[4])'),
    # "also run h=512" (BASELINE.md section 4 prices it at 49.8 GFLOP per clip forward)
    'c5_hs512': dict(H=2, O=9, N=30, h=512, bs=16, classes=14,
                     name='Bimanual layout at h=512, 16 clips per GPU (SURVEY section 8 C5 "also run h=512")'),
    # the constructor's OWN defaults (vhoi/models.py:179-190: relational + receiver-specific messages, concat attention, no
    # segment-level messages, h = 128) and the same with segment-level messages: the general single-relation kernels
    # (relation.hip) and the host-composed segment loop -- no shipped configuration uses them; informational
    'defaults': dict(H=2, O=4, N=26, h=128, bs=8, classes=13, cfg={},
                     name='TGGCN constructor defaults (relational / specific / concat, h=128), MPHOI layout bs8'),
    'defaults_seg': dict(H=2, O=4, N=26, h=128, bs=8, classes=13, cfg=dict(message_segment=True),
                         name='TGGCN constructor defaults + message_segment (general segment loop), MPHOI layout bs8'),
}


def select_workload(name):
    global H, O, N_NODES, BS, N_CLASSES
    w = WORKLOADS[name]
    H, O, N_NODES, BS, N_CLASSES = w['H'], w['O'], w['N'], w['bs'], w['classes']
    if 'cfg' in w:   # constructor defaults + the listed overrides instead of the 2G-GCN_stage1 parameters
        CFG.clear()
        CFG.update(w['cfg'])
    CFG['hidden_size'], CFG['gcn_node'] = w['h'], w['N']
    return w
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 FLOP/clk/CU
PEAK_BF16_MFMA_TFLOPS = 2516.6  # dense bf16 (MI355X_MICROARCH.md: ~2.5 PF): v_mfma_f32_32x32x16_bf16, 32 768 FLOP / 32 clk / SIMD at 2.4 GHz
X3_PRODUCTS = 6                 # bf16 chunk products the X3 kernels execute per fp32 multiply-add (csrc/gemm_f32.hip)
PEAK_HBM_GBS = 8000.0


def synthetic_batch(bs, device, seed):
    g = torch.Generator().manual_seed(seed)
    vis = torch.relu(torch.randn(bs, T, H, 2048, generator=g))
    pos = torch.rand(bs, T, N_NODES, 2, generator=g)
    vel = torch.randn(bs, T, N_NODES, 2, generator=g) * 0.5
    geo = torch.cat([pos, vel], -1).reshape(bs, T, 1, 4 * N_NODES).expand(bs, T, H, 4 * N_NODES)
    x_human = torch.cat([vis, geo], -1).contiguous()
    x_objects = torch.relu(torch.randn(bs, T, O, 2048, generator=g))
    mask = torch.ones(bs, O)
    targets = [torch.randint(0, N_CLASSES, (bs, T, H), generator=g) for _ in range(2)]
    to = lambda t: t.to(device)
    return to(x_human), to(x_objects), to(mask), [to(t) for t in targets]


class GemmProfiler:
    """HIP-event timing of every GEMM launch issued by the host composition (on torch's current stream, which is the
    stream the kernels are enqueued on). Launches are classed by the tile variant the library picks."""

    def __init__(self, K, pool=None):
        """pool: pre-created timing events (created OUTSIDE the timed region: on a fresh box the first few hundred
        hipEventCreate calls of a process cost milliseconds each -- measured: 245 ms steps instead of 86)."""
        self.K, self.orig, self.records, self.shapes = K, K.gemm, [], []
        self.orig_attn = (K.attn_fwd, K.attn_bwd)
        self.attn = {'fwd': [], 'bwd': []}   # HIP events around the frame-level attention launches
        self.orig_chain = {n: getattr(K, n) for n in ('bigru_fwd', 'bigru_bwd', 'segrnn_fwd', 'segrnn_bwd')}
        self.chain = {n: [] for n in self.orig_chain}   # HIP events around the four time loops (they run inside the library)
        self.pool, self.used = pool if pool is not None else [], 0

    def event(self):
        if self.used < len(self.pool):
            e = self.pool[self.used]
        else:
            e = torch.cuda.Event(enable_timing=True)
            self.pool.append(e)
        self.used += 1
        return e

    def __enter__(self):
        def timed(kind, fn):
            def call(descs):
                d = descs[0] if kind == 'fwd' else descs[0]['f']
                if d['inst_per_clip'] <= 1:          # (segment-level calls happen inside the library's time loop)
                    return fn(descs)
                e0, e1 = self.event(), self.event()
                e0.record()
                fn(descs)
                e1.record()
                self.attn[kind].append((e0, e1))
            return call
        self.K.attn_fwd, self.K.attn_bwd = timed('fwd', self.orig_attn[0]), timed('bwd', self.orig_attn[1])

        def chain_timed(name, fn):
            def call(*a, **kw):
                e0, e1 = self.event(), self.event()
                e0.record()
                r = fn(*a, **kw)
                e1.record()
                self.chain[name].append((e0, e1))
                return r
            return call
        for name, fn in self.orig_chain.items():
            setattr(self.K, name, chain_timed(name, fn))

        def gemm(problems, a_kmajor=False, b_kmajor=False, split_k_workspace=True, **kw):
            from twog_gcn_amd.kernels import n_rows
            flops, abytes, tiles128, kmax, wide = 0.0, 0.0, 0, 0, True
            for p in problems:
                A, Cm = p['A'], p['C']
                M, Nn = n_rows(Cm), Cm.shape[-1]
                Kk = n_rows(A) if a_kmajor else A.shape[-1]
                nb = p['batch'][0] if p.get('batch') else 1
                flops += 2.0 * M * Nn * Kk * nb
                abytes += 4.0 * (M * Kk + Nn * Kk + M * Nn) * nb  # each operand once
                tiles128 += nb * math.ceil(M / 128) * math.ceil(Nn / 128)
                kmax = max(kmax, Kk)
                wide = wide and M >= 96 and Nn >= 96
            e0, e1 = self.event(), self.event()
            e0.record()
            self.orig(problems, a_kmajor, b_kmajor, split_k_workspace, **kw)
            e1.record()
            cls = self.K.gemm_last_class()   # the tile class / arithmetic the library actually picked
            kind = (('128x128' if cls & self.K.GEMM_TILE128 else '64x64') + (' bf16x3' if cls & self.K.GEMM_X3 else ''))
            self.records.append((kind, flops, e0, e1, abytes))
            self.shapes.append((a_kmajor, b_kmajor, [(n_rows(p['C']), p['C'].shape[-1], (n_rows(p['A']) if a_kmajor else p['A'].shape[-1]), (p['batch'][0] if p.get('batch') else 1)) for p in problems]))
        self.K.gemm = gemm
        return self

    def __exit__(self, *a):
        self.K.gemm = self.orig
        self.K.attn_fwd, self.K.attn_bwd = self.orig_attn
        for name, fn in self.orig_chain.items():
            setattr(self.K, name, fn)

    def chain_ms(self):
        """Average HIP-event time of one call of each time loop (ms)."""
        torch.cuda.synchronize()
        return {n: (sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None) for n, ev in self.chain.items()}

    def attn_ms(self, kind):
        torch.cuda.synchronize()
        ev = self.attn[kind]
        return sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1) if ev else None

    def detail(self, steps):
        """Per distinct launch signature: calls per step, average time, TFLOP/s (TWOG_BENCH_GEMM_DETAIL=1)."""
        torch.cuda.synchronize()
        agg = {}
        for (kind, flops, e0, e1, _), (akm, bkm, shp) in zip(self.records, self.shapes):
            a = agg.setdefault((kind, akm, bkm, tuple(shp)), [0.0, 0.0, 0])
            a[0] += flops
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        for (kind, akm, bkm, shp), (fl, sec, n) in rows:
            t128 = sum(b * math.ceil(m / 128) * math.ceil(nn / 128) for m, nn, _, b in shp)
            amb = sum(4.0 * (m * k + nn * k + m * nn) * b for m, nn, k, b in shp) / 1e6
            log(f'{kind} {"T" if akm else "N"}{"T" if bkm else "N"} {n / steps:5.1f}/step {sec / n * 1e3:8.3f} ms {fl / sec / 1e12:6.1f} TF  '
                f'tiles128 {t128:5d}  algorithmic {amb:8.1f} MB  {list(shp)[:4]}')

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for kind, flops, e0, e1, abytes in self.records:
            a = agg.setdefault(kind, [0.0, 0.0, 0, 0.0])
            a[0] += flops
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
            a[3] += abytes
        return agg


def chain_model(bs, h):
    """The four time loops of the shipped configuration (csrc/gru.hip, csrc/segrnn.hip) as the library issues them at a real
    batch: per loop the GEMM launches per call, their algorithmic FLOPs and which kernel family serves them. Counts follow
    the loops' code: frame-level BiGRU forward = one fused launch per step (W_hh product + gates); backward = one carry GEMM
    with the next step's gate backward in its epilogue per step; segment level forward = sender MLPs + [W_hh | W_ih[:, msg]]
    projections per step (+ attention and gate kernels); backward = projections' dX, sender-MLP dX with the fused gates."""
    E = {'h': H, 'o': O, 's': 1}
    rows_f = bs * (H + O + 1)
    nsh, nso = 2, 2   # sender MLPs on human states (hh | ho), on object states (oh | oo)
    nmh, nmo = 2, 2   # message blocks received by a human (hh | oh), by an object (ho | oo)
    f = {}
    f['bigru_fwd'] = dict(gemm_launches=T, flops=2.0 * 2 * rows_f * 3 * h * h * T,
                          kernel='gemm_gru_fwd_kernel<2, 2, true> (64 rows x 64 units x 3 gates, bf16x3)')
    f['bigru_bwd'] = dict(gemm_launches=T - 1, flops=2.0 * 2 * rows_f * h * 3 * h * (T - 1),
                          kernel='gemm_gate_bwd_x3s* (64x64, bf16x3, gate backward in the epilogue)')
    send = 2.0 * 2 * (bs * H * nsh * h * h + bs * O * nso * h * h)
    proj = 2.0 * 2 * (bs * H * 3 * h * (h + nmh * h) + bs * O * 3 * h * (h + nmo * h))
    f['segrnn_fwd'] = dict(gemm_launches=2 * T, flops=(send + proj) * T,
                           kernel='gemm_x3s* (sender MLPs, 64x64) + gemm_x3_kernel (projections, 128x128), bf16x3')
    f['segrnn_bwd'] = dict(gemm_launches=2 * T - 1, flops=(send + proj) * T,
                           kernel='gemm_x3s* (projections dX, 64x64) + gemm_gate_bwd_x3s* (sender-MLP dX + gates), bf16x3')
    return f


def latest_pmc_rows(substrs):
    """MFMA-busy of the chain kernels from the latest committed counter pass (profiles/r*_bench_bs64_pmc_summary.csv)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_bs64_pmc_summary.csv')))
    if not files:
        return None, {}
    out = {}
    for r in csv.DictReader(open(files[-1])):
        if any(s_ in r['kernel'] for s_ in substrs):
            out[r['kernel']] = dict(dispatches=int(r['dispatches']), mfma_util_pct=float(r['mfma_util_pct']))
    return 'profiles/' + os.path.basename(files[-1]), out


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(sample_frames=None, sample_clips=4, repeats=3, threads=None):
    """The oracle (port of the reference's PyTorch-CPU path) on a bounded sample of the same workload: `sample_clips`
    full-length clips, 1 warm-up + `repeats` timed forward+backward passes (median reported; forward alone as well) --
    about 30 s of CPU work on the GPU box's 16 granted cores. (`sample_frames` < T shortens the clips; the cost is
    linear in frames.)"""
    sample_frames = T if sample_frames is None else sample_frames
    from oracle import cpu_ref
    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd.models import TGGCN
    from twog_gcn_amd.hostcpu import effective_cpu_count
    # the reference's default resources.num_threads is 32 (conf/config.yaml:9); `value` never uses more than the cgroup
    # grants (threads=None); by_config also times the reference's literal setting of 32 threads on those cores
    granted = effective_cpu_count()
    cores = min(32, granted) if threads is None else threads
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = TGGCN(input_size=(2048 + 4 * N_NODES, 2048), num_classes=(N_CLASSES, None), **CFG)
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone())
          for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    Ts, nb = sample_frames, sample_clips
    x_human = torch.rand(nb, Ts, H, 2048 + 4 * N_NODES, generator=g)
    x_objects = torch.rand(nb, Ts, O, 2048, generator=g)
    mask, seg = torch.ones(nb, O), torch.ones(nb, Ts, H)
    tgt = [torch.randint(0, N_CLASSES, (nb, Ts, H), generator=g) for _ in range(2)]
    t_fwd, t_all = [], []
    for rep in range(repeats + 1):
        for v in sd.values():
            if v.is_floating_point():
                v.grad = None
        t0 = time.perf_counter()
        out = cpu_ref.tggcn_forward(sd, dict(m.cfg), x_human, x_objects, mask, human_segmentation=seg, training=True)
        t1 = time.perf_counter()
        loss = torch.nn.functional.nll_loss(out[4], tgt[0]) + torch.nn.functional.nll_loss(out[5], tgt[1])
        loss.backward()
        t2 = time.perf_counter()
        if rep > 0:  # rep 0 is the warm-up
            t_fwd.append(t1 - t0)
            t_all.append(t2 - t0)
    med = lambda v: sorted(v)[len(v) // 2]
    clips = nb * Ts / T
    return dict(value=clips / med(t_all), unit='clips/s', cores=min(cores, granted), kind='port', cpu_model=_cpu_model(),
                threads=cores, repeats=repeats, warmup=1, sample_clips=nb, sample_frames=Ts,
                protocol='r02: 1 warm-up + 3 timed forward+backward passes of 4 full-length clips, median (round 1 timed ONE '
                         'pass of 8 clips: values are not comparable across that change)',
                sample=f'{nb} clips x {Ts} of {T} frames (H={H},O={O},N={N_NODES},h={CFG["hidden_size"]}); 1 warm-up + '
                       f'{repeats} timed forward+backward passes, {cores} threads on {granted} granted cores, median {med(t_all):.1f} s (all: '
                       f'{", ".join(f"{t:.1f}" for t in t_all)})' + ('' if Ts == T else ', scaled linearly in frames'),
                forward_clips_per_s=clips / med(t_fwd), forward_seconds=[round(t, 2) for t in t_fwd])


def cpu_baseline_all(workload):
    """The `cpu_baseline` object of the bench line: `value` as before (the workload's own shape, as many threads as cores
    are granted, 4 clips, 1 + 3 passes) plus `by_config`: the configs[1] bs8 layout and the reference's literal
    `resources.num_threads: 32` (conf/config.yaml:9) beside it, each on a smaller sample (2 clips, 1 + 2 passes; 2 clips x 30 frames, 1 + 1
    passes for the oversubscribed 32-thread setting) so that the whole leg stays within about two and a half minutes."""
    from twog_gcn_amd.hostcpu import effective_cpu_count
    granted = min(32, effective_cpu_count())
    main_ = cpu_baseline()
    keep = ('value', 'unit', 'threads', 'cores', 'forward_clips_per_s', 'sample')
    by = {f'{workload}_threads{granted}': {k: main_[k] for k in keep}}
    plan = [(workload, 32)] if granted != 32 else []
    if workload == 'c3':
        plan += [('c2', granted)] + ([('c2', 32)] if granted != 32 else [])
    for wl, thr in plan:
        select_workload(wl)
        # (32 threads on fewer granted cores is oversubscribed -- 4.5 x slower on a 16-core grant with 2 clips per pass, 15 x with
        # one: 2 clips x 30 of the 120 frames, 1 + 1 passes, scaled linearly in frames)
        r = (cpu_baseline(sample_frames=30, sample_clips=2, repeats=1, threads=thr) if thr > granted
             else cpu_baseline(sample_clips=2, repeats=2, threads=thr))
        by[f'{wl}_threads{thr}'] = {k: r[k] for k in keep}
    select_workload(workload)
    main_['by_config'] = by
    return main_


def cpu_baseline_in_child(workload='c3', timeout_s=420):
    """Runs the CPU leg in a child process (bounded by a timeout) so it can never take the bench line down."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--workload', workload], capture_output=True,
                           text=True, timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES=''))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith('{'):
                return json.loads(line)
        return dict(value=None, unit='clips/s', cores=None, kind='port', sample=f'failed: {r.stderr[-300:]}')
    except subprocess.TimeoutExpired:
        return dict(value=None, unit='clips/s', cores=None, kind='port', sample=f'timed out after {timeout_s} s')


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def launch_ranks(n):
    """Starts `n` ranks of this script through torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 at a
    free port) as a child process and returns its exit code. Must run before anything initialises the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: RCCL needs it on this driver
    if not _OMP_FROM_USER:
        env.pop('OMP_NUM_THREADS', None)               # each rank sizes its own pools from LOCAL_WORLD_SIZE
    log(f'launching {n} ranks: {" ".join(cmd)}')
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=None, help='clips per GPU per step (default: the workload\'s)')
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='c3',
                    help='c3 = the headline configuration; c2 / c5 = the other single-GPU shapes (informational)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--forward-only', action='store_true', help=argparse.SUPPRESS)  # always reported now
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help='weak: the workload\'s batch per GPU (global = batch * N); strong: the workload\'s batch is the '
                         'GLOBAL batch, batch / N clips per GPU (SURVEY 8e: fixed global bs64 = 8 x 8 at N = 8)')
    args = ap.parse_args()
    wl = select_workload(args.workload)
    args.workload_batch = args.batch   # as given (None = the workload's): the other scaling mode is derived from it
    if args.batch is None:
        args.batch = BS
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_all(args.workload)), flush=True)
        return
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')

    # ---- rank launch. Under torch.distributed.run the environment carries the world; started bare with --gpus N > 1
    # this process becomes the launcher: it starts N ranks as CHILD processes (nothing here has touched the GPU yet --
    # no exec after HIP initialisation) and exits with their return code.
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with\n  python -m torch.distributed.run '
              f'--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port 29500 bench.py '
              f'--gpus {args.gpus} ...\nor start bench.py bare (no WORLD_SIZE) and it spawns the ranks itself',
              file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.scaling == 'strong':
        if args.batch % world:
            ap.error(f'--scaling strong needs the global batch ({args.batch}) divisible by --gpus ({world})')
        args.batch //= world
    import torch.distributed as dist
    # TWOG_BENCH_BACKEND=gloo lets several ranks share one GPU (plumbing check of the N > 1 flow on a 1-GPU box);
    # the measured configuration is always nccl (= RCCL), one rank per GPU
    backend = os.environ.get('TWOG_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    devices = [f'rank{rank}:cuda:{local_rank}']
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, devices[0])
        devices = gathered

    import twog_gcn_amd  # noqa: F401
    from twog_gcn_amd.models import TGGCN
    from twog_gcn_amd.kernels import get_kernels
    from twog_gcn_amd.distributed import DataParallel, FusedAdam
    from twog_gcn_amd.hostcpu import limit_host_threads
    # thread pools sized from this rank's share of the cgroup quota, not from the node's core count (hostcpu.py)
    limit_host_threads(share=int(os.environ.get('LOCAL_WORLD_SIZE', '1')))
    K = get_kernels()
    assert K.name == 'hip'

    torch.manual_seed(0)
    model = TGGCN(input_size=(2048 + 4 * N_NODES, 2048), num_classes=(N_CLASSES, None), **CFG).to(device).train()
    dp = DataParallel(model)
    opt = FusedAdam(dp.flat, lr=1e-4)
    bs = args.batch
    # the reference's criterion for this model (vhoi/losses.py:8-61) with the stage-1 configuration: the budget, BCE and
    # frame-level NLL terms weigh 0, the two segment-level NLL terms weigh 1; all six terms run in one fused launch
    from twog_gcn_amd.losses import select_loss
    criterion, loss_names = select_loss('2G-GCN', 'multiple', 'mphoi', dict(misc={}))

    def make_step(nb):
        """One training step on `nb` resident clips per GPU: returns (step function, the tensors it reads)."""
        x_human, x_objects, mask, targets = synthetic_batch(nb, device, seed=1234 + rank)
        seg = torch.ones(nb, T, H, device=device)  # feeder semantics: impose_segmentation_pattern == 1
        seg_target = torch.zeros(nb, T, H, device=device)
        loss_targets = [seg_target, seg_target, targets[0], targets[1], targets[0], targets[1]]

        def step():
            dp.zero_grad()
            out = model(x_human, x_objects, mask, human_segmentation=seg)
            loss = sum(criterion(out, loss_targets))
            loss.backward()
            dp.all_reduce_gradients()
            opt.step(dp.grad_scale)
            # detached: a caller that keeps the returned loss must not keep the step's autograd node (and the ~3 GB of
            # buffers it saved) alive into the next step -- the next step's buffers would land at other addresses, i.e.
            # under other hipGraph keys: first-sighting launches and a fresh capture of every time loop inside the timed
            # region (seen as one 190 ms step in a default run)
            return loss.detach()
        return step, (x_human, x_objects, mask, seg, loss_targets)

    step, (x_human, x_objects, mask, seg, loss_targets) = make_step(bs)

    def barrier():
        if world > 1:
            dist.barrier()

    # device warm-up, untimed and outside the W warm-up steps: a fresh box runs its first steps at a fraction of the
    # steady rate (measured: 190-240 ms instead of 86 for the first dozen steps -- clocks, page-in, allocator growth, graph
    # captures). Steps are repeated until three in a row agree within 3 % (at most 40); the W warm-up steps and the K
    # timed steps follow.
    if os.environ.get('TWOG_BENCH_DEBUG'):   # where a step spends its time, phase by phase (host clock, synchronised)
        for i in range(3):
            ts = [time.perf_counter()]
            dp.zero_grad()
            out = model(x_human, x_objects, mask, human_segmentation=seg)
            ts.append(time.perf_counter()); torch.cuda.synchronize(); ts.append(time.perf_counter())
            loss = sum(criterion(out, loss_targets))
            torch.cuda.synchronize(); ts.append(time.perf_counter())
            loss.backward()
            ts.append(time.perf_counter()); torch.cuda.synchronize(); ts.append(time.perf_counter())
            dp.all_reduce_gradients()
            opt.step(dp.grad_scale)
            torch.cuda.synchronize(); ts.append(time.perf_counter())
            d = [(b - a) * 1e3 for a, b in zip(ts[:-1], ts[1:])]
            log(f'debug step {i}: forward enqueue {d[0]:.1f} + drain {d[1]:.1f}, loss {d[2]:.1f}, backward enqueue {d[3]:.1f} + '
                f'drain {d[4]:.1f}, adam {d[5]:.1f} ms')
    def settle(step_fn):
        hist = []
        for i in range(40):
            tw = time.perf_counter()
            step_fn()
            torch.cuda.synchronize()
            hist.append(time.perf_counter() - tw)
            settled = len(hist) >= 4 and max(hist[-3:]) <= 1.03 * min(hist[-3:])
            if world > 1:   # every rank must run the same number of steps (each holds a collective): stop when ALL have settled
                flag = torch.tensor([1.0 if settled else 0.0], device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                settled = bool(flag.item() > 0.5)
            if settled:
                break
        return hist

    hist = settle(step)
    log(f'device warm-up: {len(hist)} steps, last {hist[-1] * 1e3:.1f} ms')
    log('model + data ready; warmup')
    for i in range(args.warmup):
        tw = time.perf_counter()
        step()
        torch.cuda.synchronize()
        log(f'warmup step {i}: {time.perf_counter() - tw:.3f} s')
    # one more untimed step under the event wrappers: counts the events a step records, so that every event of the timed
    # region exists before it starts (and the wrappers' code paths are warm)
    with GemmProfiler(K) as dry:
        step()
    torch.cuda.synchronize()
    # A host-composed configuration (constructor defaults: thousands of small GEMM launches per step from Python) is
    # host-bound, and two event records per launch would be a large part of what is measured (general segment loop: 374 ms
    # per step with them, 211 without). Such a step is timed WITHOUT the per-launch events; the roofline figures then come
    # from `prof_steps` extra instrumented steps after the timed region. The BASELINE workloads (~100 host-issued GEMM
    # launches per step; the chains run inside the library) keep the events inside the timed region.
    instrument_timed = dry.used <= 4000
    prof_steps = args.steps if instrument_timed else min(args.steps, 3)
    pool = [torch.cuda.Event(enable_timing=True) for _ in range(dry.used * prof_steps + 64)]
    for e in pool[:64]:
        e.record()
    torch.cuda.synchronize()
    # The interpreter's start-up heap (torch's modules, ~1e6 objects) leaves the collector's generations: a full
    # collection that falls into the timed region then walks only what the steps allocated (microseconds) instead of
    # everything (tens of ms of a stalled launch queue). Collection stays ENABLED -- a leak would still show.
    import gc
    gc.collect()
    if not os.environ.get('TWOG_BENCH_NO_GC_FREEZE'):
        gc.freeze()
    gc_log, gc_t0 = [], [0.0]

    def _gc_watch(phase, info):
        if phase == 'start':
            gc_t0[0] = time.perf_counter()
        else:
            gc_log.append((info['generation'], (time.perf_counter() - gc_t0[0]) * 1e3))
    gc.callbacks.append(_gc_watch)
    barrier()
    torch.cuda.synchronize()
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    import contextlib
    with (GemmProfiler(K, pool) if instrument_timed else contextlib.nullcontext()) as prof:
        t0 = time.perf_counter()
        step_ev[0].record()
        for i in range(args.steps):
            loss = step()
            step_ev[i + 1].record()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    if not instrument_timed:
        with GemmProfiler(K, pool) as prof:
            for i in range(prof_steps):
                step()
            torch.cuda.synchronize()
        log(f'host-composed step ({dry.used // 2} instrumented launches): timed without per-launch events; GEMM figures '
            f'from {prof_steps} extra steps')
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    gc.callbacks.remove(_gc_watch)
    log('collector passes inside the timed region (generation: ms): '
        + (' '.join(f'{g}:{ms:.1f}' for g, ms in gc_log if ms >= 0.5 or g == 2) or 'none above 0.5 ms')
        + f' ({len(gc_log)} passes)')
    agg = prof.summary()
    if os.environ.get('TWOG_BENCH_GEMM_DETAIL') and rank == 0:
        prof.detail(prof_steps)
    log(f'timed region done: {dt / args.steps * 1e3:.1f} ms/step; per step (device): '
        + ' '.join(f'{step_ev[i].elapsed_time(step_ev[i + 1]):.0f}' for i in range(args.steps)))

    # ---- N > 1: what the step's collectives cost, and whether the group really spans N devices
    dist_diag = None
    if world > 1:
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        buf = torch.empty_like(dp.flat.grad)
        for _ in range(2):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        barrier()
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for _ in range(10):
            dist.all_reduce(buf)
        eb.record()
        torch.cuda.synchronize()
        ar_ms = ea.elapsed_time(eb) / 10
        del buf
        # the same step with every collective switched off (each rank trains alone on its shard)
        was = dp.collective
        dp.collective = False
        from twog_gcn_amd import ops as _ops2
        saved_hook = _ops2.get_model_extra(model, 'stage_hook')
        _ops2.set_grad_stage_hook(model, None)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        barrier()
        tn = time.perf_counter()
        for _ in range(max(3, args.steps // 2)):
            step()
        torch.cuda.synchronize()
        t_nc = (time.perf_counter() - tn) / max(3, args.steps // 2) * 1e3
        dp.collective = was
        _ops2.set_grad_stage_hook(model, saved_hook)
        tt = torch.tensor([ar_ms, t_nc], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist_diag = dict(ranks_seen=int(round(float(ones.item()))), allreduce_ms_isolated=float(tt[0]),
                         allreduce_bytes=int(dp.flat.grad.numel() * 4), allreduce_repeats=10,
                         step_ms_without_collectives=float(tt[1]),
                         note='max over ranks; the gradient all-reduce of the timed step is bucketed and overlaps the backward pass')

    # ---- N > 1: the OTHER scaling mode in the same line (SURVEY 8e asks for both). `value` is the mode --scaling names
    # (default weak: the workload's 64 clips on every GPU); the secondary run keeps everything else and changes only the
    # clips per GPU: strong = BASELINE configs[3]'s fixed GLOBAL batch of 64 (64 / N per GPU), weak = 64 per GPU.
    other_scaling = None
    if world > 1:
        other = 'strong' if args.scaling == 'weak' else 'weak'
        base = BS if args.workload_batch is None else args.workload_batch
        nb2 = base // world if other == 'strong' else base
        if nb2 >= 1 and (other == 'weak' or base % world == 0):
            del step
            step2, _keep = make_step(nb2)
            h2 = settle(step2)
            for _ in range(args.warmup):
                step2()
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                step2()
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t2
            tmax = torch.tensor([dt2], device=device, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt2 = float(tmax.item())
            other_scaling = {'scaling': other, 'value': nb2 * world * args.steps / dt2, 'unit': 'clips/s',
                             'ms_per_step': dt2 / args.steps * 1e3, 'per_gpu_batch': nb2, 'global_batch': nb2 * world,
                             'steps': args.steps, 'warmup': args.warmup, 'device_warmup_steps': len(h2)}
            log(f'{other} scaling: {nb2} clips per GPU, {dt2 / args.steps * 1e3:.1f} ms/step')
            del step2, _keep

    # secondary roofline: the geometric-level GCN forward alone (the kernel group the north star's HBM-roofline target
    # names), HIP events around its launches; algorithmic bytes T*(16N + 512N) per clip
    from twog_gcn_amd import ops as _ops
    with torch.no_grad():
        Pd = {n: p_ for n, p_ in model.named_parameters()}
        bn = model.geometry_embedding_gcn.joint_embed.cnn[0].bn  # cloned: the measurement must not touch the statistics
        bn_bufs = dict(running_mean=bn.running_mean.clone(), running_var=bn.running_var.clone(),
                       num_batches_tracked=bn.num_batches_tracked.clone())
        for _ in range(2):
            _ops.geo_gcn_forward(K, Pd, x_human, bs, T, N_NODES, True, bn_bufs, {})
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            _ops.geo_gcn_forward(K, Pd, x_human, bs, T, N_NODES, True, bn_bufs, {})
        e1.record()
        torch.cuda.synchronize()
        gcn_ms = e0.elapsed_time(e1) / 10
    gcn_bytes = bs * T * (16 * N_NODES + 512 * N_NODES)
    # frame-level message attention (the other HBM-bound kernel pair of SURVEY 8d): algorithmic bytes per clip, fp32, each
    # tensor once. Forward, SURVEY's figure: read the entity features [x, h] (4 T E 2h), write every receiver's message
    # block (4 T (2H + 3O) h); the sender messages the kernel also has to read (4 T (2H + 2O + 1) h) are reported beside
    # it. Backward: read d(received blocks) + features + sender messages, write d(sender messages), read-modify-write
    # d(features).
    hh_ = CFG['hidden_size']
    E_ = H + O + 1
    att_feat, att_out, att_msg = 4 * T * (H + O) * 2 * hh_, 4 * T * (2 * H + 3 * O) * hh_, 4 * T * (2 * H + 2 * O + 1) * hh_
    att_fwd_bytes = bs * (4 * T * E_ * 2 * hh_ + att_out)
    att_fwd_bytes_all = bs * (att_feat + att_out + att_msg)
    att_bwd_bytes = bs * (att_out + att_feat + att_msg + att_msg + 2 * att_feat)
    att_fwd_ms, att_bwd_ms = prof.attn_ms('fwd'), prof.attn_ms('bwd')
    chain_ms = prof.chain_ms()


    fwd_only = None
    if True:  # forward-only clips/s is part of every line (north star: ">= 50x the reference CPU forward")
        with torch.no_grad():
            model(x_human, x_objects, mask, human_segmentation=seg)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(max(2, args.steps // 2)):
                model(x_human, x_objects, mask, human_segmentation=seg)
            torch.cuda.synchronize()
            barrier()
            fwd_only = bs * world * max(2, args.steps // 2) / (time.perf_counter() - t1)

    if rank == 0:
        total_gemm_s = sum(v[1] for v in agg.values())
        dom = max(agg.items(), key=lambda kv: kv[1][1])
        kind, (flops, secs, calls, bytes_alg) = dom
        achieved = flops / secs / 1e12 if secs > 0 else 0.0
        # HBM-side bytes per launch of the dominant kernel come from the committed rocprofv3 PMC passes of this same
        # command (tools/bench_pmc.sh -> profiles/): counters cannot be read from inside the process.
        traffic, traffic_src = None, None
        import glob
        tfiles = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles',
                                               'r*_gemm128_hbm_traffic.json')))
        if kind.startswith('128x128') and tfiles:   # the latest round's committed counter passes
            tj = json.load(open(tfiles[-1]))
            traffic = tj['hbm_bytes_per_launch']
            traffic_src = 'profiles/' + os.path.basename(tfiles[-1]) + ': ' + tj['source']
        result = {
            'metric': 'clips/sec fwd+bwd, T=120 N=34 C=512' if args.workload == 'c3' else f'clips/sec fwd+bwd, {wl["name"]} (informational)', 'value': bs * world * args.steps / dt, 'unit': 'clips/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'fp32', 'data': 'synthetic',
            'arithmetic': ('fp32 storage, fp32 accumulation everywhere. THREE GEMM kernel families multiply on the bf16 matrix cores '
                           'after an EXACT three-way bf16 split of every fp32 operand element (6 of the 9 chunk products per '
                           'multiply-add): (1) the 128x128 class gemm_x3_kernel (forward, dX and dW projections, and the segment '
                           'level\'s per-step projections inside the library); (2) the 64x64 chain class gemm_x3s* / '
                           'gemm_gate_bwd_x3s* (launches with >= 96 tiles and K >= 256: sender MLPs, backward carries with the fused '
                           'gate backward); (3) the fused frame-level GRU step gemm_gru_fwd_kernel<2, 2, true>; (4) at small batches (at most one '
                           '16-row tile per wave) the frame-level recurrence as persistent launches bigru_persist_fwd / _bwd_kernel (same '
                           'split, v_mfma_f32_16x16x32_bf16) and the segment-level recurrence as seg_persist_fwd / _bwd_kernel (the same split for the '
                           'products a weight fragment shares between several row tiles, the native fp32 matrix pipe v_mfma_f32_16x16x4_f32 '
                           'for the others; see roofline_chain.loops.*.kernels for what ran). Error against fp64 '
                           'within 1.25x the fp32-MFMA kernels on random operands; on SAME-SIGN operands the bf16 MFMA\'s '
                           'accumulate adds a relative bias (towards zero) of up to 4e-8 (chain class) / 4e-7 at K = 1 536 and '
                           '2e-6 at K = 61 440 (128x128 class) where the fp32 MFMA has 4e-10 '
                           '(tests/test_kernels_gpu.py::test_gemm_x3_*, profiles/r04_x3_products_6_vs_8*.txt). Every other kernel '
                           '(attention, gates, GCN, BatchNorm, loss, Adam, GEMMs of other shapes) computes in fp32. '
                           'TWOG_GEMM_X3=0: native fp32 MFMA throughout'
                           if os.environ.get('TWOG_GEMM_X3', '1') != '0' else 'fp32 throughout (fp32 MFMA; TWOG_GEMM_X3=0)'),
            'config': {'workload': f'{wl["name"]}: bs{bs} per GPU, T={T}, H={H}, O={O}, N={N_NODES}, h={CFG["hidden_size"]}, '
                                   f'classes {N_CLASSES}, ' + ('constructor defaults' + (f' + {wl["cfg"]}' if wl['cfg'] else '') if 'cfg' in wl else '2G-GCN_stage1 parameters'),
                       'global_batch': bs * world, 'per_gpu_batch': bs, 'parallelism': f'dp{world}',
                       'devices': devices, 'collective_backend': backend if world > 1 else None,
                       'step': 'forward + multi-task loss (fused HIP criterion) + backward + gradient all-reduce + fused Adam',
                       'loss_last': float(loss.detach())},
            'roofline': ({
                # X3 kernels: fp32 operands split EXACTLY into three bf16 chunks, six chunk products per multiply-add on
                # v_mfma_f32_32x32x16_bf16, fp32 accumulate (same error against fp64 as the fp32-MFMA kernels, tested).
                # `achieved` = matrix FLOP/s actually executed (6 x the algorithmic 2 M N K), `peak` = dense bf16 MFMA.
                'bound': 'mfma', 'kernel': 'gemm_x3_kernel<*> 128x128 tiles (v_mfma_f32_32x32x16_bf16 on 3 exact bf16 chunks per fp32 operand, 6 products, fp32 accumulate)',
                'achieved': achieved * X3_PRODUCTS, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved * X3_PRODUCTS / PEAK_BF16_MFMA_TFLOPS,
                'fp32_equivalent_tflops': achieved, 'fp32_mfma_peak_tflops': PEAK_FP32_MFMA_TFLOPS,
                'fp32_equivalent_over_fp32_mfma_peak': achieved / PEAK_FP32_MFMA_TFLOPS,
                'note': 'TWOG_GEMM_X3=0 selects the native fp32-MFMA kernels (v_mfma_f32_32x32x2_f32: 0.80 of their 157.3 TFLOP/s peak, slower in wall time)',
            } if kind == '128x128 bf16x3' else {
                'bound': 'mfma', 'kernel': f'gemm_kernel<{kind.replace("x", ",")},*> (fp32 v_mfma_f32_32x32x2_f32)',
                'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / PEAK_FP32_MFMA_TFLOPS,
            }),
            'roofline_common': {'traffic': traffic,
                         'traffic_unit': 'bytes per launch (L2<->fabric, Infinity-Cache hits included)',
                         'traffic_source': traffic_src,
                         'algorithmic_bytes_per_launch': bytes_alg / max(calls, 1),
                         'launches_per_step': calls / prof_steps, 'avg_launch_ms': secs / max(calls, 1) * 1e3,
                         'algorithmic_gflop_per_step': flops / prof_steps / 1e9,
                         'share_of_step_time': secs / prof_steps / (dt / args.steps)},
            'roofline_gcn': {'bound': 'hbm', 'kernel': 'geo_gcn forward (bn_stats, bn_finalize + similarity fold, gcn_fused_fwd, projection GEMM)',
                             'achieved': gcn_bytes / (gcn_ms * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                             'frac': gcn_bytes / (gcn_ms * 1e-3) / 8e12, 'ms_per_batch': gcn_ms,
                             'algorithmic_bytes_per_batch': gcn_bytes,
                             'note': 'arithmetic-bound, not HBM-bound: fp32 MFMA floor of this block (2.41 MFLOP/frame at 157.3 '
                                     'TFLOP/s) = 0.118 ms per bs64 batch; on the bf16 pipe with the exact three-way operand split '
                                     '(6 products per multiply-add, 2 516.6 TFLOP/s) the floor would be 0.044 ms -- still above '
                                     'the 0.043 ms that 40 % of the HBM peak would take for its 137.9 MB, and far above its HBM '
                                     'floor (0.017 ms): at fp32-equivalent arithmetic the north star\'s >= 40 % of the HBM '
                                     'roofline is out of reach for this block whatever the kernel. The bf16 x 3 port of '
                                     'gcn_fused_fwd_kernel (0.13 ms of a 63 ms step at stake) was not built.'},
            'roofline_attn_fwd': None if not att_fwd_ms else {
                'bound': 'hbm', 'kernel': 'attn_fwd_kernel (frame level: 4 relations + geometry, one launch)',
                'achieved': att_fwd_bytes / (att_fwd_ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                'frac': att_fwd_bytes / (att_fwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 'ms_per_launch': att_fwd_ms,
                'algorithmic_bytes_per_launch': att_fwd_bytes,
                'bytes_incl_sender_messages': att_fwd_bytes_all,
                'achieved_incl_sender_messages': att_fwd_bytes_all / (att_fwd_ms * 1e-3) / 1e9},
            'roofline_attn_bwd': None if not att_bwd_ms else {
                'bound': 'hbm', 'kernel': 'attn_bwd_kernel (frame level)',
                'achieved': att_bwd_bytes / (att_bwd_ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                'frac': att_bwd_bytes / (att_bwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 'ms_per_launch': att_bwd_ms,
                'algorithmic_bytes_per_launch': att_bwd_bytes},
            'gemm_classes': {k: {'tflops': (v[0] / v[1] / 1e12 if v[1] else 0.0), 'ms_per_step': v[1] / prof_steps * 1e3,
                                 'launches_per_step': v[2] / prof_steps} for k, v in agg.items()},
            'host_gemm_share_of_step': total_gemm_s / prof_steps / (dt / args.steps),
        }
        result['roofline'].update(result.pop('roofline_common'))   # traffic, launches, shares: common to both kernel families
        # ---- the recurrent chains (they run inside the library: timed per loop call, launch counts from the loops' code)
        cm = chain_model(bs, CFG['hidden_size'])
        from twog_gcn_amd.kernels import get_kernels as _gk
        _K = _gk()
        for _n, _flag in (('bigru_fwd', 'last_bigru_persistent'), ('bigru_bwd', 'last_bigru_bwd_persistent')):
            if getattr(_K, _flag, False):   # small batches: ONE persistent launch for all time steps (csrc/gru_persist.hip)
                cm[_n].update(gemm_launches=1, kernel='bigru_persist_' + _n[6:] + '_kernel (one launch for all steps: W_hh slices resident in LDS as '
                              'bf16x3 fragments, v_mfma_f32_16x16x32_bf16, steps ordered by agent-scope counters)')
        for _n, _flag in (('segrnn_fwd', 'last_segrnn_persistent'), ('segrnn_bwd', 'last_segrnn_bwd_persistent')):
            if getattr(_K, _flag, False):   # small batches: the segment level as ONE persistent launch (csrc/seg_persist.hip)
                cm[_n].update(gemm_launches=1, kernel='seg_persist_' + _n[7:] + '_kernel (one launch for all steps: four roles per slice of 16 '
                              'units, weights resident in registers for the whole sequence, two in-launch hand-offs per step; sender MLPs, '
                              'W_hh and the attention Gram matrix on 3 x bf16 (v_mfma_f32_16x16x32_bf16), the W_ih[:, messages] '
                              'products / backward products on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32))')
        x3_on = os.environ.get('TWOG_GEMM_X3', '1') != '0' and CFG['hidden_size'] >= 256 and bs * (H + O + 1) >= 6 * 64
        src, pmc = latest_pmc_rows(('gemm_gate_bwd_x3s', 'gemm_x3s', 'gemm_gru_fwd', 'gemm_x3d', 'gemm_gate_bwd_x3d'))
        chain_total_ms = sum(v for v in chain_ms.values() if v)
        result['roofline_chain'] = {
            'bound': 'latency (240 dependent time steps) + mfma',
            'loops': {n: {'ms_per_call': chain_ms.get(n), 'gemm_launches_per_call': cm[n]['gemm_launches'],
                          'us_per_time_step': (chain_ms[n] * 1e3 / T if chain_ms.get(n) else None),
                          'algorithmic_gflop': cm[n]['flops'] / 1e9,
                          'fp32_equivalent_tflops': (cm[n]['flops'] / (chain_ms[n] * 1e-3) / 1e12 if chain_ms.get(n) else None),
                          'kernels': cm[n]['kernel']} for n in cm},
            'ms_per_step': chain_total_ms, 'share_of_step_time': chain_total_ms / (dt / args.steps * 1e3),
            'mfma_busy_pct_from_counters': pmc, 'counters_source': src,
            'note': 'per-loop HIP events on the launch stream; each loop call issues its launches inside lib2ggcn_hip.so'}
        host_x3 = sum(v[0] for k, v in agg.items() if 'bf16x3' in k)
        host_all = sum(v[0] for v in agg.values())
        lib = sum(c['flops'] for c in cm.values()) * prof_steps
        result['x3_share_of_gemm_flops'] = ((host_x3 + (lib if x3_on else 0.0)) / (host_all + lib)) if host_all + lib > 0 else None
        # which form of each recurrence served this shape (persistent = one launch for all time steps, csrc/gru_persist.hip /
        # seg_persist.hip; otherwise one launch per dependency level, csrc/gru.hip / segrnn.hip)
        result['recurrence_paths'] = {
            n: ('persistent launch' if getattr(_K, f, False) else 'launch per step')
            for n, f in (('bigru_fwd', 'last_bigru_persistent'), ('bigru_bwd', 'last_bigru_bwd_persistent'),
                         ('segrnn_fwd', 'last_segrnn_persistent'), ('segrnn_bwd', 'last_segrnn_bwd_persistent'))}
        if dist_diag is not None:
            result['distributed_diagnostics'] = dist_diag
        if fwd_only is not None:
            result['forward_only_clips_per_s'] = fwd_only
        if world > 1:
            # both scaling modes in one line: `value` / `scaling` above are the mode --scaling names, this is the other
            result['value_is'] = (f'{args.scaling} scaling: {bs} clips per GPU, global batch {bs * world}')
            result['other_scaling'] = other_scaling
            key = 'strong_scaling' if args.scaling == 'weak' else 'weak_scaling'
            result[key] = other_scaling
        if world == 1 and not args.no_cpu_baseline:
            log('cpu baseline ...')
            result['cpu_baseline'] = cpu_baseline_in_child(args.workload)
            log('cpu baseline done')
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
