#!/usr/bin/env python3
"""The geometric-level GCN forward (ops.geo_gcn_forward) of the bench batch alone, for a kernel-level profile:
  rocprofv3 --kernel-trace --stats -d gpurun_out/geo -- python3 tools/geo_prof.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench.select_workload(sys.argv[1] if len(sys.argv) > 1 else 'c3')
import twog_gcn_amd  # noqa
from twog_gcn_amd.models import TGGCN
from twog_gcn_amd.kernels import get_kernels
from twog_gcn_amd import ops
K = get_kernels()
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = TGGCN(input_size=(2048 + 4 * bench.N_NODES, 2048), num_classes=(bench.N_CLASSES, None), **bench.CFG).to(dev).train()
x_human = bench.synthetic_batch(bench.BS, dev, 1234)[0]
Pd = dict(model.named_parameters())
bn = model.geometry_embedding_gcn.joint_embed.cnn[0].bn
bufs = dict(running_mean=bn.running_mean.clone(), running_var=bn.running_var.clone(), num_batches_tracked=bn.num_batches_tracked.clone())
with torch.no_grad():
    for _ in range(3):
        ops.geo_gcn_forward(K, Pd, x_human, bench.BS, bench.T, bench.N_NODES, True, bufs, {})
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.geo_gcn_forward(K, Pd, x_human, bench.BS, bench.T, bench.N_NODES, True, bufs, {})
    e1.record()
    torch.cuda.synchronize()
print(f'geo_gcn_forward: {e0.elapsed_time(e1) / 20:.4f} ms per batch of {bench.BS} clips')
