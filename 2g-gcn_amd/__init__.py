"""2g-gcn_amd: MI355X-native (gfx950) implementation of the 2G-GCN hot path.

The directory name is not a Python identifier, so import it through the repo-root shim ``import twog_gcn_amd`` (or
``importlib``); inside the package everything uses relative imports.

  models.py        TGGCN / select_model -- drop-in for the reference's vhoi.models (same ctor kwargs, forward, state_dict)
  ops.py           the forward/backward composition (one autograd node) on top of the kernels
  kernels.py       tensor-level interface to lib2ggcn_hip.so (C ABI in include/twog_gcn.h)
  data_loading.py  per-clip batching: tensor assembly, fetcher, feeder (mirror of vhoi.data_loading)
  distributed.py   batch data-parallel wrapper: RCCL gradient all-reduce over xGMI
  csrc/            the HIP kernels
"""
from .models import TGGCN, select_model, build_mlp  # noqa: F401
from .losses import select_loss, multi_task_loss  # noqa: F401

__all__ = ['TGGCN', 'select_model', 'build_mlp', 'select_loss', 'multi_task_loss']
