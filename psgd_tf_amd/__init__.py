"""psgd_tf_amd -- MI355X-native engine for the PSGD preconditioner hot path.

Contents (only what the hot path needs):
  csrc/                                            HIP kernels + the C ABI (include/psgd_hip.h)
  _lib.py                                          ctypes binding / in-tree build
  preconditioned_stochastic_gradient_descent.py    host-side mirror of the reference module
  kron.py                                          Kronecker dispatch + formats
  sharded.py                                       row-sharded multi-GPU UVd (torch.distributed / RCCL)
"""
from . import preconditioned_stochastic_gradient_descent as psgd  # noqa: F401

__all__ = ["psgd"]
