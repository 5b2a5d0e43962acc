"""Drop-in shim: `import preconditioned_stochastic_gradient_descent as psgd` (hello_psgd.py:5)
resolves to the MI355X-native module of the same name inside the package."""
from psgd_tf_amd.preconditioned_stochastic_gradient_descent import *  # noqa: F401,F403
from psgd_tf_amd.preconditioned_stochastic_gradient_descent import (  # noqa: F401
    _tiny, dtype, UVd, IpUVtmatvec, update_precond_UVd_math_, precond_grad_UVd_math,
    update_precond_dense, precond_grad_dense, update_precond_kron, precond_grad_kron,
    manual_seed, uvd_workspace, uvd_param_index, update_precond_kron_batched, precond_grad_kron_batched,
    update_precond_UVd_math_and_precond_grad, update_precond_splu, precond_grad_splu)
