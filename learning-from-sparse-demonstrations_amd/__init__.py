"""MI355X-native batched Continuous Pontryagin Differentiable Programming (CPDP).

Drop-in for the hot path of wanxinjin/Learning-from-Sparse-Demonstrations:
``CPDP.COCSys`` / ``COCSys_TimeVarying`` (cocSolver, auxSysSolver), the
``JinEnv`` robot models that feed it, and the loss / parameter-update loop
around it — executed as hand-written HIP kernels for gfx950.
"""
from . import symbolic, codegen, runtime  # noqa: F401
from .symbolic import SX, vertcat, vcat, horzcat, mtimes, dot, jacobian  # noqa: F401
