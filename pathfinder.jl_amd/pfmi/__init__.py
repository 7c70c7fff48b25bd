"""pfmi -- host-side (Python) mirror of the Pathfinder hot-path API over libpfmi.so (MI355X / gfx950).

The product path has NO CPU fallback: importing this package is cheap, but creating an Engine
requires the built HIP library and a visible MI355X, and raises otherwise.
"""
from ._lib import PfmiError, build, lib  # noqa: F401
from .api import (DEFAULT_HISTORY_LENGTH, DEFAULT_NDRAWS_ELBO, ELBOEstimate, MultiPathfinderResult,  # noqa: F401
                  MvNormal, PathfinderResult, PosDefException, PSISResult, UniformSampler, WoodburyPDMat,
                  fit_mvnormals, maximize_elbo, multipathfinder, pathfinder, resample)
from .core import Comm, Engine, StaleHandleError  # noqa: F401
from .hostrng import HostRNG  # noqa: F401
from .optimize import OptimizationTrace, optimize_with_trace  # noqa: F401
from .targets import (CallbackTarget, DeviceCallbackTarget, HostFnTarget, TorchDeviceTarget, FunnelTarget, GaussTarget, t_diag, t_funnel, t_iso, t_lowrank)  # noqa: F401
