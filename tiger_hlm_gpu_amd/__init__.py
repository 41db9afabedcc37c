"""tiger_hlm_gpu_amd -- MI355X-native per-hillslope RK45 path of Tiger_HLM_GPU.

Only what the hot path needs: csrc/ (HIP kernels + C ABI -> lib/libhlm_hip.so) and the
host-side mirror of the reference's plug-in / rk45_api interface.  Importing the package
never touches the GPU; compute entry points load the shared library lazily and fail
loudly if it is missing.
"""
from ._lib import (HlmError, MATH_FAST, MATH_STRICT, SYS_CAPPED, SYS_OK, SYS_STIFF, UID_204, UID_DUMMY,
                   LIB_PATH)
from .models import DummyModel, Model204, ToyModel3, Parameters, SP_DTYPE, initial_step_guess
from .params import LookupMapper, load_spatial_params
from . import api

__all__ = ["HlmError", "MATH_FAST", "MATH_STRICT", "SYS_OK", "SYS_STIFF", "SYS_CAPPED", "UID_204",
           "UID_DUMMY", "LIB_PATH", "DummyModel", "Model204", "ToyModel3", "Parameters", "SP_DTYPE",
           "initial_step_guess", "LookupMapper", "load_spatial_params", "api"]
