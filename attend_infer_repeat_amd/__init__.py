"""attend_infer_repeat_amd -- MI355X-native AIR (Attend, Infer, Repeat) hot path.

Same API surface as akosiorek/attend_infer_repeat's `AIRCell` / `AIRModel` / `AIRonMNIST`, backed by hand-written
gfx950 HIP kernels behind a C ABI (include/air_hip.h -> attend_infer_repeat_amd/lib/libair_hip.so).
"""
__version__ = "0.1.0"

from . import runtime_env as _runtime_env       # HIP runtime settings: must be in the environment before the first HIP call

_runtime_env.apply()
