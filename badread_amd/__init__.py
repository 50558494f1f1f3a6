"""
badread_amd -- MI355X-native replacement for Badread's per-read simulate hot path.

Only what the path needs lives here: csrc/ (hand-written HIP kernels + the C-ABI library
libbrx_hip.so), engine.py (ctypes binding), and the host-side mirror of the reference interface
(simulate, ErrorModel, QScoreModel, FragmentLengths, Identities, misc, CLI).
"""
from .version import __version__  # noqa: F401
