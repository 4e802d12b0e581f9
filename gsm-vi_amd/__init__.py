"""gsmvi_amd: MI355X-native GSM / BaM update engine behind the GSM-VI Python API.

Mirrors the reference's public surface for the hot path (reference file:line):
    GSM, gsm_update                              gsmvi/gsm.py:31-133, gsmvi/gsm_numpy.py:27-129
    BaM, bam_update, bam_lowrank_update,
    Regularizers                                 gsmvi/bam.py:31-274
    KLMonitor (diagnostics callback, host side),
    DeviceKLMonitor (same protocol, on the GPU)  gsmvi/monitors.py:43-125
    lbfgs_init, ADVI (initialiser and the ELBO
    baseline of the examples; off the hot path)  gsmvi/initializers.py:5-17, gsmvi/advi.py:8-112
All GSM / BaM numerics run in hand-written HIP kernels (libgsmvi_hip.so, C ABI in include/gsmvi_hip.h)
called through ctypes; torch is used for device memory, streams and torch.distributed only.
There is no CPU fallback: without the library or a GPU every compute entry point raises.
(lbfgs_init and ADVI are the examples' comparison tools, not the update path: scipy / torch autograd.)
"""
from ._lib import load_library, library_path, GsmviError            # noqa: F401
from .engine import HipEngine, get_engine                            # noqa: F401
from .gsm import GSM, gsm_update                                     # noqa: F401
from .bam import BaM, bam_update, bam_lowrank_update, Regularizers   # noqa: F401
from .targets import GaussianTarget, device_score, score_from_logp   # noqa: F401
from .monitors import KLMonitor, DeviceKLMonitor                     # noqa: F401
from .initializers import lbfgs_init                                 # noqa: F401
from .advi import ADVI                                               # noqa: F401

__version__ = "0.1.0"
