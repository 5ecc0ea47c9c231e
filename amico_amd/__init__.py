"""amico_amd -- MI355X-native per-voxel AMICO fitter (NODDI / FreeWater / SANDI / CylinderZeppelinBall).

Drop-in for the hot path ``model.fit(evaluation)`` of daducci/AMICO (amico/models.pyx); the
solver runs as hand-written HIP kernels behind the C ABI of include/amico_amd.h.
"""
from . import _capi
from .models import NODDI, FreeWater, SANDI, CylinderZeppelinBall, BaseModel, get_context, get_contexts, set_devices, reset_context  # noqa: F401
from .core import Evaluation  # noqa: F401

__version__ = '0.1.0'
