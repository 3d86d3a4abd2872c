"""quattro_ilqr_amd — host side of the MI355X-native Quattro iLQR hot path.

Python mirrors of the reference's interface (iLQR_TF, TransformerILQR, the two MPC wrappers) over the HIP
library libquattro_hip.so (C ABI in include/quattro_hip.h).  No CPU fallback: without the library the ops raise.
"""
from . import _lib, models, ops  # noqa: F401
from .models import DeviceModel, cartpole_model, model_by_name, quadrotor_model  # noqa: F401
from .user_model import UserDeviceModel, compile_model  # noqa: F401
from .solver import QuattroILQR, iLQR_TF  # noqa: F401,E402
from .mpc import BatchedMPC, CartPoleMPC, ControllerSwitcher, QuadrotorMPC  # noqa: F401,E402
from .transformer import TransformerILQR  # noqa: F401,E402
from . import datagen, training  # noqa: F401,E402
