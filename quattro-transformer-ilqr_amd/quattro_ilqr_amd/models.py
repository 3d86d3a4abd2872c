"""Device-model descriptions: the problem definitions a HIP kernel can evaluate.

The reference hands Python callables f, L, Lf to iLQR_TF (quattro_ilqr_tf/quattro_ilqr_tf.py:66-84).  A kernel
cannot call Python, so the two problems the reference ships are built into the library and described here by
their parameters:
  cart-pole : examples/cartpole/cartpole_dynamics.py:14 (physical defaults), cartpole_mpc.py:187-189 (Q, R, Qf)
  quadrotor : examples/quadrotor/quadrotor_dynamics.py:17-23,144, quadrotor_mpc.py:34-46 (Q, R, Qf, barrier)
"""
from dataclasses import dataclass, field, replace

import numpy as np

from . import _lib

_INTEGRATORS = {"euler": _lib.INTEGRATOR_EULER, "rk4": _lib.INTEGRATOR_RK4}
_PARAM_CACHE = {}


@dataclass(frozen=True)
class DeviceModel:
    name: str
    model_id: int
    n: int
    m: int
    dt: float
    integrator: str
    x_ref: tuple
    q: tuple            # diagonal of Q
    r: tuple            # diagonal of R
    qf: tuple           # diagonal of Qf
    barrier_alpha: float = 0.0
    barrier_beta: float = 1.0
    phys: tuple = field(default_factory=tuple)

    def with_(self, **kw):
        if "x_ref" in kw:
            kw["x_ref"] = tuple(float(v) for v in np.asarray(kw["x_ref"]).reshape(-1))
        return replace(self, **kw)

    def c_params(self):
        """The by-value struct the C ABI takes (built once per model: the dataclass is frozen, and filling a ctypes
        struct field by field costs ~20 us of host time per call, which is comparable to a kernel of the hot loop)."""
        cached = _PARAM_CACHE.get(self)
        if cached is not None:
            return cached
        p = self._build_c_params()
        if len(_PARAM_CACHE) > 256:
            _PARAM_CACHE.clear()
        _PARAM_CACHE[self] = p
        return p

    def _build_c_params(self):
        if self.integrator not in _INTEGRATORS:
            raise ValueError(f"Unknown integration method: {self.integrator}")
        p = _lib.ModelParams()
        p.model_id, p.integrator, p.n, p.m = self.model_id, _INTEGRATORS[self.integrator], self.n, self.m
        p.dt, p.barrier_alpha, p.barrier_beta = self.dt, self.barrier_alpha, self.barrier_beta
        for i, v in enumerate(self.phys):
            p.phys[i] = v
        for i in range(self.n):
            p.q[i], p.qf[i], p.x_ref[i] = self.q[i], self.qf[i], self.x_ref[i]
        for a in range(self.m):
            p.r[a] = self.r[a]
        return p


def cartpole_model(dt=0.01, integrator="euler", x_ref=None):
    return DeviceModel(
        name="cartpole", model_id=_lib.MODEL_CARTPOLE, n=4, m=1, dt=float(dt), integrator=integrator,
        x_ref=tuple(np.zeros(4)) if x_ref is None else tuple(float(v) for v in x_ref),
        q=(5.0, 0.1, 10.0, 0.1), r=(0.001,), qf=(50.0, 6.0, 100.0, 0.1),
        phys=(1.0, 0.1, 0.15, 9.81))                       # m_cart, m_pole, length, gravity


def quadrotor_model(dt=0.01, integrator="euler", x_ref=None):
    if x_ref is None:
        x_ref = np.zeros(12)
        x_ref[2] = 0.5
    return DeviceModel(
        name="quadrotor", model_id=_lib.MODEL_QUADROTOR, n=12, m=4, dt=float(dt), integrator=integrator,
        x_ref=tuple(float(v) for v in x_ref),
        q=(10.0, 10.0, 50.0, 1.0, 1.0, 1.0, 10.0, 10.0, 50.0, 1.0, 1.0, 1.0), r=(0.01,) * 4,
        qf=(100.0, 100.0, 500.0, 10.0, 10.0, 10.0, 100.0, 100.0, 500.0, 10.0, 10.0, 10.0),
        barrier_alpha=1000.0, barrier_beta=10.0,
        phys=(1.0, 0.02, 0.02, 0.04, 0.1, 9.81, 0.01))     # mass, Ix, Iy, Iz, arm, gravity, k_yaw


def model_by_name(name, **kw):
    if name == "cartpole":
        return cartpole_model(**kw)
    if name == "quadrotor":
        return quadrotor_model(**kw)
    raise ValueError(f"unknown device model {name!r}")
