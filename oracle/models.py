"""Oracle plant models and stage/terminal costs (fp64 NumPy).  TEST INFRASTRUCTURE ONLY.

Restates, without importing it, the arithmetic of the reference's
  * examples/cartpole/cartpole_dynamics.py:32-108   (CartPoleDynamics)
  * examples/cartpole/cartpole_mpc.py:187-189,244-269 (Q, R, Qf, running/final cost)
  * examples/quadrotor/quadrotor_dynamics.py:47-198 (QuadrotorDynamics)
  * examples/quadrotor/quadrotor_mpc.py:40-46,74-100 (Q, R, Qf, softplus barrier)

The floating-point operation ORDER of the reference is kept on purpose: the
reference differentiates these functions by finite differences with eps=1e-5,
so a 1-ulp change of L(x,u) moves its Hessians by ~1e-7 and its gains by ~1e-5
(SURVEY.md F6).  Bit-identical L/f values are what lets the FD oracle be pinned
to the reference far below that noise floor (tests/test_oracle_golden.py).

Every function accepts a single point (x: (n,), u: (m,)).  Batched analytic
derivatives live in `oracle/linearize.py`.
"""
from dataclasses import dataclass, field

import numpy as np

# model ids shared with include/quattro_hip.h (QUATTRO_MODEL_*)
MODEL_CARTPOLE = 1
MODEL_QUADROTOR = 2

INTEGRATOR_EULER = 0
INTEGRATOR_RK4 = 1


@dataclass
class ModelSpec:
    """Problem definition = plant + cost, the data a device model needs."""

    model_id: int
    n: int
    m: int
    dt: float
    integrator: int
    x_ref: np.ndarray
    Q: np.ndarray          # (n, n) running state weight
    R: np.ndarray          # (m, m) running control weight
    Qf: np.ndarray         # (n, n) terminal weight
    barrier_alpha: float = 0.0   # quadrotor soft barrier weight (quadrotor_mpc.py:45)
    barrier_beta: float = 1.0    # softplus sharpness (quadrotor_mpc.py:46)
    phys: dict = field(default_factory=dict)

    # -- callables in the reference's signature: f(x,u), L(x,u), Lf(x) -------
    def f(self, x, u):
        return discrete_step(self, x, u)

    def L(self, x, u):
        return running_cost(self, x, u)

    def Lf(self, x):
        return final_cost(self, x)


def cartpole_spec(dt=0.01, integrator=INTEGRATOR_EULER, x_ref=None):
    """cartpole_mpc.py:183-191 weights, cartpole_dynamics.py:14 physical defaults."""
    return ModelSpec(
        model_id=MODEL_CARTPOLE, n=4, m=1, dt=dt, integrator=integrator,
        x_ref=np.zeros(4) if x_ref is None else np.asarray(x_ref, dtype=np.float64),
        Q=np.diag([5.0, 0.1, 10.0, 0.1]),
        R=np.diag([0.001]),
        Qf=np.diag([50.0, 6.0, 100.0, 0.1]),
        phys=dict(m_cart=1.0, m_pole=0.1, length=0.15, gravity=9.81),
    )


def quadrotor_spec(dt=0.01, integrator=INTEGRATOR_EULER, x_ref=None):
    """quadrotor_mpc.py:34-46 weights, quadrotor_dynamics.py:17-23,144 physical defaults."""
    if x_ref is None:
        x_ref = np.zeros(12)
        x_ref[2] = 0.5
    return ModelSpec(
        model_id=MODEL_QUADROTOR, n=12, m=4, dt=dt, integrator=integrator,
        x_ref=np.asarray(x_ref, dtype=np.float64),
        Q=np.diag([10.0, 10.0, 50.0, 1.0, 1.0, 1.0, 10.0, 10.0, 50.0, 1.0, 1.0, 1.0]),
        R=np.diag([0.01, 0.01, 0.01, 0.01]),
        Qf=np.diag([100.0, 100.0, 500.0, 10.0, 10.0, 10.0, 100.0, 100.0, 500.0, 10.0, 10.0, 10.0]),
        barrier_alpha=1000.0, barrier_beta=10.0,
        phys=dict(mass=1.0, Ix=0.02, Iy=0.02, Iz=0.04, arm=0.1, gravity=9.81, k_yaw=0.01),
    )


# --------------------------------------------------------------------------- plants
def cartpole_xdot(p, x, u):
    """cartpole_dynamics.py:48-71.  theta = 0 is upright."""
    pos_rate = x[1]
    th = x[2]
    th_rate = x[3]
    force = u[0]
    M, mp, l, g = p["m_cart"], p["m_pole"], p["length"], p["gravity"]
    s = np.sin(th)
    c = np.cos(th)
    mtot = M + mp
    common = (force + mp * l * (th_rate ** 2) * s) / mtot
    th_acc = (-g * s + c * common) / (l * (4.0 / 3.0 - (mp * c ** 2) / mtot))
    pos_acc = common - (mp * l * th_acc * c) / mtot
    return np.array([pos_rate, pos_acc, th_rate, th_acc])


def quadrotor_xdot(p, x, u):
    """quadrotor_dynamics.py:63-164.  x = [p(3), v(3), (phi,theta,psi), (p,q,r)]."""
    vx, vy, vz = x[3], x[4], x[5]
    phi, theta, psi = x[6], x[7], x[8]
    wp, wq, wr = x[9], x[10], x[11]
    u1, u2, u3, u4 = u
    thrust = u1 + u2 + u3 + u4
    mass, Ix, Iy, Iz = p["mass"], p["Ix"], p["Iy"], p["Iz"]

    cphi, sphi = np.cos(phi), np.sin(phi)
    cth, sth = np.cos(theta), np.sin(theta)
    cpsi, spsi = np.cos(psi), np.sin(psi)

    acc_x = (thrust / mass) * (spsi * sphi + cpsi * sth * cphi)
    acc_y = (thrust / mass) * (cpsi * sphi - spsi * sth * cphi)
    acc_z = -p["gravity"] + (thrust / mass) * (cth * cphi)

    phi_rate = wp + wq * sphi * np.tan(theta) + wr * cphi * np.tan(theta)
    theta_rate = wq * cphi - wr * sphi
    psi_rate = (wq * sphi + wr * cphi) / np.cos(theta)

    tau_phi = p["arm"] * ((u2 + u3) - (u1 + u4))
    tau_theta = p["arm"] * ((u1 + u2) - (u3 + u4))
    tau_psi = p["k_yaw"] * (u1 - u2 + u3 - u4)

    wp_rate = ((Iy - Iz) / Ix) * (wq * wr) + (tau_phi / Ix)
    wq_rate = ((Iz - Ix) / Iy) * (wp * wr) + (tau_theta / Iy)
    wr_rate = ((Ix - Iy) / Iz) * (wp * wq) + (tau_psi / Iz)

    return np.array([vx, vy, vz, acc_x, acc_y, acc_z,
                     phi_rate, theta_rate, psi_rate, wp_rate, wq_rate, wr_rate])


def continuous_rate(spec, x, u):
    if spec.model_id == MODEL_CARTPOLE:
        return cartpole_xdot(spec.phys, x, u)
    if spec.model_id == MODEL_QUADROTOR:
        return quadrotor_xdot(spec.phys, x, u)
    raise ValueError(f"unknown model id {spec.model_id}")


def discrete_step(spec, x, u):
    """x_{t+1} = f(x_t, u_t): explicit Euler or classic RK4 with zero-order-hold u
    (cartpole_dynamics.py:93-108, quadrotor_dynamics.py:186-198)."""
    dt = spec.dt
    if spec.integrator == INTEGRATOR_EULER:
        rate = continuous_rate(spec, x, u)
        return x + dt * rate
    if spec.integrator == INTEGRATOR_RK4:
        k1 = continuous_rate(spec, x, u)
        k2 = continuous_rate(spec, x + 0.5 * dt * k1, u)
        k3 = continuous_rate(spec, x + 0.5 * dt * k2, u)
        k4 = continuous_rate(spec, x + dt * k3, u)
        return x + (dt / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
    raise ValueError(f"unknown integrator {spec.integrator}")


# --------------------------------------------------------------------------- costs
def softplus_beta(z, beta):
    """quadrotor_mpc.py:74-80: log(1+exp(beta z))/beta."""
    return np.log1p(np.exp(beta * z)) / beta


def running_cost(spec, x, u):
    """cartpole_mpc.py:255-256 / quadrotor_mpc.py:86-93."""
    dx = x - spec.x_ref
    if spec.model_id == MODEL_CARTPOLE:
        return float(dx @ spec.Q @ dx + u @ spec.R @ u)
    c = dx @ spec.Q @ dx + u @ spec.R @ u
    barrier = np.sum(softplus_beta(-u, spec.barrier_beta) ** 2)
    c += spec.barrier_alpha * barrier
    return c


def final_cost(spec, x):
    """cartpole_mpc.py:268-269 / quadrotor_mpc.py:99-100."""
    dx = x - spec.x_ref
    if spec.model_id == MODEL_CARTPOLE:
        return float(dx @ spec.Qf @ dx)
    return dx @ spec.Qf @ dx
