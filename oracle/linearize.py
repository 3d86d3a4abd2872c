"""Oracle analytic linearisation (batched fp64 NumPy).  TEST INFRASTRUCTURE ONLY.

The reference obtains A, B, l_x, l_u, l_xx, l_uu, l_xu and the terminal V_x, V_xx by finite
differences of Python callables (quattro_ilqr_tf.py:149-275).  A device kernel cannot call Python
and fp32 cannot do eps=1e-5 second differences, so the HIP path evaluates the EXACT derivatives
of the two shipped models.  This file is the fp64 statement of those exact derivatives; it is
checked against the reference's own finite differences in tests/test_oracle_golden.py
(first derivatives to ~1e-9, second derivatives to the reference's FD noise ~1e-6 abs).

Models: oracle/models.py (reference examples/*/…_dynamics.py, …_mpc.py).
All functions take x (..., n) and u (..., m) with arbitrary leading axes.
"""
import numpy as np

from .models import (INTEGRATOR_EULER, INTEGRATOR_RK4, MODEL_CARTPOLE, MODEL_QUADROTOR)


# ------------------------------------------------------------------ continuous-time rates + Jacobians
def _cartpole_rate_jac(p, x, u):
    M, mp, l, g = p["m_cart"], p["m_pole"], p["length"], p["gravity"]
    th, thd, F = x[..., 2], x[..., 3], u[..., 0]
    s, c = np.sin(th), np.cos(th)
    mt = M + mp
    tmp = (F + mp * l * thd ** 2 * s) / mt
    den = l * (4.0 / 3.0 - mp * c ** 2 / mt)
    num = -g * s + c * tmp
    thdd = num / den
    xdd = tmp - mp * l * thdd * c / mt
    rate = np.stack([x[..., 1], xdd, thd, thdd], axis=-1)

    dtmp_th = mp * l * thd ** 2 * c / mt
    dtmp_thd = 2.0 * mp * l * thd * s / mt
    dtmp_F = 1.0 / mt
    dden_th = l * (2.0 * mp * c * s / mt)
    dnum_th = -g * c - s * tmp + c * dtmp_th
    dthdd_th = (dnum_th * den - num * dden_th) / den ** 2
    dthdd_thd = c * dtmp_thd / den
    dthdd_F = c * dtmp_F / den
    kk = mp * l / mt
    dxdd_th = dtmp_th - kk * (dthdd_th * c - thdd * s)
    dxdd_thd = dtmp_thd - kk * c * dthdd_thd
    dxdd_F = dtmp_F - kk * c * dthdd_F

    Jx = np.zeros(x.shape[:-1] + (4, 4))
    Ju = np.zeros(x.shape[:-1] + (4, 1))
    Jx[..., 0, 1] = 1.0
    Jx[..., 1, 2] = dxdd_th
    Jx[..., 1, 3] = dxdd_thd
    Jx[..., 2, 3] = 1.0
    Jx[..., 3, 2] = dthdd_th
    Jx[..., 3, 3] = dthdd_thd
    Ju[..., 1, 0] = dxdd_F
    Ju[..., 3, 0] = dthdd_F
    return rate, Jx, Ju


def _quadrotor_rate_jac(p, x, u):
    mass, Ix, Iy, Iz = p["mass"], p["Ix"], p["Iy"], p["Iz"]
    arm, grav, kyaw = p["arm"], p["gravity"], p["k_yaw"]
    phi, th, psi = x[..., 6], x[..., 7], x[..., 8]
    wp, wq, wr = x[..., 9], x[..., 10], x[..., 11]
    T = u[..., 0] + u[..., 1] + u[..., 2] + u[..., 3]
    cph, sph = np.cos(phi), np.sin(phi)
    cth, sth = np.cos(th), np.sin(th)
    cps, sps = np.cos(psi), np.sin(psi)
    tth = sth / cth
    sec = 1.0 / cth
    tm = T / mass
    rx = sps * sph + cps * sth * cph
    ry = cps * sph - sps * sth * cph
    rz = cth * cph
    qr_mix = wq * sph + wr * cph
    c1, c2, c3 = (Iy - Iz) / Ix, (Iz - Ix) / Iy, (Ix - Iy) / Iz
    tau_phi = arm * ((u[..., 1] + u[..., 2]) - (u[..., 0] + u[..., 3]))
    tau_th = arm * ((u[..., 0] + u[..., 1]) - (u[..., 2] + u[..., 3]))
    tau_psi = kyaw * (u[..., 0] - u[..., 1] + u[..., 2] - u[..., 3])
    rate = np.stack([
        x[..., 3], x[..., 4], x[..., 5],
        tm * rx, tm * ry, -grav + tm * rz,
        wp + qr_mix * tth, wq * cph - wr * sph, qr_mix * sec,
        c1 * wq * wr + tau_phi / Ix, c2 * wp * wr + tau_th / Iy, c3 * wp * wq + tau_psi / Iz], axis=-1)

    Jx = np.zeros(x.shape[:-1] + (12, 12))
    Ju = np.zeros(x.shape[:-1] + (12, 4))
    for i in range(3):
        Jx[..., i, 3 + i] = 1.0
    # accelerations wrt (phi, theta, psi)
    Jx[..., 3, 6] = tm * (sps * cph - cps * sth * sph)
    Jx[..., 3, 7] = tm * (cps * cth * cph)
    Jx[..., 3, 8] = tm * ry
    Jx[..., 4, 6] = tm * (cps * cph + sps * sth * sph)
    Jx[..., 4, 7] = tm * (-sps * cth * cph)
    Jx[..., 4, 8] = -tm * rx
    Jx[..., 5, 6] = -tm * cth * sph
    Jx[..., 5, 7] = -tm * sth * cph
    for j in range(4):
        Ju[..., 3, j] = rx / mass
        Ju[..., 4, j] = ry / mass
        Ju[..., 5, j] = rz / mass
    # Euler-angle kinematics
    dmix_phi = wq * cph - wr * sph
    Jx[..., 6, 6] = dmix_phi * tth
    Jx[..., 6, 7] = qr_mix * sec * sec
    Jx[..., 6, 9] = 1.0
    Jx[..., 6, 10] = sph * tth
    Jx[..., 6, 11] = cph * tth
    Jx[..., 7, 6] = -qr_mix
    Jx[..., 7, 10] = cph
    Jx[..., 7, 11] = -sph
    Jx[..., 8, 6] = dmix_phi * sec
    Jx[..., 8, 7] = qr_mix * sth * sec * sec
    Jx[..., 8, 10] = sph * sec
    Jx[..., 8, 11] = cph * sec
    # body rates
    Jx[..., 9, 10] = c1 * wr
    Jx[..., 9, 11] = c1 * wq
    Jx[..., 10, 9] = c2 * wr
    Jx[..., 10, 11] = c2 * wp
    Jx[..., 11, 9] = c3 * wq
    Jx[..., 11, 10] = c3 * wp
    sgn_phi = np.array([-1.0, 1.0, 1.0, -1.0]) * (arm / Ix)
    sgn_th = np.array([1.0, 1.0, -1.0, -1.0]) * (arm / Iy)
    sgn_psi = np.array([1.0, -1.0, 1.0, -1.0]) * (kyaw / Iz)
    Ju[..., 9, :] = sgn_phi
    Ju[..., 10, :] = sgn_th
    Ju[..., 11, :] = sgn_psi
    return rate, Jx, Ju


def rate_jac(spec, x, u):
    if spec.model_id == MODEL_CARTPOLE:
        return _cartpole_rate_jac(spec.phys, x, u)
    if spec.model_id == MODEL_QUADROTOR:
        return _quadrotor_rate_jac(spec.phys, x, u)
    raise ValueError(spec.model_id)


# ------------------------------------------------------------------ discrete step + Jacobians
def step_jac(spec, x, u):
    """x_next, A = d x_next / d x, B = d x_next / d u for Euler or RK4 (zero-order-hold u)."""
    dt = spec.dt
    n = spec.n
    I = np.eye(n)
    if spec.integrator == INTEGRATOR_EULER:
        r, Jx, Ju = rate_jac(spec, x, u)
        return x + dt * r, I + dt * Jx, dt * Ju
    if spec.integrator == INTEGRATOR_RK4:
        k1, J1x, J1u = rate_jac(spec, x, u)
        k2, J2x, J2u = rate_jac(spec, x + 0.5 * dt * k1, u)
        D2x = J2x @ (I + 0.5 * dt * J1x)
        D2u = J2x @ (0.5 * dt * J1u) + J2u
        k3, J3x, J3u = rate_jac(spec, x + 0.5 * dt * k2, u)
        D3x = J3x @ (I + 0.5 * dt * D2x)
        D3u = J3x @ (0.5 * dt * D2u) + J3u
        k4, J4x, J4u = rate_jac(spec, x + dt * k3, u)
        D4x = J4x @ (I + dt * D3x)
        D4u = J4x @ (dt * D3u) + J4u
        xn = x + (dt / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        A = I + (dt / 6.0) * (J1x + 2.0 * D2x + 2.0 * D3x + D4x)
        B = (dt / 6.0) * (J1u + 2.0 * D2u + 2.0 * D3u + D4u)
        return xn, A, B
    raise ValueError(spec.integrator)


def step(spec, x, u):
    return step_jac(spec, x, u)[0]


# ------------------------------------------------------------------ costs
def _softplus(z, beta):
    # stable log(1+exp(beta z))/beta
    bz = beta * z
    return (np.maximum(bz, 0.0) + np.log1p(np.exp(-np.abs(bz)))) / beta


def _sigmoid(z):
    return 0.5 * (1.0 + np.tanh(0.5 * z))


def stage_cost(spec, x, u):
    dx = x - spec.x_ref
    c = np.einsum("...i,ij,...j->...", dx, spec.Q, dx) + np.einsum("...i,ij,...j->...", u, spec.R, u)
    if spec.barrier_alpha != 0.0:
        c = c + spec.barrier_alpha * np.sum(_softplus(-u, spec.barrier_beta) ** 2, axis=-1)
    return c


def terminal_cost(spec, x):
    dx = x - spec.x_ref
    return np.einsum("...i,ij,...j->...", dx, spec.Qf, dx)


def stage_cost_derivs(spec, x, u):
    """l_x, l_u, l_xx, l_uu, l_ux of L = dx^T Q dx + u^T R u + alpha sum softplus_beta(-u_i)^2."""
    dx = x - spec.x_ref
    Qs = spec.Q + spec.Q.T
    Rs = spec.R + spec.R.T
    lead = x.shape[:-1]
    lx = dx @ Qs.T
    lu = u @ Rs.T
    lxx = np.broadcast_to(Qs, lead + Qs.shape).copy()
    luu = np.broadcast_to(Rs, lead + Rs.shape).copy()
    lux = np.zeros(lead + (spec.m, spec.n))
    if spec.barrier_alpha != 0.0:
        b = spec.barrier_beta
        sp = _softplus(-u, b)
        sg = _sigmoid(-b * u)
        lu = lu + spec.barrier_alpha * (-2.0 * sp * sg)
        g2 = 2.0 * sg * sg + 2.0 * sp * b * sg * (1.0 - sg)
        idx = np.arange(spec.m)
        luu[..., idx, idx] += spec.barrier_alpha * g2
    return lx, lu, lxx, luu, lux


def terminal_derivs(spec, xN):
    dx = xN - spec.x_ref
    Qs = spec.Qf + spec.Qf.T
    return dx @ Qs.T, np.broadcast_to(Qs, xN.shape[:-1] + Qs.shape).copy()


def linearize_analytic(spec, x_seq, u_seq, t_start=0):
    """Derivative blocks for t in [t_start, N) with leading batch axes kept:
    x_seq (..., N+1, n), u_seq (..., N, m) -> dict like oracle.ilqr.linearize_fd (time axis second-to-last
    of the leading shape, i.e. A has shape (..., S, n, n))."""
    x_seq = np.asarray(x_seq, dtype=np.float64)
    u_seq = np.asarray(u_seq, dtype=np.float64)
    xs = x_seq[..., t_start:-1, :]
    us = u_seq[..., t_start:, :]
    _, A, B = step_jac(spec, xs, us)
    lx, lu, lxx, luu, lux = stage_cost_derivs(spec, xs, us)
    VxN, VxxN = terminal_derivs(spec, x_seq[..., -1, :])
    return dict(A=A, B=B, lx=lx, lu=lu, lxx=lxx, luu=luu, lux=lux, VxN=VxN, VxxN=VxxN)


# ------------------------------------------------------------------ batched rollouts (analytic models)
def rollout_batched(spec, x0, u_seq):
    """x0 (Bt,n), u_seq (Bt,N,m) -> x_seq (Bt,N+1,n), cost (Bt,)."""
    Bt, N = u_seq.shape[0], u_seq.shape[1]
    xs = np.zeros((Bt, N + 1, spec.n))
    xs[:, 0] = x0
    J = np.zeros(Bt)
    for t in range(N):
        J = J + stage_cost(spec, xs[:, t], u_seq[:, t])
        xs[:, t + 1] = step(spec, xs[:, t], u_seq[:, t])
    J = J + terminal_cost(spec, xs[:, N])
    return xs, J


def closed_loop_rollout_batched(spec, x0, x_seq, u_seq, k, K, alpha):
    """Batched forward pass: alpha scalar or (Bt,).  Returns new_x, new_u, cost."""
    Bt, N = u_seq.shape[0], u_seq.shape[1]
    alpha = np.broadcast_to(np.asarray(alpha, dtype=np.float64), (Bt,))[:, None]
    nx = np.zeros_like(x_seq)
    nu = np.zeros_like(u_seq)
    nx[:, 0] = x0
    J = np.zeros(Bt)
    for t in range(N):
        dx = nx[:, t] - x_seq[:, t]
        du = k[:, t] + np.einsum("bij,bj->bi", K[:, t], dx)
        nu[:, t] = u_seq[:, t] + alpha * du
        J = J + stage_cost(spec, nx[:, t], nu[:, t])
        nx[:, t + 1] = step(spec, nx[:, t], nu[:, t])
    J = J + terminal_cost(spec, nx[:, N])
    return nx, nu, J
