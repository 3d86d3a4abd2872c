"""Receding-horizon wrappers with the reference's interface, solving on the GPU.

Mirrors (problem definition + control_step + warm-start shift only; the DARE/LQR blending modes of the cart-pole
example are out of scope):
  QuadrotorMPC  <- examples/quadrotor/quadrotor_mpc.py:6-124
  CartPoleMPC   <- examples/cartpole/cartpole_mpc.py:122-332 (iLQR-only / iLQR+TF-only branches)

The `discrete_dynamics` / `running_cost` / `final_cost` methods exist because the reference's callers read them; they
are handles that identify the built-in device model to iLQR_TF (via `device_model()`), and are NOT evaluated by the
solver.  Calling them directly evaluates one point on the GPU through the same kernels.
"""
import numpy as np
import torch

from . import ops
from .models import cartpole_model, quadrotor_model
from .solver import QuattroILQR, iLQR_TF


class _DeviceProblem:
    """Shared plumbing: the three reference callables, evaluated on the device for a single point."""

    _dev = "cuda:0"

    def device_model(self):
        raise NotImplementedError

    def _pt(self, v, d):
        return torch.as_tensor(np.asarray(v, dtype=np.float32).reshape(1, d), device=self._dev)

    def discrete_dynamics(self, x, u):
        md = self.device_model()
        xs, _ = ops.simulate(md, self._pt(x, md.n), self._pt(u, md.m).reshape(1, 1, md.m))
        return xs[0, 1].double().cpu().numpy()

    def running_cost(self, x, u):
        # L(x,u) = total cost of a one-step sequence minus the terminal term of its end state
        md = self.device_model()
        xx = torch.cat([self._pt(x, md.n), self._pt(self.x_ref, md.n)], dim=0).reshape(1, 2, md.n)
        return float(ops.total_cost(md, xx, self._pt(u, md.m).reshape(1, 1, md.m))[0].item())

    def final_cost(self, x):
        md = self.device_model().with_(q=(0.0,) * len(self.x_ref), r=(0.0,) * self.device_model().m, barrier_alpha=0.0)
        xx = torch.cat([self._pt(self.x_ref, md.n), self._pt(x, md.n)], dim=0).reshape(1, 2, md.n)
        return float(ops.total_cost(md, xx, torch.zeros((1, 1, md.m), device=self._dev))[0].item())


class QuadrotorMPC(_DeviceProblem):
    def __init__(self, horizon=30, dt=0.01, integration_method="rk4", transformer_model=None,
                 log_filename="quad_ilqr_log.pkl", device="cuda:0"):
        self.horizon, self.dt, self.integration_method, self.log_filename = horizon, dt, integration_method, log_filename
        self._dev = device
        self.x_ref = np.zeros(12)
        self.x_ref[2] = 0.5
        self.Q = np.diag([10.0, 10.0, 50.0, 1.0, 1.0, 1.0, 10.0, 10.0, 50.0, 1.0, 1.0, 1.0])
        self.R = np.diag([0.01, 0.01, 0.01, 0.01])
        self.Qf = np.diag([100.0, 100.0, 500.0, 10.0, 10.0, 10.0, 100.0, 100.0, 500.0, 10.0, 10.0, 10.0])
        self.alpha, self.beta = 1000.0, 10.0
        self.u_init = [np.zeros(4) for _ in range(horizon)]
        self.transformer_model = transformer_model
        self.ilqr = iLQR_TF(dynamics=self.discrete_dynamics, cost=self.running_cost, cost_final=self.final_cost,
                            x0=self.x_ref, u_init=self.u_init, horizon=horizon, tf=self.transformer_model,
                            device=device)
        offset = self.ilqr.get_state_offset()
        offset[2] = 0.5                                      # quadrotor_mpc.py:64-66
        self.ilqr.set_state_offset(offset)

    def device_model(self):
        return quadrotor_model(dt=self.dt, integrator=self.integration_method, x_ref=self.x_ref).with_(
            q=tuple(np.diag(self.Q)), r=tuple(np.diag(self.R)), qf=tuple(np.diag(self.Qf)),
            barrier_alpha=float(self.alpha), barrier_beta=float(self.beta))

    def control_step(self, x_current):
        self.ilqr.x0 = x_current
        optimal_u_seq, optimal_x_seq = self.ilqr.optimize(x_ref=self.x_ref, verbose=False)
        self.ilqr.u = optimal_u_seq[1:].copy()               # warm start: shift, hold the last input (:121-122)
        self.ilqr.u.append(optimal_u_seq[-1])
        return optimal_x_seq, optimal_u_seq


class CartPoleMPC(_DeviceProblem):
    def __init__(self, horizon=30, dt=0.01, integration_method="rk4", transformer_model=None,
                 log_filename="ilqr_log.pkl", ilqr_only=False, ilqr_tf_only=False, device="cuda:0"):
        self.horizon, self.dt, self.integration_method, self.log_filename = horizon, dt, integration_method, log_filename
        self.ilqr_only, self.ilqr_tf_only = ilqr_only, ilqr_tf_only
        self._dev = device
        self.x_ref = np.array([0.0, 0.0, 0.0, 0.0])
        self.Q = np.diag([5.0, 0.1, 10.0, 0.1])
        self.R = np.diag([0.001])
        self.Qf = np.diag([50.0, 6.0, 100.0, 0.1])
        self.u_init = [np.array([0.0]) for _ in range(horizon)]
        tf_model = transformer_model if ilqr_tf_only else None      # cartpole_mpc.py:198-205
        self.ilqr = iLQR_TF(dynamics=self.discrete_dynamics, cost=self.running_cost, cost_final=self.final_cost,
                            x0=self.x_ref, u_init=self.u_init, horizon=self.horizon, tf=tf_model, tol=1e-1,
                            device=device)

    def device_model(self):
        return cartpole_model(dt=self.dt, integrator=self.integration_method, x_ref=self.x_ref).with_(
            q=tuple(np.diag(self.Q)), r=tuple(np.diag(self.R)), qf=tuple(np.diag(self.Qf)))

    def control_step(self, x_current):
        self.ilqr.x0 = x_current
        optimal_u_seq, optimal_x_seq = self.ilqr.optimize(x_ref=self.x_ref)
        u_final = optimal_u_seq[0]
        self.ilqr.u = optimal_u_seq[1:].copy() + [optimal_u_seq[-1]]   # :331
        return optimal_x_seq, u_final


class BatchedMPC:
    """B receding-horizon controllers advancing together on the device: what the reference does with
    one `control_step` per simulator tick and per process (quadrotor_mpc.py:102-124, cartpole_mpc.py:326-332;
    `multiprocessing.Pool` in training_data_collection.py:298-305), here as batched kernels with no host round trip per
    controller: solve -> apply u_0 -> shift the warm start (u[1:] + [u[-1]])."""

    def __init__(self, model, horizon, max_iter=100, tol=1e-3, tf=None, tf_window=10, device="cuda:0",
                 state_offset=None, check_every=4):
        self.model, self.horizon = model, int(horizon)
        self.solver = QuattroILQR(model, horizon, max_iter=max_iter, tol=tol, tf=tf, tf_window=tf_window,
                                  device=device, state_offset=state_offset, check_every=check_every)
        self.device = self.solver.device
        self.u_warm = None                      # (B, N, m) warm start for the next control step

    def control_step(self, x_current, x_ref=None):
        """x_current (B, n) -> (x_seq (B,N+1,n), u_seq (B,N,m), iters (B,)) as fresh device tensors; keeps the shifted
        control sequence as the next warm start."""
        x_current = torch.as_tensor(x_current, dtype=torch.float32, device=self.device).reshape(-1, self.model.n)
        B = x_current.shape[0]
        if self.u_warm is not None and self.u_warm.shape[0] != B:
            raise ValueError("batch size changed between control steps")
        out = self.solver.solve(x_current, self.u_warm, x_ref=x_ref)
        u = out["u"]
        self.u_warm = torch.cat([u[:, 1:], u[:, -1:]], dim=1).contiguous()
        return out["x"].clone(), u.clone(), out["iters"].clone()

    def plant_step(self, x, u0):
        """One step of the (same) device dynamics: x (B,n), u0 (B,m) -> x_next (B,n)."""
        xs, _ = ops.simulate(self.model, x.contiguous(), u0.reshape(-1, 1, self.model.m).contiguous())
        return xs[:, 1].contiguous()

    def run(self, x0, steps, disturbance=None):
        """Closed loop for `steps` control steps from x0 (B,n); the plant is the device model itself (the reference's
        plant is MuJoCo, out of scope), plus an optional additive state disturbance tensor (steps, B, n).
        Returns dict(x (B,steps+1,n), u (B,steps,m), iters (B,steps))."""
        x = torch.as_tensor(x0, dtype=torch.float32, device=self.device).reshape(-1, self.model.n).contiguous()
        xs, us, its = [x], [], []
        for s in range(steps):
            _, u_seq, iters = self.control_step(x)
            x = self.plant_step(x, u_seq[:, 0])
            if disturbance is not None:
                x = (x + disturbance[s]).contiguous()
            xs.append(x); us.append(u_seq[:, 0]); its.append(iters)
        return dict(x=torch.stack(xs, dim=1), u=torch.stack(us, dim=1), iters=torch.stack(its, dim=1))
