"""Receding-horizon wrappers with the reference's interface, solving on the GPU.

Mirrors:
  QuadrotorMPC       <- examples/quadrotor/quadrotor_mpc.py:6-124
  CartPoleMPC        <- examples/cartpole/cartpole_mpc.py:122-359 (every mode: LQR only, iLQR only, iLQR + transformer,
                        and the blending of either with the LQR law)
  ControllerSwitcher <- examples/cartpole/cartpole_mpc.py:10-116
The LQR law is a constant of the problem (infinite-horizon DARE on the upright linearisation); it is host arithmetic,
solved once with scipy and cached, not a kernel.

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
        # rebuilt only when an attribute it is made of has changed (optimize() asks on every call: the weights stay
        # assignable like the reference's, and an unchanged problem costs a few byte comparisons)
        b = lambda a: np.asarray(a, dtype=np.float64).tobytes()
        key = (self.dt, self.integration_method, b(self.x_ref), b(self.Q), b(self.R), b(self.Qf), self.alpha, self.beta)
        if getattr(self, "_md_key", None) != key:
            self._md = quadrotor_model(dt=self.dt, integrator=self.integration_method, x_ref=self.x_ref).with_(
                q=tuple(np.diag(self.Q)), r=tuple(np.diag(self.R)), qf=tuple(np.diag(self.Qf)),
                barrier_alpha=float(self.alpha), barrier_beta=float(self.beta))
            self._md_key = key
        return self._md

    def control_step(self, x_current):
        self.ilqr.x0 = x_current
        optimal_u_seq, optimal_x_seq = self.ilqr.optimize(x_ref=self.x_ref, verbose=False)
        self.ilqr.u = optimal_u_seq[1:].copy()               # warm start: shift, hold the last input (:121-122)
        self.ilqr.u.append(optimal_u_seq[-1])
        return optimal_x_seq, optimal_u_seq


class ControllerSwitcher:
    """Blending weight between the nonlinear controller (1) and the LQR law (0) from the norm of the state error:
    0 below epsilon_low, 1 above epsilon_high, linear in between.  The error history (3 entries) and the acceleration
    norm exist as in the reference, whose acceleration damping is commented out (cartpole_mpc.py:98-112)."""

    def __init__(self, epsilon_low=0.05, epsilon_high=0.2, epsilon_dd=0.1, gamma=10.0, use_sigmoid=False):
        self.epsilon_low, self.epsilon_high, self.epsilon_dd = epsilon_low, epsilon_high, epsilon_dd
        self.gamma, self.use_sigmoid = gamma, use_sigmoid
        self.error_history = []

    def update_error(self, error):
        self.error_history.append(error)
        if len(self.error_history) > 3:
            self.error_history.pop(0)

    def compute_current_error_norm(self):
        return float(np.linalg.norm(self.error_history[-1])) if self.error_history else 0.0

    def compute_acceleration_norm(self, dt):
        if len(self.error_history) < 3:
            return 0.0
        e0, e1, e2 = self.error_history
        return float(np.linalg.norm((e2 - 2 * e1 + e0) / dt ** 2))

    def get_blending_weight(self, dt):
        e = self.compute_current_error_norm()
        if e <= self.epsilon_low:
            return 0.0
        if e >= self.epsilon_high:
            return 1.0
        return (e - self.epsilon_low) / (self.epsilon_high - self.epsilon_low)


class CartPoleMPC(_DeviceProblem):
    """Modes as in the reference (flags; `lqr_only` wins, then `ilqr_only`; default = iLQR only):
    lqr_only | ilqr_only | ilqr_tf_only | ilqr_tf_blend (| use_transformer, kept for signature compatibility)."""

    def __init__(self, horizon=30, dt=0.01, integration_method="rk4", transformer_model=None,
                 log_filename="ilqr_log.pkl", switcher_params=None, lqr_only=False, ilqr_only=False, ilqr_tf_only=False,
                 ilqr_tf_blend=False, use_transformer=False, device="cuda:0"):
        self.horizon, self.dt, self.integration_method, self.log_filename = horizon, dt, integration_method, log_filename
        self.lqr_only, self.ilqr_only, self.ilqr_tf_only = lqr_only, ilqr_only, ilqr_tf_only
        self.ilqr_tf_blend, self.use_transformer = ilqr_tf_blend, use_transformer
        self._dev = device
        self.x_ref = np.array([0.0, 0.0, 0.0, 0.0])
        self.Q = np.diag([5.0, 0.1, 10.0, 0.1])
        self.R = np.diag([0.001])
        self.Qf = np.diag([50.0, 6.0, 100.0, 0.1])
        self.Q_lqr = np.diag([1.0, 0.1, 10.0, 0.1])
        self.R_lqr = np.diag([0.001])
        self.phys = dict(m_cart=1.0, m_pole=0.1, length=0.15, gravity=9.81)      # cartpole_dynamics.py:14
        self.u_init = [np.array([0.0]) for _ in range(horizon)]
        if not lqr_only:
            tf_model = None if ilqr_only else (transformer_model if (ilqr_tf_only or ilqr_tf_blend) else None)   # :196-205
            self.ilqr = iLQR_TF(dynamics=self.discrete_dynamics, cost=self.running_cost, cost_final=self.final_cost,
                                x0=self.x_ref, u_init=self.u_init, horizon=self.horizon, tf=tf_model, tol=1e-1,
                                device=device)
        else:
            self.ilqr = None
        sp = switcher_params or {}
        self.switcher = ControllerSwitcher(epsilon_low=sp.get("epsilon_low", 0.5), epsilon_high=sp.get("epsilon_high", 1.5),
                                           epsilon_dd=sp.get("epsilon_dd", 0.01), gamma=sp.get("gamma", 10.0),
                                           use_sigmoid=sp.get("use_sigmoid", False))
        self._lqr_gain = None

    def device_model(self):
        b = lambda a: np.asarray(a, dtype=np.float64).tobytes()
        key = (self.dt, self.integration_method, b(self.x_ref), b(self.Q), b(self.R), b(self.Qf))
        if getattr(self, "_md_key", None) != key:
            self._md = cartpole_model(dt=self.dt, integrator=self.integration_method, x_ref=self.x_ref).with_(
                q=tuple(np.diag(self.Q)), r=tuple(np.diag(self.R)), qf=tuple(np.diag(self.Qf)))
            self._md_key = key
        return self._md

    # ------------------------------------------------------------------ LQR law (cartpole_mpc.py:272-301)
    def linearized_dynamics(self, dt):
        """Euler discretisation of the linearisation about the upright equilibrium (cartpole_dynamics.py:110-142)."""
        M, m, l, g = (self.phys[k] for k in ("m_cart", "m_pole", "length", "gravity"))
        A = np.array([[0.0, 1.0, 0.0, 0.0], [0.0, 0.0, -(m * g) / M, 0.0], [0.0, 0.0, 0.0, 1.0],
                      [0.0, 0.0, ((M + m) * g) / (M * l), 0.0]])
        B = np.array([[0.0], [1.0 / M], [0.0], [-1.0 / (M * l)]])
        return np.eye(4) + dt * A, dt * B

    def compute_linear_lqr_control(self, x_current):
        """u = -K (x - x_ref) with the infinite-horizon discrete LQR gain; the gain is a constant and cached."""
        if self._lqr_gain is None:
            from scipy.linalg import inv, solve_discrete_are
            A_d, B_d = self.linearized_dynamics(self.dt)
            P = solve_discrete_are(A_d, B_d, self.Q_lqr, self.R_lqr)
            self._lqr_gain = inv(self.R_lqr + B_d.T @ P @ B_d) @ (B_d.T @ P @ A_d)
        return (-self._lqr_gain @ (np.asarray(x_current, dtype=np.float64) - self.x_ref)).flatten()

    def _ilqr_step(self, x_current):
        self.ilqr.x0 = x_current
        optimal_u_seq, optimal_x_seq = self.ilqr.optimize(x_ref=self.x_ref)
        self.ilqr.u = optimal_u_seq[1:].copy() + [optimal_u_seq[-1]]   # :331
        return optimal_x_seq, optimal_u_seq[0]

    def control_step(self, x_current):
        """(optimal_x_seq or [], u_final).  The LQR branches apply MINUS compute_linear_lqr_control, exactly as the
        reference does (:322, :342, :355)."""
        if self.lqr_only:
            return [], -self.compute_linear_lqr_control(x_current)
        if self.ilqr_only or self.ilqr_tf_only:
            return self._ilqr_step(x_current)
        # everything else — ilqr_tf_blend, and also a controller built with no flag at all — takes the reference's
        # blending branch (:334-359)
        self.switcher.update_error(np.asarray(x_current, dtype=np.float64) - self.x_ref)
        w = self.switcher.get_blending_weight(self.dt)
        if w <= 0.05:
            return [], -self.compute_linear_lqr_control(x_current)
        x_seq, u_primary = self._ilqr_step(x_current)
        if w >= 0.95:
            return x_seq, u_primary
        return x_seq, w * u_primary + (1 - w) * (-self.compute_linear_lqr_control(x_current))


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

    def run(self, x0, steps, disturbance=None, device_loop=True):
        """Closed loop for `steps` control steps from x0 (B,n); the plant is the device model itself (the reference's
        plant is MuJoCo, out of scope), plus an optional additive state disturbance tensor (steps, B, n).
        Returns dict(x (B,steps+1,n), u (B,steps,m), iters (B,steps)).

        device_loop (default True: where a persistent kernel is the model's fastest form and there is no predictor; "always":
        wherever one exists, i.e. also for a user-compiled model): ALL `steps` control
        steps of all B controllers are ONE launch (quattro_mpc_run_f32) — no host call, synchronisation or tensor
        operation per control step; a controller that converges early goes on to its next control step at once.
        Results are bit-identical to the host-driven loop below."""
        x = torch.as_tensor(x0, dtype=torch.float32, device=self.device).reshape(-1, self.model.n).contiguous()
        sv = self.solver
        use_kernel = (ops.model_can_device_loop(self.model) if device_loop == "always"
                      else bool(device_loop) and ops.model_has_device_loop(self.model))
        if use_kernel and sv.tf is None:
            B, N, n, m = x.shape[0], self.horizon, self.model.n, self.model.m
            if self.u_warm is not None and self.u_warm.shape[0] != B:
                raise ValueError("batch size changed between control steps")
            sv._alloc(B)
            if self.u_warm is None:
                sv.u.zero_()
            else:
                sv.u.copy_(self.u_warm)
            if sv._ws is None:
                sv._ws = ops.workspace(self.model, B, N, self.device)
            x_cur = x.clone()
            traj_x = torch.empty((B, steps + 1, n), dtype=torch.float32, device=self.device)
            traj_u = torch.empty((B, steps, m), dtype=torch.float32, device=self.device)
            traj_it = torch.empty((B, steps), dtype=torch.int32, device=self.device)
            dist_t = None
            if disturbance is not None:
                dist_t = torch.as_tensor(disturbance, dtype=torch.float32, device=self.device).reshape(steps, B, n).contiguous()
            ops.mpc_run(self.model, x_cur, sv.x, sv.u, sv.K, sv.k, sv.cost, sv.tol, sv.max_iter, steps, sv._ws, traj_x,
                        traj_u, traj_it, disturbance=dist_t, alphas=sv.alphas, reg=sv.reg, alpha_idx=sv.alpha_idx,
                        active=sv.active, iters=sv.iters, status=sv.status)
            self.u_warm = sv.u.clone()
            return dict(x=traj_x, u=traj_u, iters=traj_it)
        xs, us, its = [x], [], []
        for s in range(steps):
            _, u_seq, iters = self.control_step(x)
            x = self.plant_step(x, u_seq[:, 0])
            if disturbance is not None:
                x = (x + disturbance[s]).contiguous()
            xs.append(x); us.append(u_seq[:, 0]); its.append(iters)
        return dict(x=torch.stack(xs, dim=1), u=torch.stack(us, dim=1), iters=torch.stack(its, dim=1))
