"""Oracle iLQR: fp64 restatement of the reference solver.  TEST INFRASTRUCTURE ONLY.

Follows `quattro_ilqr_tf/quattro_ilqr_tf.py` of the reference (class iLQR_TF):
  rollout                  <- simulate                       :127-132
  trajectory_cost          <- compute_total_cost             :138-143
  fd_terminal              <- _finite_diff_gradient_final    :149-157
                              _finite_diff_hessian_final     :163-174
  fd_dynamics              <- _compute_dynamics_jacobians    :182-204
  fd_cost                  <- _compute_cost_derivatives      :217-275
  riccati_sweep            <- the algebra of backward_pass   :297-317 (and _segment :343-364)
  backward_pass            <- backward_pass / backward_pass_segment :282-366
  closed_loop_rollout      <- forward_pass                   :377-390
  pack_prompt / unpack_prediction <- optimize                :496-518 (layout quirk)
  optimize                 <- optimize                       :424-591

The finite-difference helpers keep the reference's evaluation points bit for bit
(`x + dx` with a one-hot dx, not `x[i] += eps`, where the reference does so): the
gains are only defined to ~1e-5 by those differences, so pinning the
oracle to the reference needs identical perturbed points, not just the same math.
"""
import numpy as np

LINE_SEARCH_ALPHAS = (1.0, 0.5, 0.25, 0.1, 0.05, 0.01)   # quattro_ilqr_tf.py:440,552
FD_EPS = 1e-5                                              # default eps everywhere in the reference
QUU_REG = 1e-6                                             # quattro_ilqr_tf.py:304,350


# ----------------------------------------------------------------------------- rollouts
def rollout(f, x0, u_seq):
    """Open-loop x_{t+1} = f(x_t, u_t).  Returns (N+1, n)."""
    xs = [np.asarray(x0)]
    for u in u_seq:
        xs.append(f(xs[-1], u))
    return np.array(xs)


def trajectory_cost(L, Lf, x_seq, u_seq):
    """sum_t L(x_t,u_t) in t order, then + Lf(x_N)."""
    total = 0.0
    for t in range(len(u_seq)):
        total += L(x_seq[t], u_seq[t])
    total += Lf(x_seq[-1])
    return total


# ----------------------------------------------------------------------------- finite differences
def fd_terminal(Lf, x, eps=FD_EPS):
    """Central gradient and 4-point Hessian of the terminal cost.
    The Hessian is NOT symmetrised (neither is the reference's)."""
    n = len(x)
    g = np.zeros_like(x)
    for i in range(n):
        xp = np.copy(x)
        xm = np.copy(x)
        xp[i] += eps
        xm[i] -= eps
        g[i] = (Lf(xp) - Lf(xm)) / (2 * eps)
    H = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            a = np.copy(x); a[i] += eps; a[j] += eps
            b = np.copy(x); b[i] += eps; b[j] -= eps
            c = np.copy(x); c[i] -= eps; c[j] += eps
            d = np.copy(x); d[i] -= eps; d[j] -= eps
            H[i, j] = (Lf(a) - Lf(b) - Lf(c) + Lf(d)) / (4 * eps ** 2)
    return g, H


def fd_dynamics(f, x, u, eps=FD_EPS):
    """A = df/dx (n,n), B = df/du (n,m); column i from f(.+eps e_i) - f(.-eps e_i)."""
    n, m = x.shape[0], u.shape[0]
    A = np.zeros((n, n))
    B = np.zeros((n, m))
    for i in range(n):
        e = np.zeros(n)
        e[i] = eps
        A[:, i] = (f(x + e, u) - f(x - e, u)) / (2 * eps)
    for i in range(m):
        e = np.zeros(m)
        e[i] = eps
        B[:, i] = (f(x, u + e) - f(x, u - e)) / (2 * eps)
    return A, B


def fd_cost(L, x, u, eps=FD_EPS):
    """L, L_x, L_u (central) and L_xx, L_uu, L_xu (4-point, every entry on its own)."""
    n, m = x.shape[0], u.shape[0]
    L0 = L(x, u)
    Lx = np.zeros(n)
    Lu = np.zeros(m)
    for i in range(n):
        e = np.zeros(n); e[i] = eps
        Lx[i] = (L(x + e, u) - L(x - e, u)) / (2 * eps)
    for i in range(m):
        e = np.zeros(m); e[i] = eps
        Lu[i] = (L(x, u + e) - L(x, u - e)) / (2 * eps)
    Lxx = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            ei = np.zeros(n); ej = np.zeros(n)
            ei[i] = eps; ej[j] = eps
            Lxx[i, j] = (L(x + ei + ej, u) - L(x + ei - ej, u)
                         - L(x - ei + ej, u) + L(x - ei - ej, u)) / (4 * eps ** 2)
    Luu = np.zeros((m, m))
    for i in range(m):
        for j in range(m):
            ei = np.zeros(m); ej = np.zeros(m)
            ei[i] = eps; ej[j] = eps
            Luu[i, j] = (L(x, u + ei + ej) - L(x, u + ei - ej)
                         - L(x, u - ei + ej) + L(x, u - ei - ej)) / (4 * eps ** 2)
    Lxu = np.zeros((n, m))
    for i in range(n):
        for j in range(m):
            ex = np.zeros(n); eu = np.zeros(m)
            ex[i] = eps; eu[j] = eps
            Lxu[i, j] = (L(x + ex, u + eu) - L(x + ex, u - eu)
                         - L(x - ex, u + eu) + L(x - ex, u - eu)) / (4 * eps ** 2)
    return L0, Lx, Lu, Lxx, Luu, Lxu


def linearize_fd(f, L, Lf, x_seq, u_seq, t_start=0):
    """All derivative blocks the sweep consumes, for t in [t_start, N), plus the terminal pair.
    Arrays are indexed t - t_start.  `lux` is L_xu transposed (quattro_ilqr_tf.py:295)."""
    N = len(u_seq)
    n, m = x_seq.shape[1], np.asarray(u_seq[0]).shape[0]
    S = N - t_start
    out = dict(A=np.zeros((S, n, n)), B=np.zeros((S, n, m)), lx=np.zeros((S, n)), lu=np.zeros((S, m)),
               lxx=np.zeros((S, n, n)), luu=np.zeros((S, m, m)), lux=np.zeros((S, m, n)))
    out["VxN"], out["VxxN"] = fd_terminal(Lf, x_seq[-1])
    for t in range(t_start, N):
        s = t - t_start
        out["A"][s], out["B"][s] = fd_dynamics(f, x_seq[t], np.asarray(u_seq[t]))
        _, out["lx"][s], out["lu"][s], out["lxx"][s], out["luu"][s], lxu = fd_cost(L, x_seq[t], np.asarray(u_seq[t]))
        out["lux"][s] = lxu.T
    return out


# ----------------------------------------------------------------------------- Riccati-like sweep
def riccati_sweep(d, reg=QUU_REG, dtype=np.float64):
    """Backward recursion on ONE trajectory's derivative blocks (dict from linearize_*).

    Q_x = l_x + A^T V_x;  Q_u = l_u + B^T V_x;  Q_xx = l_xx + A^T V_xx A;
    Q_ux = l_ux + B^T V_xx A;  Q_uu = l_uu + B^T V_xx B;
    W = inv(Q_uu + reg I);  k = -W Q_u;  K = -W Q_ux;
    V_x <- Q_x + K^T Q_uu k + K^T Q_u + Q_ux^T k;
    V_xx <- Q_xx + K^T Q_uu K + K^T Q_ux + Q_ux^T K, then symmetrised.
    (un-regularised Q_uu in the V update, as the reference.)
    Returns k (S, m), K (S, m, n), indexed t - t_start.
    """
    A = d["A"].astype(dtype); Bm = d["B"].astype(dtype)
    lx = d["lx"].astype(dtype); lu = d["lu"].astype(dtype)
    lxx = d["lxx"].astype(dtype); luu = d["luu"].astype(dtype); lux = d["lux"].astype(dtype)
    Vx = d["VxN"].astype(dtype); Vxx = d["VxxN"].astype(dtype)
    S, n, m = A.shape[0], A.shape[1], Bm.shape[2]
    k_out = np.zeros((S, m), dtype=dtype)
    K_out = np.zeros((S, m, n), dtype=dtype)
    eye = np.eye(m, dtype=dtype)
    for s in reversed(range(S)):
        Qx = lx[s] + A[s].T @ Vx
        Qu = lu[s] + Bm[s].T @ Vx
        Qxx = lxx[s] + A[s].T @ Vxx @ A[s]
        Qux = lux[s] + Bm[s].T @ Vxx @ A[s]
        Quu = luu[s] + Bm[s].T @ Vxx @ Bm[s]
        W = np.linalg.inv(Quu + dtype(reg) * eye)
        k = -W @ Qu
        K = -W @ Qux
        k_out[s] = k
        K_out[s] = K
        Vx = Qx + K.T @ Quu @ k + K.T @ Qu + Qux.T @ k
        Vxx = Qxx + K.T @ Quu @ K + K.T @ Qux + Qux.T @ K
        Vxx = dtype(0.5) * (Vxx + Vxx.T)
    return k_out, K_out


def riccati_sweep_batched(d, reg=QUU_REG, dtype=np.float64):
    """Same recursion over a batch: every array of `d` has a leading batch axis
    (A: (Bt,S,n,n), ..., VxN: (Bt,n), VxxN: (Bt,n,n)).  Returns k (Bt,S,m), K (Bt,S,m,n)."""
    A = d["A"].astype(dtype); Bm = d["B"].astype(dtype)
    lx = d["lx"].astype(dtype); lu = d["lu"].astype(dtype)
    lxx = d["lxx"].astype(dtype); luu = d["luu"].astype(dtype); lux = d["lux"].astype(dtype)
    Vx = d["VxN"].astype(dtype); Vxx = d["VxxN"].astype(dtype)
    Bt, S, n, m = A.shape[0], A.shape[1], A.shape[2], Bm.shape[3]
    k_out = np.zeros((Bt, S, m), dtype=dtype)
    K_out = np.zeros((Bt, S, m, n), dtype=dtype)
    eye = np.eye(m, dtype=dtype)
    T = lambda M: np.swapaxes(M, -1, -2)
    mv = lambda M, v: np.einsum("bij,bj->bi", M, v)
    for s in reversed(range(S)):
        At, Bt_ = T(A[:, s]), T(Bm[:, s])
        Qx = lx[:, s] + mv(At, Vx)
        Qu = lu[:, s] + mv(Bt_, Vx)
        Qxx = lxx[:, s] + At @ Vxx @ A[:, s]
        Qux = lux[:, s] + Bt_ @ Vxx @ A[:, s]
        Quu = luu[:, s] + Bt_ @ Vxx @ Bm[:, s]
        W = np.linalg.inv(Quu + dtype(reg) * eye)
        k = -mv(W, Qu)
        K = -(W @ Qux)
        k_out[:, s] = k
        K_out[:, s] = K
        Vx = Qx + mv(T(K) @ Quu, k) + mv(T(K), Qu) + mv(T(Qux), k)
        Vxx = Qxx + T(K) @ Quu @ K + T(K) @ Qux + T(Qux) @ K
        Vxx = dtype(0.5) * (Vxx + T(Vxx))
    return k_out, K_out


def backward_pass(f, L, Lf, x_seq, u_seq, t_start=0):
    """FD linearisation + sweep for t in [t_start, N): lists of k (m,), K (m,n), index t - t_start.
    t_start=0 is the reference's backward_pass, t_start>0 its backward_pass_segment."""
    d = linearize_fd(f, L, Lf, x_seq, u_seq, t_start)
    k, K = riccati_sweep(d)
    return [k[s] for s in range(k.shape[0])], [K[s] for s in range(K.shape[0])]


# ----------------------------------------------------------------------------- forward pass
def closed_loop_rollout(f, L, Lf, x0, x_seq, u_seq, k_seq, K_seq, alpha=1.0):
    """u'_t = u_t + alpha (k_t + K_t (x'_t - x_t)); x'_{t+1} = f(x'_t, u'_t); then the total cost.
    alpha scales the feedback term too (quattro_ilqr_tf.py:382-383)."""
    new_x = [np.asarray(x0)]
    new_u = []
    for t in range(len(u_seq)):
        dx = new_x[t] - x_seq[t]
        du = k_seq[t] + K_seq[t] @ dx
        ut = u_seq[t] + alpha * du
        new_u.append(ut)
        new_x.append(f(new_x[t], ut))
    new_x = np.array(new_x)
    return new_x, new_u, trajectory_cost(L, Lf, new_x, new_u)


# ----------------------------------------------------------------------------- transformer glue
def pack_prompt(k_seg, K_seg):
    """Prompt rows = [k (m) | K row-major flattened (m*n)]  (quattro_ilqr_tf.py:498-502).
    NOTE this is NOT the row-interleaved layout the prediction is unpacked with."""
    k_arr = np.array(k_seg)
    K_arr = np.array(K_seg)
    return np.concatenate([k_arr, K_arr.reshape(k_arr.shape[0], -1)], axis=-1)


def unpack_prediction(pred_flat, m, n):
    """(T, m*(1+n)) -> k (T,m) = [:, :, 0], K (T,m,n) = [:, :, 1:] of the (T, m, 1+n) view
    (quattro_ilqr_tf.py:510-514)."""
    T = pred_flat.shape[0]
    v = pred_flat.reshape(T, m, 1 + n)
    return v[:, :, 0], v[:, :, 1:]


# ----------------------------------------------------------------------------- outer loop
def optimize(f, L, Lf, x0, u_init, horizon, x_ref=None, max_iter=100, tol=1e-3,
             tf_predict=None, tf_window=10, state_offset=None, keep_logs=True):
    """One call of the reference's optimize().  Pure iLQR when tf_predict is None, otherwise the
    hybrid: gains for t < N - tf_window come from tf_predict(x_seq - x_ref + offset, prompt),
    the last tf_window steps from the sweep.  Returns (u_seq, final_x_seq, logs)."""
    u_seq = list(u_init)
    logs = []
    n = len(x0)
    if state_offset is None:
        state_offset = np.zeros(n)
    for it in range(max_iter):
        x_seq = rollout(f, x0, u_seq)
        J = trajectory_cost(L, Lf, x_seq, u_seq)
        rec = dict(iteration=it, x_seq=x_seq, current_cost=J)
        if tf_predict is None:
            k_seq, K_seq = backward_pass(f, L, Lf, x_seq, u_seq, 0)
            k_full, K_full = k_seq, K_seq
            rec.update(k_seq=k_seq, K_seq=K_seq)
        else:
            t0 = horizon - tf_window
            k_seg, K_seg = backward_pass(f, L, Lf, x_seq, u_seq, t0)
            prompt = pack_prompt(k_seg, K_seg)
            x_err = x_seq - x_ref + state_offset
            pred = tf_predict(x_err, prompt)
            m_u = k_seg[0].shape[0]
            pk, pK = unpack_prediction(pred, m_u, K_seg[0].shape[1])
            k_full = np.concatenate([pk, np.array(k_seg)], axis=0)
            K_full = np.concatenate([pK, np.array(K_seg)], axis=0)
            rec.update(k_seq_seg=k_seg, K_seq_seg=K_seg, prompt=prompt, prediction=pred)
        found, chosen, new_x, new_u, new_J = False, None, None, None, None
        for alpha in LINE_SEARCH_ALPHAS:
            cx, cu, cJ = closed_loop_rollout(f, L, Lf, x0, x_seq, u_seq, k_full, K_full, alpha)
            if cJ <= J:
                found, chosen, new_x, new_u, new_J = True, alpha, cx, cu, cJ
                u_seq = cu
                break
        rec.update(u_seq=u_seq, alpha=chosen, new_x_seq=new_x, new_u_seq=new_u, new_cost=new_J,
                   found_update=found)
        if keep_logs:
            logs.append(rec)
        if (not found) or (abs(J - new_J) < tol):
            break
    return u_seq, rollout(f, x0, u_seq), logs
