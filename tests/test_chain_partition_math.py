"""The algorithm behind vdo_slam_amd/csrc/ba_solve.hip pchain_solve_partitioned / k_pchain_prefix, restated in numpy and checked against the
plain recurrences (no GPU): the block-bidiagonal substitutions of the pose-chain preconditioner are LINEAR recurrences, so a chain can be cut
into segments that run from a zero input in parallel and are corrected afterwards with prefix products of the factor blocks.
(The HIP kernels themselves are compared across partitions on the GPU: tests/test_ba_gpu.py::test_pose_chain_solver_is_the_same_operator_...)"""
import numpy as np
import pytest


def seg_len(n, nwave):                      # pc_seg_len of ba_solve.hip
    g = (n + nwave - 1) // nwave
    return max(8, g)


def solve_plain(L, Dinv, r):
    n = len(r)
    y = np.zeros_like(r); y[0] = r[0]
    for k in range(1, n):
        y[k] = r[k] - L[k] @ y[k - 1]
    w = np.einsum("kij,kj->ki", Dinv, y)
    z = np.zeros_like(r); z[n - 1] = w[n - 1]
    for k in range(n - 2, -1, -1):
        z[k] = w[k] - L[k + 1].T @ z[k + 1]
    return z


def solve_partitioned(L, Dinv, r, nwave):
    n = len(r)
    G = seg_len(n, nwave); S = (n + G - 1) // G
    segs = [(s * G, min((s + 1) * G, n)) for s in range(S)]
    # prefix products, once per factorisation: P_k = (-L_k) ... (-L_k0) inside segments s >= 1 ; Q_k = (-L_{k+1}^T) ... (-L_{e+1}^T) inside s <= S-2
    P = np.zeros_like(L); Q = np.zeros_like(L)
    for s, (a, b) in enumerate(segs):
        if s >= 1:
            P[a] = -L[a]
            for k in range(a + 1, b):
                P[k] = -L[k] @ P[k - 1]
        if s <= S - 2:
            Q[b - 1] = -L[b].T
            for k in range(b - 2, a - 1, -1):
                Q[k] = -L[k + 1].T @ Q[k + 1]
    # forward: every segment from a zero input
    y = np.zeros_like(r)
    for a, b in segs:
        y[a] = r[a]
        for k in range(a + 1, b):
            y[k] = r[k] - L[k] @ y[k - 1]
    # boundary pass (wave 0), then the recurrence-free correction
    Y = [y[segs[0][1] - 1].copy()]
    for s in range(1, S - 1):
        e = segs[s][1] - 1
        Y.append(y[e] + P[e] @ Y[-1])
    for s in range(1, S):
        a, b = segs[s]
        y[a:b] += np.einsum("kij,j->ki", P[a:b], Y[s - 1])
    w = np.einsum("kij,kj->ki", Dinv, y)
    # backward: the same with Q and the first z of the segment behind
    z = np.zeros_like(r)
    for a, b in segs:
        z[b - 1] = w[b - 1]
        for k in range(b - 2, a - 1, -1):
            z[k] = w[k] - L[k + 1].T @ z[k + 1]
    Z = {S - 1: z[segs[S - 1][0]].copy()}
    for s in range(S - 2, 0, -1):
        f = segs[s][0]
        Z[s] = z[f] + Q[f] @ Z[s + 1]
    for s in range(0, S - 1):
        a, b = segs[s]
        z[a:b] += np.einsum("kij,j->ki", Q[a:b], Z[s + 1])
    return z


@pytest.mark.parametrize("n,nwave", [(1, 1), (5, 1), (12, 2), (20, 3), (40, 5), (200, 16), (203, 16), (129, 16), (1000, 16)])
def test_partitioned_substitutions_equal_the_plain_recurrences(n, nwave):
    rng = np.random.default_rng(n * 31 + nwave)
    L = 0.35 * rng.normal(size=(n, 6, 6)) / np.sqrt(6)          # |L_k| < 1, as E^T Delta^-1 of a diagonally dominant chain
    A = rng.normal(size=(n, 6, 6))
    Dinv = np.einsum("kij,klj->kil", A, A) + 0.5 * np.eye(6)    # SPD blocks
    r = rng.normal(size=(n, 6))
    zp = solve_plain(L, Dinv, r)
    zq = solve_partitioned(L, Dinv, r, nwave)
    assert np.abs(zp - zq).max() <= 1e-12 * max(1.0, np.abs(zp).max())


def test_the_operator_is_symmetric():
    """M^-1 = (I + L)^-T D^-1 (I + L)^-1 is symmetric: u . (M^-1 v) == v . (M^-1 u) - what PCG needs of its preconditioner."""
    rng = np.random.default_rng(3)
    n = 57
    L = 0.3 * rng.normal(size=(n, 6, 6)) / np.sqrt(6)
    A = rng.normal(size=(n, 6, 6))
    Dinv = np.einsum("kij,klj->kil", A, A) + 0.5 * np.eye(6)
    u, v = rng.normal(size=(n, 6)), rng.normal(size=(n, 6))
    a = np.sum(u * solve_partitioned(L, Dinv, v, 7)); b = np.sum(v * solve_partitioned(L, Dinv, u, 7))
    assert abs(a - b) <= 1e-11 * max(abs(a), abs(b))
