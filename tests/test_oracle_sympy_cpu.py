"""Known answers for the barrier rows, derived symbolically (the reference derives the same quantities in
cbf/symb_lie_deriv.ipynb): h = (ex^2+ey^2)^2 + (ez/c)^4 - Ds^4 and its Lie derivatives along the hover linearisations
(model/linear_omega.py:46-53, model/linear_yank_omega.py:45-51), against the closed forms the oracle (and the HIP kernels,
which are tested against the oracle) use.  Independent of the reference-minted cbf_rows_o{2,3}.npz fixtures."""
import numpy as np
import pytest
import sympy as sp

from oracle import np_oracle as O

M, G = O.CF2P.M, O.CF2P.G


def _model(order):
    """(A, B, position rows) of the error dynamics d' = A d + B du; d = (x_i - xdes_i) - (x_j - xdes_j)."""
    if order == 2:      # [r, p, y, vx, vy, vz, x, y, z]
        A = sp.zeros(9, 9); B = sp.zeros(9, 4)
        A[3, 1], A[4, 0] = G, -G
        for k in range(3):
            A[6 + k, 3 + k] = 1
            B[k, 1 + k] = 1
        B[5, 0] = sp.Rational(1) / M
        return A, B, (6, 7, 8)
    A = sp.zeros(10, 10); B = sp.zeros(10, 4)     # [r, p, y, F, vx, vy, vz, x, y, z]
    A[4, 1], A[5, 0], A[6, 3] = G, -G, sp.Rational(1) / M
    for k in range(3):
        A[7 + k, 4 + k] = 1
        B[k, 1 + k] = 1
    B[3, 0] = 1
    return A, B, (7, 8, 9)


def _lie(order):
    """h and L_f^k h (k = 1..order), L_g L_f^{order-1} h as sympy expressions of (e, d, c, Ds).  e is the actual position
    difference, d the error-state difference (cbf.py:135-178 evaluates h on e and A x_hat on d); e' = position rows of A d."""
    A, B, prow = _model(order)
    n = A.shape[0]
    e = sp.Matrix(sp.symbols("ex ey ez"))
    d = sp.Matrix(sp.symbols(f"d0:{n}"))
    c, Ds = sp.symbols("c Ds")
    h = (e[0] ** 2 + e[1] ** 2) ** 2 + (e[2] / c) ** 4 - Ds ** 4
    Ad = A * d
    edot = sp.Matrix([Ad[r] for r in prow])

    def Lf(phi):
        return (sp.Matrix([phi]).jacobian(e) * edot + sp.Matrix([phi]).jacobian(d) * Ad)[0]

    out = [h]
    for _ in range(order):
        out.append(sp.expand(Lf(out[-1])))
    Lg = sp.Matrix([out[order - 1]]).jacobian(d) * B      # 1 x 4
    return e, d, c, Ds, out, Lg, A


@pytest.mark.parametrize("order", [2, 3])
def test_pair_terms_match_symbolic_lie_derivatives(order):
    e, d, c, Ds, L, Lg, A = _lie(order)
    n = len(d)
    args = list(e) + list(d) + [c, Ds]
    fL = [sp.lambdify(args, x, "numpy") for x in L]
    fLg = sp.lambdify(args, Lg, "numpy")
    rng = np.random.default_rng(order)
    K = np.array([5.28, 4.6]) if order == 2 else np.array([60.48, 47.76, 12.2])
    for trial in range(200):
        xi, xj, xid, xjd = (rng.normal(size=n) * np.r_[0.3 * np.ones(n - 3), 1.0, 1.0, 0.5] for _ in range(4))
        zs, ds = rng.uniform(0.5, 2.0), rng.uniform(0.1, 0.5)
        ev = (xi - xj)[-3:]
        dv = (xi - xid) - (xj - xjd)
        vals = [float(f(*ev, *dv, zs, ds)) for f in fL]
        lg = np.asarray(fLg(*ev, *dv, zs, ds), dtype=np.float64).reshape(4)
        # K with the second-derivative gain removed for order 3: that term is the reference's slot quirk, checked below
        Kt = K.copy()
        if order == 3:
            Kt[2] = 0.0
        hij, Lg_o = O._cbf_pair_terms(xi, xj, xid, xjd, order, ds, zs, Kt, M, G)
        want = Kt[0] * vals[0] + Kt[1] * vals[1] + vals[order]
        scale = max(1.0, abs(want))
        assert abs(float(hij) - want) < 1e-9 * scale, (trial, float(hij), want)
        np.testing.assert_allclose(np.asarray(Lg_o, dtype=np.float64), lg, rtol=1e-10, atol=1e-12)


def test_order3_second_derivative_term_is_the_reference_slot_quirk():
    """cbf/cbf.py:158-169 evaluates the i == 2 term of custom_hdots with hard-coded slots 6,7,8 -- positions of the 9-state,
    but (vz, x, y) of the 10-state: grad h . (A A d)[6:9] + (A d)[6:9]^T H (A d)[6:9].  The oracle must reproduce that, and it is
    NOT the true second Lie derivative."""
    e, d, c, Ds, L, _, A = _lie(3)
    h = L[0]
    grad = sp.Matrix([h]).jacobian(e)
    H = sp.hessian(h, list(e))
    Ad, AAd = A * d, A * A * d
    w = sp.Matrix([Ad[6], Ad[7], Ad[8]])
    quirk = (grad * sp.Matrix([AAd[6], AAd[7], AAd[8]]))[0] + (w.T * H * w)[0]
    args = list(e) + list(d) + [c, Ds]
    fq, ftrue = sp.lambdify(args, quirk, "numpy"), sp.lambdify(args, L[2], "numpy")
    rng = np.random.default_rng(5)
    differs = 0
    for trial in range(100):
        xi, xj, xid, xjd = (rng.normal(size=10) * 0.5 for _ in range(4))
        zs, ds = rng.uniform(0.5, 2.0), 0.2
        ev, dv = (xi - xj)[-3:], (xi - xid) - (xj - xjd)
        K0 = np.array([0.0, 0.0, 0.0])
        K1 = np.array([0.0, 0.0, 1.0])
        a, _ = O._cbf_pair_terms(xi, xj, xid, xjd, 3, ds, zs, K0, M, G)
        b, _ = O._cbf_pair_terms(xi, xj, xid, xjd, 3, ds, zs, K1, M, G)
        got, want = float(b) - float(a), float(fq(*ev, *dv, zs, ds))
        assert abs(got - want) < 1e-9 * max(1.0, abs(want))
        differs += abs(want - float(ftrue(*ev, *dv, zs, ds))) > 1e-6
    assert differs > 90
