#!/usr/bin/env python3
"""Extract the reference's only numeric golden artefact into a committed fixture.

Run in the BUILD container only (needs /root/reference):

    python tests/golden/make_golden.py

Reads  /root/reference/test/obj_1.mat   (saved `Dynamic_Solver` object, MAT v5 MCOS)
       /root/reference/test/obj_1.txt   (its constructor parameters, :1-17)
Writes tests/golden/obj_1.npz           (data only: params, knots, J_star, u_star
                                         indices, closed-loop trajectory)

The .mat stores the object as an opaque MCOS blob; scipy exposes the payload as
`__function_workspace__`, itself a MAT-5 stream whose first variable holds a
`FileWrapper__` cell array: cell 0 = metadata (property names), property p
(1-based in the name list) lives in cell p+1.

u_star is stored as the uint8 index into U_mesh = linspace(u_min,u_max,du)
(exactly recoverable: every stored value is one of the 100 mesh values) plus the
mesh itself, so the fixture stays small; J_star is stored in full (float64).
The trajectory is what test/test_coder.m:39-65 (get_optimal_path) computes from
X0=[2;1]: bilinear lookup of u_star(:,:,k) then x+ = A x + B u.
"""
import io
import struct
import sys
from pathlib import Path

import numpy as np
import scipy.io
from scipy.io.matlab._mio5 import MatFile5Reader

REF = Path("/root/reference/test/obj_1.mat")
OUT = Path(__file__).resolve().parent / "obj_1.npz"


def decode(path):
    m = scipy.io.loadmat(str(path))
    fw = m["__function_workspace__"].tobytes()
    hdr = b"MATLAB 5.0 MAT-file, repacked function workspace".ljust(116) + b"\x00" * 8 + fw[0:4]
    buf = io.BytesIO(hdr + fw[8:])
    r = MatFile5Reader(buf)
    r.initialize_read()
    buf.seek(128)
    h, _ = r.read_var_header()
    arr = r.read_var_array(h, process=False)
    cells = arr["MCOS"][0, 0]["arr"][0]
    meta = cells[0, 0].tobytes()
    _ver, nstr = struct.unpack("<II", meta[:8])
    names = [s.decode() for s in meta[40:].split(b"\x00")[:nstr]]
    props = {}
    for p, name in enumerate(names[:-1], start=1):  # last name is the class name
        props[name] = np.asarray(cells[p + 1, 0])
    return props


def matlab_linspace(a, b, n):
    """MATLAB's linspace: y(i) = a + (i*(b-a))/(n-1) (multiply first, then divide),
    end points forced exact.  numpy.linspace differs by <= 1 ulp on some entries."""
    y = a + (np.arange(n) * (b - a)) / (n - 1)
    y[0], y[-1] = a, b
    return y


def bilinear(k1, k2, V, x1, x2):
    def cw(k, q):
        i = int(np.clip(np.searchsorted(k, q, side="right") - 1, 0, len(k) - 2))
        return i, (q - k[i]) / (k[i + 1] - k[i])
    i, s = cw(k1, x1)
    j, t = cw(k2, x2)
    return ((1 - s) * (1 - t) * V[i, j] + s * (1 - t) * V[i + 1, j]
            + (1 - s) * t * V[i, j + 1] + s * t * V[i + 1, j + 1])


def main():
    if not REF.exists():
        sys.exit("reference fixture not found (this script runs in the build container only)")
    P = decode(REF)
    A, B, Q, R = P["A"], P["B"], P["Q"], float(P["R"].item())
    N, dx, du = int(P["N"].item()), int(P["dx"].item()), int(P["du"].item())
    x_min, x_max = float(P["x_min"].item()), float(P["x_max"].item())
    u_min, u_max = float(P["u_min"].item()), float(P["u_max"].item())
    J_star, u_star = P["J_star"], P["u_star"]
    X1, X2 = P["X1_mesh"], P["X2_mesh"]
    assert J_star.shape == (dx, dx, N) == u_star.shape
    k1, k2 = X1[:, 0].copy(), X2[0, :].copy()  # MATLAB's own linspace knots (ndgrid)
    assert np.array_equal(k1, matlab_linspace(x_min, x_max, dx)), "linspace restatement differs from the fixture"
    assert np.array_equal(X1, np.repeat(k1[:, None], dx, 1)) and np.array_equal(X2, np.repeat(k2[None, :], dx, 0))
    U_mesh = matlab_linspace(u_min, u_max, du)
    idx = np.abs(u_star[:, :, : N - 1, None] - U_mesh[None, None, None, :]).argmin(-1).astype(np.uint8)
    assert np.array_equal(U_mesh[idx], u_star[:, :, : N - 1]), "u_star values are not exact U_mesh members"
    assert not u_star[:, :, N - 1].any() and not J_star[:, :, N - 1].any()
    # closed-loop trajectory (test_coder.m:39-65)
    X = np.zeros((2, N))
    U = np.zeros(N)
    X[:, 0] = [2.0, 1.0]
    for k in range(N - 1):
        U[k] = bilinear(k1, k2, u_star[:, :, k], X[0, k], X[1, k])
        X[:, k + 1] = A @ X[:, k] + B[:, 0] * U[k]
    U[N - 1] = bilinear(k1, k2, u_star[:, :, N - 1], X[0, N - 1], X[1, N - 1])
    np.savez_compressed(
        OUT, A=A, B=B, Q=Q, R=R, N=N, dx=dx, du=du, x_min=x_min, x_max=x_max, u_min=u_min, u_max=u_max,
        knots1=k1, knots2=k2, U_mesh=U_mesh, J_star=J_star, u_star_idx=idx, traj_X=X, traj_U=U)
    print("wrote", OUT, OUT.stat().st_size, "bytes; u*(1..3) =", U[:3], "x(N) =", X[:, -1])


if __name__ == "__main__":
    main()
