"""ORACLE (test infrastructure only; never imported by the product path).

numpy restatement of /root/reference/src/turbulence.jl (Float32, the reference's operation order).  No reference test
exercises these closures: parity is pinned only by the analytic properties in tests/test_turbulence.py (law of the
wall limits, pure shear / pure rotation, model constants) -- "parity unpinned" against Julia's libm otherwise."""
import numpy as np

f32 = np.float32
EPS = np.finfo(np.float32).eps


def von_Karman(yp, kappa=f32(0.41), C=f32(4.9)):
    """:11-16"""
    return np.minimum(np.log(np.maximum(yp, f32(1.0))) / kappa + C, yp)


def wall_function_rey(Rey, kappa=f32(0.41), C=f32(4.9), A=f32(19.0), beta=f32(0.075), betastar=f32(0.09), D=f32(4.2),
                      Aplus=f32(360.0), omega=f32(0.5), n_iter=20):
    """:27-70"""
    Rey = np.clip(np.abs(Rey), EPS, np.float32(np.inf))
    yp = np.sqrt(Rey)
    for _ in range(n_iter):
        up = von_Karman(yp, kappa, C)
        yp = omega * (Rey / up) + (f32(1.0) - omega) * yp
    up = Rey / yp
    mup = kappa * yp * (f32(1.0) - np.exp(-yp / A)) ** 2
    dudy = f32(1.0) / (f32(1.0) + mup)
    kp = np.minimum(yp ** 2 / (f32(6.0) * betastar / beta - f32(2.0)), D * np.exp(-yp / Aplus))
    return dict(yplus=yp, uplus=up, muplus=mup, kplus=kp, duplus_dyplus=dudy)


def wall_function(y, u, nu, betastar=f32(0.09), **kw):
    """:72-100"""
    nt = wall_function_rey(u * y / nu, betastar=betastar, **kw)
    utau = u / nt["uplus"]
    nut = nt["muplus"] * nu
    k = nt["kplus"] * utau ** 2
    om = k / nut
    return dict(utau=utau, nut=nut, k=k, omega=om, epsilon=betastar * om * k,
                du_dn=nt["duplus_dyplus"] * utau ** 2 / nu)


def shear_rate(g):
    """:110-124"""
    s = np.zeros_like(g[0][0])
    n = len(g)
    for i in range(n):
        for j in range(n):
            s = s + ((g[i][j] + g[j][i]) / f32(2)) ** 2
    return np.sqrt(f32(2) * s)


def Smagorinsky_nuSGS(Delta, S, Cs=f32(0.17)):
    """:135-138"""
    return (Cs * Delta) ** 2 * S


def standard_k_epsilon(k, e, S, Cmu=f32(0.09), sk=f32(1.0), se=f32(1.3), C1=f32(1.44), C2=f32(1.92)):
    """:176-196"""
    nut = Cmu * k ** 2 / e
    Pk = nut * S ** 2
    return dict(nuk=nut / sk, nueps=nut / se, Sk=Pk - e, Seps=C1 * Pk * e / k - C2 * e ** 2 / k, nut=nut)


def Wray_Agarwal(R, S, gR, gS, sigmaR=f32(0.72), C1=f32(0.0829), kappa=f32(0.41)):
    """:222-241"""
    C2 = sigmaR + C1 / kappa ** 2
    dot = gR[:, 0] * gS[:, 0]
    for d in range(1, gR.shape[1]):
        dot = dot + gR[:, d] * gS[:, d]
    src = C1 * R * S + C2 * dot * (R / (S + EPS))
    return dict(nut=R, nuR=R * sigmaR, S=np.minimum(src, f32(10.0) * R))


def Ducros_sensor(g):
    """:252-282"""
    nd = len(g)
    div = np.zeros_like(g[0][0])
    for i in range(nd):
        div = div + g[i][i]
    div2 = div ** 2
    if nd == 2:
        curl2 = (g[1][0] - g[0][1]) ** 2
    else:
        curl2 = (g[2][1] - g[1][2]) ** 2 + (g[0][2] - g[2][0]) ** 2 + (g[1][0] - g[0][1]) ** 2
    return (div2 + EPS) / (div2 + curl2 + EPS)


def WALE_nuSGS(Delta, g, Cw=f32(0.325)):
    """:291-337"""
    nd = 3
    g2 = [[None] * nd for _ in range(nd)]
    for i in range(nd):
        for j in range(nd):
            s = np.zeros_like(g[0][0])
            for k in range(nd):
                s = s + g[i][k] * g[k][j]
            g2[i][j] = s
    SS = np.zeros_like(g[0][0])
    SdSd = np.zeros_like(g[0][0])
    for i in range(nd):
        for j in range(nd):
            SS = SS + ((g[i][j] + g[j][i]) / f32(2)) ** 2
            SdSd = SdSd + ((g2[i][j] + g2[j][i]) / f32(2) - g2[i][j] * f32((1.0 if i == j else 0.0) / 3)) ** 2
    return Cw * Delta ** 2 * SdSd ** f32(1.5) / (SS ** f32(2.5) + SdSd ** f32(1.25) + EPS)
