import numpy as np
from scipy.special import erfc, erf
from scipy.optimize import least_squares
A=5.75
a=np.linspace(0,A,20001)
qs=-np.log2(erfc(a/np.sqrt(2)))
def model(c,a):
    q=np.zeros_like(a)
    for k in reversed(range(len(c))): q=q*a+c[k]
    return q*a   # q(0)=0 exactly -> E(0)=1
for deg in (5,6,7,8):
    # init LSQ in q-space weighted by E
    E=erfc(a/np.sqrt(2))
    V=np.stack([a**(k+1) for k in range(deg)],1)
    w=E+1e-3
    c0=np.linalg.lstsq(V*w[:,None],qs*w,rcond=None)[0]
    f=lambda c: (np.exp2(-model(c,a))-E)
    r=least_squares(f,c0,xtol=1e-15,ftol=1e-15,gtol=1e-15,max_nfev=2000)
    c=r.x
    # minimax-ish refinement via iterative reweighting
    wts=np.ones_like(a)
    for it in range(60):
        res=np.exp2(-model(c,a))-E
        wts*= (1+ 2*np.abs(res)/np.abs(res).max()); wts/=wts.mean()
        r=least_squares(lambda cc:(np.exp2(-model(cc,a))-E)*wts,c,xtol=1e-15,ftol=1e-15,gtol=1e-15,max_nfev=400)
        c=r.x
    c32=c.astype(np.float32)
    a32=a.astype(np.float32)
    q=np.zeros_like(a32)
    for k in reversed(range(deg)): q=(q*a32+c32[k]).astype(np.float32)
    q=(q*a32)
    E32=np.exp2(-q.astype(np.float64))
    print(deg, "maxerr E", np.abs(E32-E).max(), "gelu abs err", np.abs(0.5*a*(E32-E)).max(), "beyond A: E(A)=",E32[-1])
    print("   ", ", ".join(f"{v:.9e}f" for v in c32))


# ---- grad mode: gelu'(x) - 1/2 = t P(t^2), t = clamp(x, -4, 4) / 4 (common.h: gelu_fast_grad)
def gelu_grad(x): return 0.5 * (1 + erf(x / np.sqrt(2))) + x * np.exp(-x * x / 2) / np.sqrt(2 * np.pi)
A, nc = 4.0, 8
x = np.linspace(0, A, 80001); y = gelu_grad(x) - 0.5; t = x / A
V = np.stack([t ** (2 * k + 1) for k in range(nc)], 1)
w = np.ones_like(x)
for it in range(200):
    c = np.linalg.lstsq(V * w[:, None], y * w, rcond=None)[0]
    r = np.abs(V @ c - y); w *= (1 + 3 * r / r.max()); w /= w.mean()
c32 = c.astype(np.float32)
xx = np.linspace(-8, 8, 400001)
tt = (np.clip(xx, -A, A).astype(np.float32) * np.float32(1 / A)).astype(np.float32); t2 = (tt * tt).astype(np.float32)
p = np.full_like(tt, c32[nc - 1])
for k in reversed(range(nc - 1)): p = (p * t2 + c32[k]).astype(np.float32)
g = (p * tt + np.float32(0.5)).astype(np.float64)
print("gelu' fit: max abs err over [-8, 8]", np.abs(g - gelu_grad(xx)).max())
print("   ", ", ".join(f"{v:.9e}f" for v in c32))
