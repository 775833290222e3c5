#!/usr/bin/env python3
"""Fit the fp32 polynomial coefficients used by the Philox->normal mapping spec
(DESIGN.md "Noise spec").  Output: hex-float literals to paste into BOTH
ccv_mppi_path_tracker_amd/csrc/noise_spec.h (product) and oracle/philox_normal.h
(independent checker).  Least-squares on Chebyshev nodes in fp64, coefficients
rounded to fp32, accuracy of the fp32 Horner evaluation reported.
"""
import numpy as np

def cheb_nodes(a, b, n):
    k = np.arange(n)
    return 0.5*(a+b) + 0.5*(b-a)*np.cos(np.pi*(2*k+1)/(2*n))

def fit(fx, a, b, deg, n=4000, rel=True):
    x = cheb_nodes(a, b, n)
    y = fx(x)
    V = np.vander(x, deg+1, increasing=True)
    w = 1.0/np.abs(y) if rel else np.ones_like(y)
    c, *_ = np.linalg.lstsq(V*w[:, None], y*w, rcond=None)
    return c.astype(np.float32)

def horner32(c, x):
    x = x.astype(np.float32)
    acc = np.full_like(x, c[-1], dtype=np.float32)
    for ci in c[-2::-1]:
        # emulate fmaf with float64 product+add then round (exact enough: one rounding)
        acc = (acc.astype(np.float64)*x.astype(np.float64) + np.float64(ci)).astype(np.float32)
    return acc

s2 = np.sqrt(0.5)
# log2(1+t)/t on [sqrt(.5)-1, sqrt(2)-1]
def fq(t):
    t = np.where(np.abs(t) < 1e-12, 1e-12, t)
    return np.log2(1.0+t)/t
for deg in (7, 8, 9):
    cq = fit(fq, s2-1, np.sqrt(2)-1, deg)
    t = np.linspace(s2-1, np.sqrt(2)-1, 200001)
    t = t[np.abs(t) > 1e-9]
    approx = horner32(cq, t).astype(np.float64)
    err = np.max(np.abs(approx - fq(t.astype(np.float32).astype(np.float64))) / np.abs(fq(t)))
    print("log2 deg", deg, "max rel err", err)
    if deg == 8:
        print("  Q:", ", ".join(float(c).hex() for c in cq))
        print("  Q dec:", ", ".join(repr(float(c)) for c in cq))

# sin(a) = a + a*w*S(w), w=a^2, a in [-pi/4, pi/4]
def fs(w):
    a = np.sqrt(np.maximum(w, 1e-30))
    return (np.sin(a)/a - 1.0)/np.maximum(w, 1e-30)
def fc(w):
    a = np.sqrt(np.maximum(w, 1e-30))
    return (np.cos(a) - 1.0)/np.maximum(w, 1e-30)
W = (np.pi/4)**2
for deg in (2, 3):
    cs = fit(fs, 1e-6, W*1.0001, deg)
    cc = fit(fc, 1e-6, W*1.0001, deg)
    a = np.linspace(-np.pi/4, np.pi/4, 200001).astype(np.float32)
    w = (a.astype(np.float64)**2).astype(np.float32)
    S = horner32(cs, w); C = horner32(cc, w)
    aw = (a.astype(np.float64)*w.astype(np.float64)).astype(np.float32)
    sin32 = (aw.astype(np.float64)*S.astype(np.float64) + a.astype(np.float64)).astype(np.float32)
    cos32 = (w.astype(np.float64)*C.astype(np.float64) + 1.0).astype(np.float32)
    es = np.max(np.abs(sin32.astype(np.float64) - np.sin(a.astype(np.float64))))
    ec = np.max(np.abs(cos32.astype(np.float64) - np.cos(a.astype(np.float64))))
    print("sincos deg", deg, "max abs err sin", es, "cos", ec)
    if deg == 3:
        print("  S:", ", ".join(float(c).hex() for c in cs))
        print("  C:", ", ".join(float(c).hex() for c in cc))
        print("  S dec:", ", ".join(repr(float(c)) for c in cs))
        print("  C dec:", ", ".join(repr(float(c)) for c in cc))
print("pi/2 as f32:", float(np.float32(np.pi/2)).hex(), "2ln2 f32:", float(np.float32(2*np.log(2))).hex())
