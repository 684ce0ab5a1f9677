"""Coefficients of rte::exp_nonpos (csrc/fastmath.h): exp(r) = 1 + r + r^2 q(r) on |r| <= ln2 / 2 (+ margin), q of degree DEG - 2
minimising the relative error of the sum (Remez exchange in 60-digit arithmetic, weight r^2 exp(-r)); then the error of the
double-precision Horner evaluation with the coefficients rounded to double, against the Taylor polynomial of degree 13 it replaces.
usage: python tools/exp_minimax.py [DEG=11]"""
import sys
import mpmath as mp
import numpy as np

mp.mp.dps = 60
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 11
NQ = DEG - 1           # number of coefficients of q (degree DEG - 2)
A = mp.log(2) / 2 * mp.mpf("1.0001")


def Q(r):
    return (mp.exp(r) - 1 - r) / (r * r) if abs(r) > mp.mpf("1e-12") else mp.mpf(1) / 2 + r / 6 + r * r / 24


def w(r):
    return r * r * mp.exp(-r)


def poly(c, r):
    p = mp.mpf(0)
    for ck in reversed(c):
        p = p * r + ck
    return p


# initial reference: Chebyshev extrema, nudged off zero
n = NQ + 1
xs = [-A * mp.cos(mp.pi * i / (n - 1)) for i in range(n)]
xs = [x if abs(x) > A / 50 else A / 50 for x in xs]
grid = [-A + 2 * A * mp.mpf(i) / 6000 for i in range(6001)]
for it in range(30):
    M = mp.matrix(n, n); b = mp.matrix(n, 1)
    for i, x in enumerate(xs):
        for k in range(NQ):
            M[i, k] = x ** k
        M[i, NQ] = -((-1) ** i) / w(x)
        b[i] = Q(x)
    sol = mp.lu_solve(M, b)
    c = [sol[k] for k in range(NQ)]; E = sol[NQ]
    e = [w(x) * (poly(c, x) - Q(x)) for x in grid]
    # local extrema of |e| between sign changes; the double zero at r = 0 splits one lobe in two: merge lobes of equal sign
    lobes = []; start = 0
    for i in range(1, len(grid) + 1):
        if i == len(grid) or (e[i] > 0) != (e[start] > 0):
            j = max(range(start, i), key=lambda t: abs(e[t])); lobes.append(j); start = i
    merged = []
    for j in lobes:
        if merged and (e[merged[-1]] > 0) == (e[j] > 0):
            if abs(e[j]) > abs(e[merged[-1]]): merged[-1] = j
        else:
            merged.append(j)
    while len(merged) > n:   # drop the smaller end lobe
        merged.pop(0 if abs(e[merged[0]]) < abs(e[merged[-1]]) else -1)
    emax = max(abs(v) for v in e)
    print("iter %d: levelled error %.3e, max weighted error %.3e, %d lobes" % (it, float(abs(E)), float(emax), len(merged)))
    if len(merged) < n: print("  (fewer lobes than reference points: keeping the reference)"); break
    new = [grid[j] for j in merged]
    if max(abs(a_ - b_) for a_, b_ in zip(new, xs)) < A / 3000 and emax < abs(E) * mp.mpf("1.02"): xs = new; break
    xs = new
cd = [float(ck) for ck in c]
print("q coefficients (c2 ... c%d), as doubles:" % DEG)
for k, v in enumerate(cd): print("  c%-2d = %s   (%.17g; Taylor 1/%d! = %.17g)" % (k + 2, v.hex(), v, k + 2, 1.0 / float(mp.factorial(k + 2))))

# double-precision Horner, as the kernel evaluates it (fma steps), against exp in 60 digits
rng = np.random.default_rng(1)
r = np.concatenate([rng.uniform(-float(A), float(A), 200000), np.linspace(-float(A), float(A), 20001)])
def fma(a, b, c_):  # exact product + one rounding, through mpmath (slow but only used on a sample)
    return float(mp.mpf(a) * mp.mpf(b) + mp.mpf(c_))
def horner(coefs, x):
    p = coefs[-1]
    for ck in reversed(coefs[:-1]): p = fma(p, x, ck)
    return p
tay = [1.0, 1.0] + [1.0 / float(mp.factorial(k)) for k in range(2, 14)]
new = [1.0, 1.0] + cd
worst = {"taylor13": 0.0, "minimax%d" % DEG: 0.0}
mp.mp.dps = 40
for x in r[::20]:
    ex = mp.exp(mp.mpf(float(x)))
    for name, cf in (("taylor13", tay), ("minimax%d" % DEG, new)):
        v = horner(cf, float(x))
        worst[name] = max(worst[name], float(abs(mp.mpf(v) - ex) / ex))
print("worst relative error of the double-precision evaluation over %d arguments: %s  (2^-53 = %.3e)" % (len(r[::20]), worst, 2.0 ** -53))
