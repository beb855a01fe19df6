#!/usr/bin/env python3
"""Generate the constants of include/peakseg_detmath.h with mpmath.

Run:  python tools/gen_detmath.py > /tmp/consts.txt   (paste into the header)
The header is committed; this script documents where every constant came from.
"""
import mpmath as mp
mp.mp.prec = 400

def d(x):           # round to nearest double, print as C hex float
    return float(x).hex()

ln2 = mp.log(2)
# ln2_hi: ln2 truncated to 32 significant bits so kd*ln2_hi is exact for |kd| < 2^21
hi = mp.floor(ln2 * 2**32) / 2**32
lo = ln2 - hi
print("LN2_HI", d(hi), "LN2_LO", d(lo), "INV_LN2", d(1/ln2))
print("exp taylor c2..c13:")
for n in range(2, 14):
    print("  C%d" % n, d(mp.mpf(1)/mp.factorial(n)))

# log: log(1+f) = 2s + s*z*P(z), s=f/(2+f), z=s^2, z in [0, zmax]
smax = (mp.sqrt(2)-1)/(mp.sqrt(2)+1)
zmax = smax**2 * mp.mpf('1.02')
def g(z):
    if z == 0:
        return mp.mpf(2)/3
    s = mp.sqrt(z)
    return (mp.log((1+s)/(1-s)) - 2*s)/(s*z)
DEG = 7   # 8 coefficients
# Chebyshev-node interpolation on [0, zmax] (near-minimax)
nodes = [ (zmax/2)*(1+mp.cos(mp.pi*(2*i+1)/(2*(DEG+1)))) for i in range(DEG+1)]
A = mp.matrix(DEG+1, DEG+1); b = mp.matrix(DEG+1, 1)
for i, z in enumerate(nodes):
    for j in range(DEG+1):
        A[i, j] = z**j
    b[i] = g(z)
c = mp.lu_solve(A, b)
print("log P(z) coefficients L0..L%d:" % DEG)
for j in range(DEG+1):
    print("  L%d" % j, d(c[j]))
# report approximation error of rounded coefficients
cd = [mp.mpf(float(c[j])) for j in range(DEG+1)]
worst = 0
for i in range(2001):
    z = zmax*i/2000
    p = sum(cd[j]*z**j for j in range(DEG+1))
    s = mp.sqrt(z)
    if z == 0: continue
    approx = 2*s + s*z*p
    exact = mp.log((1+s)/(1-s))
    worst = max(worst, abs((approx-exact)/exact))
print("log poly rel err (in 2^-53 units):", mp.nstr(worst*2**53, 5))
print("SQRT2", d(mp.sqrt(2)))
