#!/usr/bin/env python3
"""Print the 0.128 fixed-point R_d constants used by the sample-table builders.

R_d (Roberts) low-discrepancy sequence: phi_d is the positive root of x^(d+1) = x + 1 and
alpha_j = phi_d^-j, j = 1..d.  d=1: 1/phi (golden ratio); d=2: 1/rho, 1/rho^2 (plastic number).
The reference gets these from the un-vendored crate quasi-rd @ ce117035 (Cargo.lock:410-417,
call sites src/sampler.rs:23-29); this script restates the published construction with exact
integer arithmetic so that both the oracle and the product can embed identical constants.
"""
from fractions import Fraction

BITS = 128
GUARD = 64


def root(d):
    # Newton on f(x) = x^(d+1) - x - 1 in exact rationals truncated to BITS+GUARD bits
    scale = 1 << (BITS + GUARD)
    x = Fraction(3, 2)
    for _ in range(200):
        f = x ** (d + 1) - x - 1
        fp = (d + 1) * x ** d - 1
        x = x - f / fp
        x = Fraction(int(x * scale), scale)
    return x


def fixed(fr):
    v = int(fr * (1 << BITS))
    return v >> 64, v & ((1 << 64) - 1)


for d in (1, 2):
    phi = root(d)
    for j in range(1, d + 1):
        a = 1 / phi ** j
        hi, lo = fixed(a)
        print(f"d={d} j={j} alpha={float(a):.20f} hi=0x{hi:016x} lo=0x{lo:016x}")
