#!/usr/bin/env python3
"""Generates tests/golden/pcg_kat.json: known answers of the PCG hash chain the reference's RNG is built on
(mixins/random/hash/pcg.glsl:3-7, squashlinear.glsl:7-9, distribution/uniformdivision.glsl:3-6).

Independent derivation in arbitrary-precision Python integers (no C, no numpy): the GLSL source uses 32-bit
unsigned arithmetic, so every product/sum is reduced mod 2^32.  The first five entries reproduce the values the
structural survey captured.  The uniform is float(state)/float(~0u): float(~0u) rounds to 2^32 in
binary32, so the quotient is RN_binary32(state) * 2^-32, evaluated here with exact rationals.
"""
import json
import os
from fractions import Fraction

M = 1 << 32


def pcg(x):
    x = (x * 747796405 + 2891336453) % M
    x = ((((x >> ((x >> 28) + 4)) ^ x) % M) * 277803737) % M
    return ((x >> 22) ^ x) % M


def hash3(x, y, z):
    return pcg((19 * x + 47 * y + 101 * z + 131) % M)


def rn_binary32(k):
    """round-to-nearest-even of the integer k to a 24-bit significand, as an exact integer"""
    if k == 0:
        return 0
    nbits = k.bit_length()
    if nbits <= 24:
        return k
    shift = nbits - 24
    q, r = k >> shift, k & ((1 << shift) - 1)
    half = 1 << (shift - 1)
    if r > half or (r == half and (q & 1)):
        q += 1
    return q << shift


def uniform_bits(state):
    """binary32 bit pattern of float(state) * 2^-32"""
    k = rn_binary32(state)
    if k == 0:
        return 0
    v = Fraction(k, M)                 # exact, representable
    e = 0
    while v < 1:
        v *= 2; e -= 1
    while v >= 2:
        v /= 2; e += 1
    mant = int((v - 1) * (1 << 23))
    return ((e + 127) << 23) | mant


def main():
    singles = [0, 1, 0xffffffff, 12345, 0x3f000000, 0xdeadbeef, 0x80000000, 0x7fffffff, 2891336453, 747796405]
    out = {"generator": "tests/golden/make_pcg_kat.py",
           "pcg": [[x, pcg(x)] for x in singles],
           "hash3": [], "uniform_chain": []}
    triples = [(0x3f000000,) * 3, (0, 0, 0), (1, 2, 3), (0x3f800000, 0x3f000000, 0x3e99999a), (0xffffffff, 0xffffffff, 0xffffffff)]
    for t in triples:
        out["hash3"].append([list(t), hash3(*t)])
    for t in triples[:3]:
        s = hash3(*t)
        chain = []
        for _ in range(8):
            s = pcg(s)
            chain.append([s, uniform_bits(s)])
        out["uniform_chain"].append({"seed_triple": list(t), "chain": chain})
    # states that round up to 2^32 (uniform == 1.0) and the smallest non-zero
    out["uniform_edge"] = [[s, uniform_bits(s)] for s in (0, 1, 0xffffff7f, 0xffffff80, 0xffffffff, 0x00ffffff, 0x01000001)]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pcg_kat.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
