"""Decode the reference's SRS fixture (halo2_verifier/params/kzg_bn254_8.srs, copied verbatim to
tests/golden/) with Python big ints.  Layout: k:u32 LE | g[n] | g_lagrange[n] |
g2 | s_g2, RawBytes = 4 x u64 LE Montgomery limbs per base-field element (R = 2^256)."""
P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
RINV = pow(1 << 256, -1, P)


def fq(b):
    return int.from_bytes(b, "little") * RINV % P


def le32(x):
    return int(x).to_bytes(32, "little")


def g1_xy(pt):
    """(x, y) ints -> 64-byte canonical x|y (None = identity = zeros)"""
    if pt is None:
        return bytes(64)
    return le32(pt[0]) + le32(pt[1])


class Srs:
    pass


def load_srs(path):
    data = open(path, "rb").read()
    s = Srs()
    s.raw = data
    s.k = int.from_bytes(data[:4], "little")
    n = 1 << s.k
    s.n = n
    off = 4

    def g1(o):
        return (fq(data[o:o + 32]), fq(data[o + 32:o + 64]))

    s.g = [g1(off + 64 * i) for i in range(n)]
    off += 64 * n
    s.g_lagrange = [g1(off + 64 * i) for i in range(n)]
    off += 64 * n

    def g2(o):
        return tuple(fq(data[o + 32 * j:o + 32 * j + 32]) for j in range(4))  # x.c0, x.c1, y.c0, y.c1

    s.g2 = g2(off)
    s.s_g2 = g2(off + 128)
    assert off + 256 == len(data)
    # verifier params in RawBytes form (poly/kzg/commitment.rs:142-152): k | g | g2 | s_g2
    s.params_raw = data[:4] + data[4:68] + data[off:off + 256]
    return s


def g2_bytes(q):
    return b"".join(le32(c) for c in q)


def omega(k):
    w = pow(7, (R_MOD - 1) >> 28, R_MOD)
    for _ in range(28 - k):
        w = w * w % R_MOD
    return w
