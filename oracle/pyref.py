"""ORACLE — TEST INFRASTRUCTURE ONLY.  Second, independent CPU restatement of the reference verifier in plain
Python big integers (small cases only).  It exists so that parity does not rest on one restatement: the C++ oracle
(oracle/verifier.cpp) and this file are written separately — different language, different field representation
(Python ints vs Montgomery limbs), a different pairing construction (Fq12 as Fq[w]/(w^12 - 18 w^6 + 82) with affine
line functions and a plain (p^12-1)/r exponentiation vs a 2-3-2 tower with projective steps) — and must agree on
every challenge, every Guard scalar/base and accept/reject.  tests/golden/generate.py uses it to produce the
committed golden vectors; nothing in the product or on the GPU box imports it.

Follows, with the same citations as oracle/verifier.hpp:
  transcript/mod.rs:104-232,484-515   lib.rs:33-425   plonk/{vk,permutation,lookup,shuffle,vanishing}.rs
  poly/domain.rs:172-212   poly/kzg/multiopen/shplonk.rs:58-267   poly/kzg/msm.rs:185-203   arithmetic.rs:137-206
"""
import hashlib

P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
S = 28
ROOT_OF_UNITY = pow(7, (R - 1) >> S, R)
DELTA = pow(7, 1 << S, R)
MONT_INV_Q = pow(1 << 256, -1, P)
MONT_INV_R = pow(1 << 256, -1, R)


# ------------------------------------------------------------------ G1 (affine, None = identity)
def g1_add(a, b):
    if a is None: return b
    if b is None: return a
    (x1, y1), (x2, y2) = a, b
    if x1 == x2:
        if (y1 + y2) % P == 0: return None
        m = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        m = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (m * m - x1 - x2) % P
    return (x3, (m * (x1 - x3) - y1) % P)


def g1_mul(k, pt):
    acc = None
    for bit in bin(k % R)[2:]:
        acc = g1_add(acc, acc)
        if bit == "1": acc = g1_add(acc, pt)
    return acc


def g1_decompress(b):
    """halo2curves >= 0.4 layout: byte 31 bit 7 identity, bit 6 sign = lsb of y."""
    v = int.from_bytes(b, "little")
    is_inf, sign = (v >> 255) & 1, (v >> 254) & 1
    x = v & ((1 << 254) - 1)
    if x >= P: raise ValueError("invalid point encoding in proof")
    if is_inf:
        if x or sign: raise ValueError("invalid point encoding in proof")
        return None
    rhs = (x * x * x + 3) % P
    y = pow(rhs, (P + 1) // 4, P)
    if y * y % P != rhs: raise ValueError("invalid point encoding in proof")
    if (y & 1) != sign: y = P - y
    return (x, y)


def g1_xy(pt):
    return bytes(64) if pt is None else pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little")


# ------------------------------------------------------------------ Fq12 = Fq[w] / (w^12 - 18 w^6 + 82)
def f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for d in range(22, 11, -1):  # w^12 = 18 w^6 - 82
        c = t[d]
        if c:
            t[d - 6] += 18 * c
            t[d - 12] -= 82 * c
    return [x % P for x in t[:12]]


def f12_one(): return [1] + [0] * 11
def f12_add(a, b): return [(x + y) % P for x, y in zip(a, b)]
def f12_sub(a, b): return [(x - y) % P for x, y in zip(a, b)]
def f12_scalar(a, k): return [x * k % P for x in a]


def _poly_deg(p):
    d = len(p) - 1
    while d and p[d] == 0: d -= 1
    return d


def f12_inv(a):
    """extended Euclid in Fq[w]"""
    mod = [82, 0, 0, 0, 0, 0, (-18) % P, 0, 0, 0, 0, 0, 1]
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], mod
    while _poly_deg(low):
        dl, dh = _poly_deg(low), _poly_deg(high)
        r = [0] * 13
        tmp = list(high)
        inv_lead = pow(low[dl], -1, P)
        for i in range(dh - dl, -1, -1):
            q = tmp[dl + i] * inv_lead % P
            r[i] = q
            for c in range(dl + 1):
                tmp[c + i] = (tmp[c + i] - q * low[c]) % P
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                new[i + j] = (new[i + j] - low[i] * r[j]) % P
        lm, low, hm, high = nm, new, lm, low
    return f12_scalar(lm[:12], pow(low[0], -1, P))


def f12_pow(a, e):
    r = f12_one()
    for bit in bin(e)[2:]:
        r = f12_mul(r, r)
        if bit == "1": r = f12_mul(r, a)
    return r


W2 = [0, 0, 1] + [0] * 9
W3 = [0, 0, 0, 1] + [0] * 8


def twist(q):
    """G2 point over Fq2 = (c0 + c1 u) -> curve point over Fq12 (u = w^6 - 9)"""
    (x0, x1), (y0, y1) = q
    nx = [(x0 - 9 * x1) % P] + [0] * 5 + [x1] + [0] * 5
    ny = [(y0 - 9 * y1) % P] + [0] * 5 + [y1] + [0] * 5
    return (f12_mul(nx, W2), f12_mul(ny, W3))


def cast_g1(pt): return ([pt[0]] + [0] * 11, [pt[1]] + [0] * 11)


def ec12_double(a):
    x, y = a
    m = f12_mul(f12_scalar(f12_mul(x, x), 3), f12_inv(f12_scalar(y, 2)))
    nx = f12_sub(f12_mul(m, m), f12_scalar(x, 2))
    return (nx, f12_sub(f12_mul(m, f12_sub(x, nx)), y))


def ec12_add(a, b):
    (x1, y1), (x2, y2) = a, b
    if x1 == x2: return ec12_double(a) if y1 == y2 else None
    m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    nx = f12_sub(f12_sub(f12_mul(m, m), x1), x2)
    return (nx, f12_sub(f12_mul(m, f12_sub(x1, nx)), y1))


def linefunc(p1, p2, t):
    (x1, y1), (x2, y2), (xt, yt) = p1, p2, t
    if x1 != x2: m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    elif y1 == y2: m = f12_mul(f12_scalar(f12_mul(x1, x1), 3), f12_inv(f12_scalar(y1, 2)))
    else: return f12_sub(xt, x1)
    return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))


ATE_LOOP_COUNT = 29793968203157093288


def miller_loop(q2, p1):
    """un-exponentiated ate Miller loop value for G2 point q2 (Fq2 coords) and G1 point p1"""
    if p1 is None or q2 is None: return f12_one()
    Q, Pt = twist(q2), cast_g1(p1)
    Rr, f = Q, f12_one()
    for i in range(63, -1, -1):
        f = f12_mul(f12_mul(f, f), linefunc(Rr, Rr, Pt))
        Rr = ec12_double(Rr)
        if ATE_LOOP_COUNT & (1 << i):
            f = f12_mul(f, linefunc(Rr, Q, Pt))
            Rr = ec12_add(Rr, Q)
    Q1 = (f12_pow(Q[0], P), f12_pow(Q[1], P))
    nQ2 = (f12_pow(Q1[0], P), f12_scalar(f12_pow(Q1[1], P), P - 1))
    f = f12_mul(f, linefunc(Rr, Q1, Pt))
    Rr = ec12_add(Rr, Q1)
    return f12_mul(f, linefunc(Rr, nQ2, Pt))


def pairing_check(left, right, s_g2, g2):
    """DualMSM::check: e(left, s_g2) * e(right, -g2) == 1   (poly/kzg/msm.rs:185-203)"""
    ng2 = (g2[0], ((-g2[1][0]) % P, (-g2[1][1]) % P))
    f = f12_mul(miller_loop(s_g2, left), miller_loop(ng2, right))
    return f12_pow(f, (P ** 12 - 1) // R) == f12_one()


# ------------------------------------------------------------------ byte readers (reference formats, RawBytes)
class Reader:
    def __init__(self, data): self.d, self.pos = data, 0
    def take(self, n):
        if self.pos + n > len(self.d): raise ValueError("failed to fill whole buffer")
        b = self.d[self.pos:self.pos + n]; self.pos += n; return b
    def u8(self): return self.take(1)[0]
    def u16(self): return int.from_bytes(self.take(2), "big")
    def u32(self): return int.from_bytes(self.take(4), "big")
    def i32(self): return int.from_bytes(self.take(4), "big", signed=True)
    def fr_raw(self): return int.from_bytes(self.take(32), "little") * MONT_INV_R % R
    def fq_raw(self): return int.from_bytes(self.take(32), "little") * MONT_INV_Q % P
    def g1_raw(self):
        x, y = self.fq_raw(), self.fq_raw()
        return None if (x == 0 and y == 0) else (x, y)


def read_params_raw(data):
    r = Reader(data)
    k = int.from_bytes(r.take(4), "little")
    g = r.g1_raw()
    g2 = ((r.fq_raw(), r.fq_raw()), (r.fq_raw(), r.fq_raw()))
    s_g2 = ((r.fq_raw(), r.fq_raw()), (r.fq_raw(), r.fq_raw()))
    return dict(k=k, g=g, g2=g2, s_g2=s_g2)


def _read_expr(r):
    num_vars, nt = r.u32(), r.u32()
    terms = []
    for _ in range(nt):
        c = r.u16(); nf = r.u32()
        terms.append((c, [(r.u32(), r.u32()) for _ in range(nf)]))
    return terms


def read_vk_raw(data):
    """VerifyingKey::read with SerdeFormat::RawBytes (plonk/vk.rs:76-115, 274-365)"""
    r = Reader(data)
    vk = dict(k=r.u32())
    vk["fixed_commitments"] = [r.g1_raw() for _ in range(r.u32())]
    vk["cs_degree"] = r.u32()
    nfix, nadv, ninst, nsel, nch, ng, nl, ns, nc = (r.u32() for _ in range(9))
    vk.update(num_fixed=nfix, num_advice=nadv, num_instance=ninst, num_challenges=nch)
    vk["advice_phase"] = [r.u8() for _ in range(nadv)]
    vk["challenge_phase"] = [r.u8() for _ in range(nch)]
    vk["num_advice_queries"] = [r.u32() for _ in range(nadv)]
    vk["advice_queries"] = [(r.u32(), r.u8(), r.i32()) for _ in range(sum(vk["num_advice_queries"]))]  # (col, phase, rot)
    vk["instance_queries"] = [(r.u32(), r.i32()) for _ in range(ninst)]
    vk["fixed_queries"] = [(r.u32(), r.i32()) for _ in range(nfix)]
    vk["perm_columns"] = [(r.u32(), r.u8()) for _ in range(r.u32())]
    vk["gates"] = [_read_expr(r) for _ in range(ng)]
    vk["lookups"] = []
    for _ in range(nl):
        m = r.u32(); ins, tabs = [], []
        for _ in range(m): ins.append(_read_expr(r)); tabs.append(_read_expr(r))
        vk["lookups"].append((ins, tabs))
    vk["shuffles"] = []
    for _ in range(ns):
        m = r.u32(); ins, shs = [], []
        for _ in range(m): ins.append(_read_expr(r)); shs.append(_read_expr(r))
        vk["shuffles"].append((ins, shs))
    vk["coeff_vals"] = [r.fr_raw() for _ in range(nc)]
    vk["perm_commitments"] = [r.g1_raw() for _ in range(len(vk["perm_columns"]))]
    r.take(nsel * (((1 << vk["k"]) + 7) // 8))
    vk["transcript_repr"] = r.fr_raw()
    return vk


# ------------------------------------------------------------------ legacy Keccak-256 (hashlib only has SHA3 padding)
_KRC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808a, 0x8000000080008000, 0x000000000000808b, 0x0000000080000001,
        0x8000000080008081, 0x8000000000008009, 0x000000000000008a, 0x0000000000000088, 0x0000000080008009, 0x000000008000000a,
        0x000000008000808b, 0x800000000000008b, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
        0x000000000000800a, 0x800000008000000a, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_M64 = (1 << 64) - 1


def _keccak_f(a):
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & _M64 if n else v
    for rc in _KRC:
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [a[i] ^ d[i % 5] for i in range(25)]
        # rho + pi, walking the (x, y) -> (y, 2x + 3y) cycle with the triangular-number offsets
        b = [0] * 25
        x, y = 1, 0
        b[0] = a[0]
        for t in range(24):
            nx, ny = y, (2 * x + 3 * y) % 5
            b[nx + 5 * ny] = rol(a[x + 5 * y], ((t + 1) * (t + 2) // 2) % 64)
            x, y = nx, ny
        a = [b[i] ^ ((~b[(i % 5 + 1) % 5 + 5 * (i // 5)]) & b[(i % 5 + 2) % 5 + 5 * (i // 5)] & _M64) for i in range(25)]
        a[0] ^= rc
    return a


def keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    pad = rate - len(msg) % rate
    msg += b"\x01" + b"\x00" * (pad - 1)
    msg[-1] |= 0x80
    st = [0] * 25
    for off in range(0, len(msg), rate):
        for i in range(17):
            st[i] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
        st = _keccak_f(st)
    return b"".join(st[i].to_bytes(8, "little") for i in range(4))


# ------------------------------------------------------------------ transcript (transcript/mod.rs:104-272)
BLAKE2B, KECCAK256 = 0, 1
SHPLONK, GWC = 0, 1


class Transcript:
    def __init__(self, proof, kind=BLAKE2B):
        self.kind = kind
        self.h = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.buf = bytearray(b"Halo2-Transcript")      # Keccak256Read::init absorbs the label (mod.rs:143-145)
        self.r = Reader(proof)
    def _absorb(self, b):
        if self.kind == KECCAK256: self.buf += b
        else: self.h.update(b)
    def common_scalar(self, s): self._absorb(b"\x02" + s.to_bytes(32, "little"))
    def common_point(self, pt):
        if pt is None: raise ValueError("cannot write points at infinity to the transcript")
        self._absorb(b"\x01" + pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little"))
    def squeeze(self):
        self._absorb(b"\x00")
        if self.kind == KECCAK256:   # mod.rs:239-254
            wide = keccak256(bytes(self.buf) + b"\x0a") + keccak256(bytes(self.buf) + b"\x0b")
            return int.from_bytes(wide, "little") % R
        return int.from_bytes(self.h.copy().digest(), "little") % R
    def read_point(self):
        pt = g1_decompress(self.r.take(32)); self.common_point(pt); return pt
    def read_scalar(self):
        v = int.from_bytes(self.r.take(32), "little")
        if v >= R: raise ValueError("invalid field element encoding in proof")
        self.common_scalar(v); return v


# ------------------------------------------------------------------ verify_proof (lib.rs:33-425) + SHPLONK
def _eval_expr(terms, coeffs, adv, fix, inst, ch):
    vars_ = adv + fix + inst + ch
    if not terms: raise ZeroDivisionError("empty polynomial: the reference panics (multilinear.rs:65)")
    total = 0
    for c, factors in terms:
        prod = 1
        for v, pw in factors: prod = prod * pow(vars_[v], pw, R) % R
        total = (total + coeffs[c] * prod) % R
    return total


def _lagrange_at(points, evals, u):
    total = 0
    for j, xj in enumerate(points):
        num, den = 1, 1
        for k2, xk in enumerate(points):
            if k2 != j: num = num * (u - xk) % R; den = den * (xj - xk) % R
        total = (total + evals[j] * num * pow(den, -1, R)) % R
    return total


def guard(params, vk, instances, proof, multiopen=SHPLONK, transcript=BLAKE2B, circuit_instances=1):
    """Returns dict(challenges, right=[(scalar, point)], left=[(scalar, point)]) or raises (ValueError: transcript
    error with .args[1] in {"transcript", "opening"}; ZeroDivisionError: the reference would panic).
    circuit_instances = M = `instances.len()` of verify_proof (lib.rs:43): `instances` then holds the M x columns of the M circuit
    instances that share the transcript, instance by instance; every per-instance read / expression / query below is repeated
    in the reference's order (lib.rs:76-161 reads, :220-253 evaluations, :273-346 expressions, :349-391 queries)."""
    n, k = 1 << vk["k"], vk["k"]
    M = circuit_instances
    if len(instances) != M * vk["num_instance"]: raise ValueError("InvalidInstances", "invalid_instances")
    NI = vk["num_instance"]
    inst_of = [instances[q * NI:(q + 1) * NI] for q in range(M)]
    omega = ROOT_OF_UNITY
    for _ in range(S - k): omega = omega * omega % R
    omega_inv = pow(omega, -1, R)
    rot = lambda x, r_: x * (pow(omega, r_, R) if r_ >= 0 else pow(omega_inv, -r_, R)) % R
    tr = Transcript(proof, transcript)
    try:
        tr.common_scalar(vk["transcript_repr"])
        for col in instances:
            for v in col: tr.common_scalar(v)
        A, Ch = vk["num_advice"], vk["num_challenges"]
        advice_c, challenges = [[None] * A for _ in range(M)], [0] * Ch
        for phase in range(max(vk["advice_phase"], default=0) + 1):
            for q in range(M):
                for i in range(A):
                    if vk["advice_phase"][i] == phase: advice_c[q][i] = tr.read_point()
            for i in range(Ch):
                if vk["challenge_phase"][i] == phase: challenges[i] = tr.squeeze()
        theta = tr.squeeze()
        L, Sh, Pn = len(vk["lookups"]), len(vk["shuffles"]), len(vk["perm_columns"])
        lk_in, lk_tab = [[None] * L for _ in range(M)], [[None] * L for _ in range(M)]
        for q in range(M):
            for i in range(L): lk_in[q][i] = tr.read_point(); lk_tab[q][i] = tr.read_point()
        beta, gamma = tr.squeeze(), tr.squeeze()
        chunk = vk["cs_degree"] - 2
        nsets = (Pn + chunk - 1) // chunk if Pn else 0
        perm_z = [[tr.read_point() for _ in range(nsets)] for _ in range(M)]
        lk_z = [[tr.read_point() for _ in range(L)] for _ in range(M)]
        sh_z = [[tr.read_point() for _ in range(Sh)] for _ in range(M)]
        random_c = tr.read_point()
        y = tr.squeeze()
        H = vk["cs_degree"] - 1
        h_c = [tr.read_point() for _ in range(H)]
        x = tr.squeeze()
        xn = pow(x, n, R)
        n_inv = pow(n, -1, R)
        l_cache = {}
        def l_i(i):
            if i not in l_cache: l_cache[i] = rot(pow((x - rot(1, i)) % R, -1, R) * ((xn - 1) * n_inv % R) % R, i)
            return l_cache[i]
        inst_evals = []
        for q in range(M):
            inst_evals.append([sum(v * l_i(j - r_) for j, v in enumerate(inst_of[q][col])) % R for col, r_ in vk["instance_queries"]])
        adv_e = [[tr.read_scalar() for _ in vk["advice_queries"]] for _ in range(M)]
        fix_e = [tr.read_scalar() for _ in vk["fixed_queries"]]
        random_e = tr.read_scalar()
        sig_e = [tr.read_scalar() for _ in range(Pn)]
        pz = []
        for q in range(M):
            pzq = []
            for i in range(nsets):
                e, nx = tr.read_scalar(), tr.read_scalar()
                pzq.append((e, nx, tr.read_scalar() if i + 1 < nsets else None))
            pz.append(pzq)
        lk_e = [[[tr.read_scalar() for _ in range(5)] for _ in range(L)] for _ in range(M)]   # product, product_next, input, input_inv, table
        sh_e = [[[tr.read_scalar() for _ in range(2)] for _ in range(Sh)] for _ in range(M)]
    except ValueError as e:
        raise ValueError(e.args[0], "transcript")

    bf = max(3, max(vk["num_advice_queries"], default=1)) + 2
    l_last, l_0 = l_i(-(bf + 1)), l_i(0)
    l_blind = sum(l_i(r_) for r_ in range(-bf, 0)) % R
    cv = vk["coeff_vals"]
    active = (1 - (l_last + l_blind)) % R
    exprs = []
    for q in range(M):
        ev = lambda t, q=q: _eval_expr(t, cv, adv_e[q], fix_e, inst_evals[q], challenges)
        exprs += [ev(g) for g in vk["gates"]]
        def col_eval(col, q=q):
            idx, typ = col
            if typ <= 2: return adv_e[q][[i for i, qq in enumerate(vk["advice_queries"]) if qq[0] == idx and qq[2] == 0][0]]
            if typ == 255: return fix_e[[i for i, qq in enumerate(vk["fixed_queries"]) if qq[0] == idx and qq[1] == 0][0]]
            return inst_evals[q][[i for i, qq in enumerate(vk["instance_queries"]) if qq[0] == idx and qq[1] == 0][0]]
        if nsets:
            pq = pz[q]
            exprs.append(l_0 * (1 - pq[0][0]) % R)
            exprs.append((pq[-1][0] * pq[-1][0] - pq[-1][0]) * l_last % R)
            for i in range(1, nsets): exprs.append((pq[i][0] - pq[i - 1][2]) * l_0 % R)
            for ci in range(nsets):
                cols = vk["perm_columns"][ci * chunk:(ci + 1) * chunk]
                left, right = pq[ci][1], pq[ci][0]
                cur = beta * x % R * pow(DELTA, ci * chunk, R) % R
                for j, col in enumerate(cols):
                    v = col_eval(col)
                    left = left * ((v + beta * sig_e[ci * chunk + j] + gamma) % R) % R
                    right = right * ((v + cur + gamma) % R) % R
                    cur = cur * DELTA % R
                exprs.append((left - right) * active % R)
        def compress(es, ev=ev):
            acc = 0
            for e in es: acc = (acc * theta + ev(e)) % R
            return acc
        for i in range(L):
            z, zn, a_, ai, s_ = lk_e[q][i]
            exprs.append(l_0 * (1 - z) % R)
            exprs.append(l_last * (z * z - z) % R)
            left = zn * (a_ + beta) % R * (s_ + gamma) % R
            right = z * (compress(vk["lookups"][i][0]) + beta) % R * (compress(vk["lookups"][i][1]) + gamma) % R
            exprs.append((left - right) * active % R)
            exprs.append(l_0 * (a_ - s_) % R)
            exprs.append((a_ - s_) * (a_ - ai) % R * active % R)
        for i in range(Sh):
            z, zn = sh_e[q][i]
            exprs.append(l_0 * (1 - z) % R)
            exprs.append(l_last * (z * z - z) % R)
            left = zn * (compress(vk["shuffles"][i][1]) + gamma) % R
            right = z * (compress(vk["shuffles"][i][0]) + gamma) % R
            exprs.append((left - right) * active % R)
    h_eval = 0
    for e in exprs: h_eval = (h_eval * y + e) % R
    if (xn - 1) % R == 0: raise ZeroDivisionError("xn - 1 == 0 (vanishing.rs:100)")
    expected_h = h_eval * pow(xn - 1, -1, R) % R
    h_msm = [(pow(xn, i, R), h_c[i]) for i in range(H - 1, -1, -1)]   # bases h_{H-1}..h_0

    # queries (lib.rs:349-414); commitment identity = a hashable key (per-instance commitments carry the instance index)
    Q = []
    base = {("rand", 0): random_c}
    for q in range(M):
        for (col, _, r_), e in zip(vk["advice_queries"], adv_e[q]): Q.append((("adv", q, col), rot(x, r_), e))
        for i in range(nsets): Q.append((("pz", q, i), x, pz[q][i][0])); Q.append((("pz", q, i), rot(x, 1), pz[q][i][1]))
        for i in range(nsets - 2, -1, -1): Q.append((("pz", q, i), rot(x, -(bf + 1)), pz[q][i][2]))
        for i in range(L):
            z, zn, a_, ai, s_ = lk_e[q][i]
            Q += [(("lkz", q, i), x, z), (("lka", q, i), x, a_), (("lks", q, i), x, s_), (("lka", q, i), rot(x, -1), ai), (("lkz", q, i), rot(x, 1), zn)]
        for i in range(Sh): Q += [(("shz", q, i), x, sh_e[q][i][0]), (("shz", q, i), rot(x, 1), sh_e[q][i][1])]
        for i in range(A): base[("adv", q, i)] = advice_c[q][i]
        for i in range(nsets): base[("pz", q, i)] = perm_z[q][i]
        for i in range(L): base[("lkz", q, i)] = lk_z[q][i]; base[("lka", q, i)] = lk_in[q][i]; base[("lks", q, i)] = lk_tab[q][i]
        for i in range(Sh): base[("shz", q, i)] = sh_z[q][i]
    for (col, r_), e in zip(vk["fixed_queries"], fix_e): Q.append((("fix", col), rot(x, r_), e))
    for i in range(Pn): Q.append((("sig", i), x, sig_e[i]))
    Q.append((("hmsm", 0), x, expected_h))
    Q.append((("rand", 0), x, random_e))
    for i, c in enumerate(vk["fixed_commitments"]): base[("fix", i)] = c
    for i, c in enumerate(vk["perm_commitments"]): base[("sig", i)] = c

    if multiopen == GWC:   # gwc.rs:54-163
        try:
            gv = tr.squeeze()
            by_point = []
            for q in Q:
                for e in by_point:
                    if e[0] == q[1]: e[1].append(q); break
                else: by_point.append((q[1], [q]))
            ws = [tr.read_point() for _ in by_point]
            gu = tr.squeeze()
        except ValueError as e:
            raise ValueError(e.args[0], "opening")
        commitment_multi, witness, witness_aux, eval_multi = [], [], [], 0
        for i, ((z, qs), wi) in enumerate(zip(by_point, ws)):
            pu = pow(gu, i, R)
            eval_batch = 0
            for j, (c, _, e) in enumerate(qs):
                pv = pow(gv, j, R)
                if c[0] == "hmsm":
                    for sc, b in h_msm: commitment_multi.append((sc * pv % R * pu % R, b))
                else:
                    commitment_multi.append((pv * pu % R, base[c]))
                eval_batch = (eval_batch + pv * e) % R
            eval_multi = (eval_multi + pu * eval_batch) % R
            witness_aux.append((pu * z % R, wi))
            witness.append((pu, wi))
        neg_g = (params["g"][0], (-params["g"][1]) % P)
        return dict(challenges=challenges + [theta, beta, gamma, y, x, gv, gu], right=witness_aux + commitment_multi + [(eval_multi, neg_g)], left=witness)

    # shplonk.rs:58-149
    cmap, super_pts = [], set()
    for c, pt, _ in Q:
        super_pts.add(pt)
        for e in cmap:
            if e[0] == c: e[1].add(pt); break
        else: cmap.append((c, {pt}))
    rsets = []
    for c, pts in cmap:
        for rs in rsets:
            if rs[0] == pts: rs[1].append(c); break
        else: rsets.append((pts, [c]))
    def eval_of(c, pt): return next(e for (c2, p2, e) in Q if c2 == c and p2 == pt)
    try:
        sy, sv = tr.squeeze(), tr.squeeze()
        h1 = tr.read_point()
        su = tr.squeeze()
        h2 = tr.read_point()
    except ValueError as e:
        raise ValueError(e.args[0], "opening")
    vanish = lambda pts: (lambda acc: acc)(__import__("functools").reduce(lambda a, p_: a * (su - p_) % R, pts, 1))
    right, r_outer, z_0, z0_diff_inv = [], 0, 0, 0
    for i, (pts, commits) in enumerate(rsets):
        pts_sorted = sorted(pts)
        z_diff = vanish([p_ for p_ in sorted(super_pts) if p_ not in pts])
        if i == 0:
            z_0 = vanish(pts_sorted)
            if z_diff == 0: raise ZeroDivisionError("z_diff_0 == 0 (shplonk.rs:215)")
            z0_diff_inv = pow(z_diff, -1, R); z_diff = 1
        else:
            z_diff = z_diff * z0_diff_inv % R
        w = pow(sv, i, R) * z_diff % R
        r_inner = 0
        for j, c in enumerate(commits):
            py = pow(sy, j, R)
            r_inner = (r_inner + py * _lagrange_at(pts_sorted, [eval_of(c, p_) for p_ in pts_sorted], su)) % R
            if c[0] == "hmsm":
                for sc, b in h_msm: right.append((sc * py % R * w % R, b))
            else:
                right.append((py * w % R, base[c]))
        r_outer = (r_outer + pow(sv, i, R) * r_inner % R * z_diff) % R
    right.append(((-r_outer) % R, params["g"]))
    right.append(((-z_0) % R, h1))
    right.append((su, h2))
    return dict(challenges=challenges + [theta, beta, gamma, y, x, sy, sv, su], right=right, left=[(1, h2)])


def msm(terms):
    acc = None
    for s, b in terms: acc = g1_add(acc, g1_mul(s, b))
    return acc


def verify_single(params, vk, instances, proof, multiopen=SHPLONK, transcript=BLAKE2B, circuit_instances=1):
    """SingleStrategy: 0 ok, -2 ConstraintSystemFailure, -5 Transcript, -4 Opening, -1 InvalidInstances, -7 panic"""
    try:
        g = guard(params, vk, instances, proof, multiopen, transcript, circuit_instances)
    except ValueError as e:
        return {"transcript": -5, "opening": -4, "invalid_instances": -1}[e.args[1]]
    except ZeroDivisionError:
        return -7
    return 0 if pairing_check(msm(g["left"]), msm(g["right"]), params["s_g2"], params["g2"]) else -2
