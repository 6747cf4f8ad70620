"""Test inputs: VK / params / proofs produced by the oracle's test-only keygen + prover
(oracle/prover.cpp).  The reference ships no proof bytes, so every proof used by the
tests is generated here, deterministically from seeds."""
import ctypes
import os

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRS_PATH = os.path.join(ROOT, "tests", "golden", "kzg_bn254_8.srs")
RAW = 1  # SerdeFormat::RawBytes
R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def le32(x):
    return int(x).to_bytes(32, "little")


SHPLONK, GWC = 0, 1          # multi-open argument (poly/kzg/multiopen/{shplonk,gwc}.rs)
BLAKE2B, KECCAK256 = 0, 1    # transcript hash (transcript/mod.rs:104-116)


class Setup:
    def __init__(self, L, handle, ninst_cols):
        self.L, self.h, self.ninst_cols = L, handle, ninst_cols
        self.multiopen, self.transcript = SHPLONK, BLAKE2B
        self.circuit_instances = 1   # instances.len() of verify_proof (lib.rs:33-49); set_circuit_instances()
        buf = ctypes.create_string_buffer(1 << 22)
        n = L.h2o_setup_vk(handle, RAW, buf, len(buf))
        self.vk = buf.raw[:n]
        n = L.h2o_setup_params(handle, RAW, buf, len(buf))
        self.params = buf.raw[:n]

    def set_options(self, multiopen=SHPLONK, transcript=BLAKE2B):
        """Options of both the test prover and the oracle verifier for this setup."""
        self.multiopen, self.transcript = multiopen, transcript
        self.L.h2o_setup_set_options(self.h, multiopen, transcript)
        return self

    def set_circuit_instances(self, m):
        """Proofs of this setup carry m circuit instances per transcript: `instances` of every verifier-side helper is then the
        m x columns list, instance-major."""
        self.circuit_instances = m
        return self

    def use(self):
        """Select this setup's options for the oracle's verifier entry points (thread-local on the C side)."""
        self.L.h2o_set_verify_options(self.multiopen, self.transcript)
        self.L.h2o_set_circuit_instances(self.circuit_instances)

    def free(self):
        if self.h:
            self.L.h2o_setup_free(self.h)
            self.h = None


def setup_vector_mul(k=8, n_mul=10, use_reference_srs=False, s_seed=42):
    L = oracle_lib.load()
    srs = open(SRS_PATH, "rb").read() if use_reference_srs else None
    h = L.h2o_setup_vector_mul(k, n_mul, srs, len(srs) if srs else 0, s_seed)
    assert h
    s = Setup(L, h, 1)
    s.n_mul = n_mul
    return s


def prove_vector_mul(s, a, b, rng_seed=7):
    """a, b: lists of ints (len n_mul) -> (proof bytes, instances [[c...]])"""
    ab = b"".join(le32(x) for x in a)
    bb = b"".join(le32(x) for x in b)
    buf = ctypes.create_string_buffer(1 << 16)
    inst = ctypes.create_string_buffer(32 * s.n_mul)
    n = s.L.h2o_prove_vector_mul(s.h, ab, bb, rng_seed, buf, len(buf), inst)
    assert n > 0
    return buf.raw[:n], [[inst.raw[32 * i:32 * i + 32] for i in range(s.n_mul)]]


def prove_vector_mul_len(s, a, b, inst_len, rng_seed=7):
    """As prove_vector_mul with only the first inst_len products public (the rest must be zero): instances [[c_0..c_{len-1}]]"""
    ab = b"".join(le32(x) for x in a)
    bb = b"".join(le32(x) for x in b)
    buf = ctypes.create_string_buffer(1 << 16)
    inst = ctypes.create_string_buffer(32 * max(inst_len, 1))
    n = s.L.h2o_prove_vector_mul_len(s.h, ab, bb, inst_len, rng_seed, buf, len(buf), inst)
    assert n > 0
    return buf.raw[:n], [[inst.raw[32 * i:32 * i + 32] for i in range(inst_len)]]


def prove_vector_mul_multi(s, m, seed=1, rng_seed=7):
    """ONE proof over m circuit instances (pseudo-random a, b from `seed`): -> (proof, instances) with instances = m columns,
    instance-major — the flattened `instances: &[&[&[Fr]]]` the verifier-side helpers take after s.set_circuit_instances(m)"""
    import random
    rnd = random.Random(seed)
    ab = b"".join(le32(rnd.randrange(R_MOD)) for _ in range(m * s.n_mul))
    bb = b"".join(le32(rnd.randrange(R_MOD)) for _ in range(m * s.n_mul))
    buf = ctypes.create_string_buffer(1 << 18)
    inst = ctypes.create_string_buffer(32 * m * s.n_mul)
    n = s.L.h2o_prove_vector_mul_multi(s.h, m, ab, bb, rng_seed, buf, len(buf), inst)
    assert n > 0
    return buf.raw[:n], [[inst.raw[32 * (q * s.n_mul + i):32 * (q * s.n_mul + i + 1)] for i in range(s.n_mul)] for q in range(m)]


def prove_shuffle_multi(s, m, data_seed=5, break_at=-1, rng_seed=9):
    buf = ctypes.create_string_buffer(1 << 18)
    n = s.L.h2o_prove_shuffle_multi(s.h, m, data_seed, break_at, rng_seed, buf, len(buf))
    assert n > 0
    return buf.raw[:n], []


def prove_wide_multi(s, m, witness_seed=3, tamper_at=-1, rng_seed=13):
    buf = ctypes.create_string_buffer(1 << 20)
    inst = ctypes.create_string_buffer(32 * 8 * m)
    n = s.L.h2o_prove_wide_multi(s.h, m, witness_seed, tamper_at, rng_seed, buf, len(buf), inst)
    assert n > 0
    return buf.raw[:n], [[inst.raw[32 * (8 * q + i):32 * (8 * q + i + 1)] for i in range(8)] for q in range(m)]


def prove_vector_mul_batch(s, count, seed=1, threads=8, proof_len=None):
    if proof_len is None:
        proof_len = len(prove_vector_mul(s, [1] * s.n_mul, [1] * s.n_mul)[0])
    proofs = ctypes.create_string_buffer(count * proof_len)
    inst = ctypes.create_string_buffer(count * 32 * s.n_mul)
    got = s.L.h2o_prove_vector_mul_batch(s.h, count, seed, threads, proofs, proof_len, inst)
    assert got == count
    P = [proofs.raw[i * proof_len:(i + 1) * proof_len] for i in range(count)]
    I = [[[inst.raw[32 * (i * s.n_mul + j):32 * (i * s.n_mul + j + 1)] for j in range(s.n_mul)]] for i in range(count)]
    return P, I


def setup_shuffle(k=8, W=4, H=32, s_seed=43):
    L = oracle_lib.load()
    h = L.h2o_setup_shuffle(k, W, H, None, 0, s_seed)
    assert h
    return Setup(L, h, 0)


def prove_shuffle(s, data_seed=5, break_it=False, rng_seed=9):
    buf = ctypes.create_string_buffer(1 << 16)
    n = s.L.h2o_prove_shuffle(s.h, data_seed, 1 if break_it else 0, rng_seed, buf, len(buf))
    assert n > 0
    return buf.raw[:n], []


def setup_wide(k=8, A=8, F=5, L_=1, Sh=1, deg=3, seed=11, s_seed=44):
    L = oracle_lib.load()
    h = L.h2o_setup_wide(k, A, F, L_, Sh, deg, seed, None, 0, s_seed)
    assert h
    return Setup(L, h, 1)


def prove_wide(s, witness_seed=3, tamper=False, rng_seed=13):
    buf = ctypes.create_string_buffer(1 << 20)
    inst = ctypes.create_string_buffer(32 * 8)
    n = s.L.h2o_prove_wide(s.h, witness_seed, 1 if tamper else 0, rng_seed, buf, len(buf), inst)
    assert n > 0
    return buf.raw[:n], [[inst.raw[32 * i:32 * i + 32] for i in range(8)]]


# ---- oracle-side verification helpers (the checker)
def _flat(instances):
    flat = b"".join(b"".join(col) for col in instances)
    lens = [len(col) for col in instances]
    return flat, (ctypes.c_size_t * max(len(lens), 1))(*lens), len(lens)


def _use(s):
    if hasattr(s, "use"):
        s.use()
    else:
        s.L.h2o_set_verify_options(getattr(s, "multiopen", 0), getattr(s, "transcript", 0))
        s.L.h2o_set_circuit_instances(1)


def oracle_verify_single(s, proof, instances):
    _use(s)
    f, cl, nc = _flat(instances)
    return s.L.h2o_verify_single(s.params, len(s.params), RAW, s.vk, len(s.vk), RAW, f, cl, nc, proof, len(proof))


def oracle_guard(s, proof, instances, cap=4096):
    _use(s)
    f, cl, nc = _flat(instances)
    rs, rb = ctypes.create_string_buffer(32 * cap), ctypes.create_string_buffer(64 * cap)
    ls, lb = ctypes.create_string_buffer(32 * 16), ctypes.create_string_buffer(64 * 16)
    ch = ctypes.create_string_buffer(32 * 64)
    nr, nl, ncz = ctypes.c_size_t(cap), ctypes.c_size_t(16), ctypes.c_size_t(64)
    rc = s.L.h2o_guard_msm(s.params, len(s.params), RAW, s.vk, len(s.vk), RAW, f, cl, nc, proof, len(proof), rs, rb, ctypes.byref(nr), ls, lb, ctypes.byref(nl), ch, ctypes.byref(ncz))
    if rc != 0:
        return rc, None
    sp = lambda buf, sz, n: [buf.raw[sz * i:sz * (i + 1)] for i in range(n)]
    return 0, dict(right_scalars=sp(rs, 32, nr.value), right_bases=sp(rb, 64, nr.value), left_scalars=sp(ls, 32, nl.value), left_bases=sp(lb, 64, nl.value),
                   challenges=sp(ch, 32, ncz.value))


def oracle_verify_batch(s, proofs, instances, rand):
    _use(s)
    n = len(proofs)
    plen = len(proofs[0]) if n else 0
    assert all(len(p) == plen for p in proofs)
    pf = b"".join(proofs)
    flat = b""
    cl, nc = (ctypes.c_size_t * 1)(0), 0
    for inst in instances:
        f, cl, nc = _flat(inst)
        flat += f
    if n == 0:
        nc = s.ninst_cols
    rb = b"".join(le32(r) if not isinstance(r, (bytes, bytearray)) else r for r in rand)
    st = (ctypes.c_int * max(n, 1))()
    ok = ctypes.c_int(0)
    left, right = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
    rc = s.L.h2o_verify_batch(s.params, len(s.params), RAW, s.vk, len(s.vk), RAW, n, pf, plen, flat, cl, nc, rb, st, ctypes.byref(ok), left, right)
    assert rc == 0, rc
    return bool(ok.value), list(st)[:n], left.raw, right.raw


def oracle_pairing_check(s, left_xy, right_xy):
    """DualMSM::check on two evaluated channels (poly/kzg/msm.rs:185-203) by the oracle"""
    ok = ctypes.c_int(0)
    rc = s.L.h2o_pairing_check(s.params, len(s.params), RAW, left_xy, right_xy, ctypes.byref(ok))
    assert rc == 0, rc
    return bool(ok.value)


def oracle_accumulate(items, rand):
    """N x verify_proof on ONE AccumulatorStrategy followed by finalize(), restated from the oracle's per-proof Guards:
    items = [(setup, proof, instances)] in call order (setups may differ in VK and instance shape, same params); proof i's Guard
    is scaled by the product of the draws of all LATER proofs (kzg/strategy.rs:129, msm.rs:173-176: process() scales the
    accumulator by a fresh draw before adding the next Guard).  -> (ok, statuses, left_xy, right_xy)"""
    n = len(items)
    mult = [1] * n
    run = 1
    for i in range(n - 1, -1, -1):
        mult[i] = run
        run = run * (int.from_bytes(rand[i], "little") if isinstance(rand[i], (bytes, bytearray)) else int(rand[i])) % R_MOD
    ls, lb, rs, rb, statuses = [], [], [], [], []
    for (s, proof, inst), m in zip(items, mult):
        rc, g = oracle_guard(s, proof, inst)
        statuses.append(rc)
        if rc != 0:
            continue
        ls += [int.from_bytes(x, "little") * m % R_MOD for x in g["left_scalars"]]; lb += g["left_bases"]
        rs += [int.from_bytes(x, "little") * m % R_MOD for x in g["right_scalars"]]; rb += g["right_bases"]
    L = items[0][0].L
    zero = b"\0" * 64
    left = oracle_lib.g1_msm(L, ls, lb) if ls else zero
    right = oracle_lib.g1_msm(L, rs, rb) if rs else zero
    ok = oracle_pairing_check(items[0][0], left, right) and not any(statuses)
    return ok, statuses, left, right
