"""ctypes/numpy binding of oracle/libgl_oracle.so -- TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Every function restates plonky2 0.1.4 behaviour; see gl_oracle.c for the per-function citations.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgl_oracle.so")
P = 0xFFFFFFFF00000001
u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the C restatement (gcc). Safe to call repeatedly."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgl_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        L = _lib
        for name in ("glo_add", "glo_sub", "glo_mul", "glo_pow"):
            getattr(L, name).restype = C.c_uint64
            getattr(L, name).argtypes = [C.c_uint64, C.c_uint64]
        L.glo_inv.restype = C.c_uint64; L.glo_inv.argtypes = [C.c_uint64]
        L.glo_root_of_unity.restype = C.c_uint64; L.glo_root_of_unity.argtypes = [C.c_int]
        L.glo_max_threads.restype = C.c_int
        L.glo_merkle_num_digests.restype = C.c_size_t
        L.glo_merkle_num_digests.argtypes = [C.c_size_t, C.c_int]
        L.glo_challenger_size.restype = C.c_size_t
        L.glo_challenger_get.restype = C.c_uint64
        L.glo_get_hasher.restype = C.c_int
    return _lib


def _a(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------- field
def add(a, b): return lib().glo_add(a, b)
def sub(a, b): return lib().glo_sub(a, b)
def mul(a, b): return lib().glo_mul(a, b)
def fpow(a, e): return lib().glo_pow(a, e)
def inv(a): return lib().glo_inv(a)
def root_of_unity(lg): return lib().glo_root_of_unity(lg)


def vec_mul(a, b):
    a, b = _a(a), _a(b); o = np.empty_like(a)
    lib().glo_vec_mul(_p(a), _p(b), _p(o), C.c_size_t(a.size)); return o


def vec_add(a, b):
    a, b = _a(a), _a(b); o = np.empty_like(a)
    lib().glo_vec_add(_p(a), _p(b), _p(o), C.c_size_t(a.size)); return o


def vec_sub(a, b):
    a, b = _a(a), _a(b); o = np.empty_like(a)
    lib().glo_vec_sub(_p(a), _p(b), _p(o), C.c_size_t(a.size)); return o


def vec_scale(a, s):
    a = _a(a); o = np.empty_like(a)
    lib().glo_vec_scale(_p(a), C.c_uint64(int(s)), _p(o), C.c_size_t(a.size)); return o


def vec_inv(a):
    a = _a(a); o = np.empty_like(a)
    lib().glo_vec_inv(_p(a), _p(o), C.c_size_t(a.size)); return o


def rand_field(rng, shape):
    """Uniform elements of [0, p) from a numpy Generator (rejection on the 2^-32 tail)."""
    x = rng.integers(0, 1 << 64, size=shape, dtype=np.uint64)
    bad = x >= np.uint64(P)
    while bad.any():
        x[bad] = rng.integers(0, 1 << 64, size=int(bad.sum()), dtype=np.uint64)
        bad = x >= np.uint64(P)
    return x


# ---------------------------------------------------------------- Poseidon
def poseidon_permute(state):
    s = _a(state).copy(); assert s.size == 12
    lib().glo_poseidon_permute(_p(s)); return s


def hash_no_pad(x):
    x = _a(x); o = np.empty(4, np.uint64)
    lib().glo_hash_no_pad(_p(x), C.c_size_t(x.size), _p(o)); return o


def hash_or_noop(x):
    x = _a(x); o = np.empty(4, np.uint64)
    lib().glo_hash_or_noop(_p(x), C.c_size_t(x.size), _p(o)); return o


def hash_pad(x):
    x = _a(x); o = np.empty(4, np.uint64)
    lib().glo_hash_pad(_p(x), C.c_size_t(x.size), _p(o)); return o


def two_to_one(l, r):
    l, r = _a(l), _a(r); o = np.empty(4, np.uint64)
    lib().glo_two_to_one(_p(l), _p(r), _p(o)); return o


# ---------------------------------------------------------------- Keccak (gl_keccak.c; KeccakGoldilocksConfig)
def keccak256(msg: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().glo_keccak256(C.c_char_p(bytes(msg)), C.c_size_t(len(msg)), out)
    return out.raw


def keccak_hash_no_pad(x):
    x = _a(x); o = np.empty(4, np.uint64)
    lib().glo_keccak_hash_no_pad(_p(x), C.c_size_t(x.size), _p(o)); return o


def keccak_hash_or_noop(x):
    x = _a(x); o = np.empty(4, np.uint64)
    lib().glo_keccak_hash_or_noop(_p(x), C.c_size_t(x.size), _p(o)); return o


def keccak_two_to_one(l, r):
    l, r = _a(l), _a(r); o = np.empty(4, np.uint64)
    lib().glo_keccak_two_to_one(_p(l), _p(r), _p(o)); return o


def keccak_hash_to_elements(h):
    h = _a(h); o = np.empty(4, np.uint64)
    lib().glo_keccak_hash_to_elements(_p(h), _p(o)); return o


def keccak_permute(state):
    s = _a(state).copy(); assert s.size == 12
    lib().glo_keccak_permute(_p(s)); return s


# ---------------------------------------------------------------- FFT
def fft(a):
    a = _a(a).copy(); lib().glo_fft(_p(a), C.c_int(int(a.size).bit_length() - 1)); return a


def ifft(a):
    a = _a(a).copy(); lib().glo_ifft(_p(a), C.c_int(int(a.size).bit_length() - 1)); return a


def coset_fft(a, shift=7):
    a = _a(a).copy(); lib().glo_coset_fft(_p(a), C.c_int(int(a.size).bit_length() - 1), C.c_uint64(shift)); return a


def coset_ifft(a, shift=7):
    a = _a(a).copy(); lib().glo_coset_ifft(_p(a), C.c_int(int(a.size).bit_length() - 1), C.c_uint64(shift)); return a


def lde(coeffs, rate_bits, shift=7):
    c = _a(coeffs); lg = int(c.size).bit_length() - 1
    o = np.empty(c.size << rate_bits, np.uint64)
    lib().glo_lde(_p(c), C.c_int(lg), C.c_int(rate_bits), C.c_uint64(shift), _p(o)); return o


def bitrev_perm(lg):
    n = 1 << lg
    idx = np.arange(n, dtype=np.uint64); r = np.zeros(n, dtype=np.uint64)
    for _ in range(lg):
        r = (r << np.uint64(1)) | (idx & np.uint64(1)); idx >>= np.uint64(1)
    return r.astype(np.int64)


# ---------------------------------------------------------------- Merkle / PolynomialBatch
def merkle_num_digests(nleaves, cap_height):
    return lib().glo_merkle_num_digests(nleaves, cap_height)


class _Hasher:
    """`with _Hasher(h):` selects GenericConfig::Hasher for the oracle calls inside (0 = Poseidon, 1 = KeccakHash<25>); the C
    side keeps it in one process-wide variable (test infrastructure)."""
    def __init__(self, h): self.h = int(h)
    def __enter__(self): self.prev = lib().glo_get_hasher(); lib().glo_set_hasher(self.h)
    def __exit__(self, *a): lib().glo_set_hasher(self.prev)


def merkle_build(leaves, cap_height):
    """leaves [nleaves][leaf_len] -> (digests [num][4] level-major bottom-up, cap [2^cap_height][4])."""
    leaves = _a(leaves); nl, ll = leaves.shape
    nd = merkle_num_digests(nl, cap_height)
    dig = np.empty((nd, 4), np.uint64); cap = np.empty((1 << cap_height, 4), np.uint64)
    rc = lib().glo_merkle_build(_p(leaves), C.c_size_t(nl), C.c_size_t(ll), C.c_int(cap_height), _p(dig), _p(cap))
    if rc: raise ValueError("cap_height exceeds log2(leaves)")
    return dig, cap


def merkle_prove(digests, nleaves, cap_height, index):
    digests = _a(digests); sib = np.empty((64, 4), np.uint64)
    k = lib().glo_merkle_prove(_p(digests), C.c_size_t(nleaves), C.c_int(cap_height), C.c_size_t(index), _p(sib))
    return sib[:k].copy()


def merkle_verify(leaf, index, cap, siblings, hasher=0):
    leaf, cap, siblings = _a(leaf), _a(cap), _a(siblings).reshape(-1, 4)
    ch = int(cap.shape[0]).bit_length() - 1
    with _Hasher(hasher):
        return lib().glo_merkle_verify(_p(leaf), C.c_size_t(leaf.size), C.c_size_t(index), _p(cap), C.c_int(ch),
                                       _p(siblings), C.c_int(siblings.shape[0])) == 0


class Batch:
    """Result of PolynomialBatch::from_values / from_coeffs (fri/oracle.rs)."""
    def __init__(self, coeffs, leaves, digests, cap, rate_bits, cap_height):
        self.coeffs, self.leaves, self.digests, self.cap = coeffs, leaves, digests, cap
        self.rate_bits, self.cap_height = rate_bits, cap_height

    def prove(self, index):
        return merkle_prove(self.digests, self.leaves.shape[0], self.cap_height, index)


def batch_from_coeffs(coeffs, rate_bits=3, cap_height=4, hasher=0):
    with _Hasher(hasher):
        return _batch_from_coeffs(coeffs, rate_bits, cap_height)


def batch_from_values(values, rate_bits=3, cap_height=4, hasher=0):
    with _Hasher(hasher):
        return _batch_from_values(values, rate_bits, cap_height)


def _batch_from_coeffs(coeffs, rate_bits=3, cap_height=4):
    coeffs = _a(coeffs); ncols, n = coeffs.shape; lg = n.bit_length() - 1
    N = n << rate_bits
    leaves = np.empty((N, ncols), np.uint64)
    dig = np.empty((merkle_num_digests(N, cap_height), 4), np.uint64)
    cap = np.empty((1 << cap_height, 4), np.uint64)
    rc = lib().glo_batch_from_coeffs(_p(coeffs), C.c_size_t(ncols), C.c_int(lg), C.c_int(rate_bits),
                                     C.c_int(cap_height), _p(leaves), _p(dig), _p(cap))
    if rc: raise ValueError("bad cap_height")
    return Batch(coeffs, leaves, dig, cap, rate_bits, cap_height)


def _batch_from_values(values, rate_bits=3, cap_height=4):
    values = _a(values); ncols, n = values.shape; lg = n.bit_length() - 1
    N = n << rate_bits
    coeffs = np.empty_like(values)
    leaves = np.empty((N, ncols), np.uint64)
    dig = np.empty((merkle_num_digests(N, cap_height), 4), np.uint64)
    cap = np.empty((1 << cap_height, 4), np.uint64)
    rc = lib().glo_batch_from_values(_p(values), C.c_size_t(ncols), C.c_int(lg), C.c_int(rate_bits),
                                     C.c_int(cap_height), _p(coeffs), _p(leaves), _p(dig), _p(cap))
    if rc: raise ValueError("bad cap_height")
    return Batch(coeffs, leaves, dig, cap, rate_bits, cap_height)


# ---------------------------------------------------------------- Challenger
class Challenger:
    """plonky2's duplex `Challenger<F, H>`; hasher = 1: H = KeccakHash<25> (KeccakPermutation, digests observed as 7-byte chunks)."""
    def __init__(self, hasher=0):
        self.hasher = int(hasher)
        self._buf = C.create_string_buffer(lib().glo_challenger_size())
        lib().glo_challenger_init(self._buf)

    def observe(self, elems):
        e = _a(elems).reshape(-1)
        with _Hasher(self.hasher):
            lib().glo_challenger_observe(self._buf, _p(e), C.c_size_t(e.size))

    def observe_hashes(self, digests):
        """observe_hash / observe_cap: digests [count][4] of the configuration's hasher"""
        d = _a(digests).reshape(-1, 4)
        with _Hasher(self.hasher):
            lib().glo_challenger_observe_hashes(self._buf, _p(d), C.c_size_t(d.shape[0]))

    def get(self):
        with _Hasher(self.hasher):
            return int(lib().glo_challenger_get(self._buf))

    def get_n(self, n):
        return [self.get() for _ in range(n)]

    def get_ext(self):
        return self.get_n(2)


def set_threads(n): lib().glo_set_threads(int(n))
def max_threads(): return lib().glo_max_threads()


# ---------------------------------------------------------------- prove / verify (gl_prover_oracle.c)
class _Gate(C.Structure):
    _fields_ = [("type", C.c_uint32), ("selector_index", C.c_uint32), ("group_start", C.c_uint32),
                ("group_end", C.c_uint32), ("row", C.c_uint32), ("num_constraints", C.c_uint32),
                ("p0", C.c_uint32), ("p1", C.c_uint32)]


class _Circuit(C.Structure):
    _fields_ = [("degree_bits", C.c_uint32), ("num_wires", C.c_uint32), ("num_routed_wires", C.c_uint32),
                ("num_constants", C.c_uint32), ("num_selectors", C.c_uint32), ("num_challenges", C.c_uint32),
                ("quotient_degree_factor", C.c_uint32), ("num_partial_products", C.c_uint32),
                ("num_gate_constraints", C.c_uint32), ("rate_bits", C.c_uint32), ("cap_height", C.c_uint32),
                ("proof_of_work_bits", C.c_uint32), ("num_query_rounds", C.c_uint32), ("num_reductions", C.c_uint32),
                ("reduction_arity_bits", C.c_uint32 * 16), ("num_gates", C.c_uint32), ("num_public_inputs", C.c_uint32),
                ("gates", C.POINTER(_Gate)), ("k_is", C.c_void_p), ("circuit_digest", C.c_uint64 * 4),
                ("constants", C.c_void_p), ("sigmas", C.c_void_p)]


class OracleCircuit:
    """Wraps a plonky2_lib_amd.synth.Circuit (plain attribute bag) for the C oracle."""

    def __init__(self, c, cs_cap=None):
        """cs_cap: optional constants+sigmas Merkle cap (verifier_only data); if omitted it is computed here."""
        self.c = c
        self.hasher = int(getattr(c, "hasher", 0))
        self._k = _a(c.k_is); self._const = _a(c.constants); self._sig = _a(c.sigmas)
        self._gates = (_Gate * len(c.gates))()
        for i, g in enumerate(c.gates):
            for f in ("type", "selector_index", "group_start", "group_end", "row", "num_constraints", "p0", "p1"):
                setattr(self._gates[i], f, int(g[f]))
        s = _Circuit()
        for f in ("degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_selectors", "num_challenges",
                  "quotient_degree_factor", "num_partial_products", "num_gate_constraints", "rate_bits", "cap_height",
                  "proof_of_work_bits", "num_query_rounds"):
            setattr(s, f, int(getattr(c, f)))
        s.num_reductions = len(c.reduction_arity_bits)
        for i, a in enumerate(c.reduction_arity_bits):
            s.reduction_arity_bits[i] = int(a)
        s.num_gates = len(c.gates)
        s.num_public_inputs = int(len(c.public_inputs))
        s.gates = C.cast(self._gates, C.POINTER(_Gate))
        s.k_is = self._k.ctypes.data; s.constants = self._const.ctypes.data; s.sigmas = self._sig.ctypes.data
        self.s = s
        L = lib()
        L.glo_proof_words.restype = C.c_size_t
        if cs_cap is None:
            self.cs_cap = np.empty((1 << c.cap_height, 4), np.uint64)
            with _Hasher(self.hasher):
                L.glo_constants_sigmas_cap(C.byref(s), _p(self.cs_cap))
        else:
            self.cs_cap = _a(cs_cap).reshape(1 << c.cap_height, 4).copy()
        if getattr(c, "circuit_digest", None) is None:
            c.circuit_digest = circuit_digest(self.cs_cap, c.degree_bits, self.hasher)
        for i in range(4):
            s.circuit_digest[i] = int(c.circuit_digest[i])
        self.proof_words = L.glo_proof_words(C.byref(s))

    def prove(self, wires=None, public_inputs=None):
        w = _a(self.c.wires if wires is None else wires)
        pi = _a(self.c.public_inputs if public_inputs is None else public_inputs)
        proof = np.zeros(self.proof_words, np.uint64)
        with _Hasher(self.hasher):
            rc = lib().glo_prove(C.byref(self.s), _p(w), _p(pi) if pi.size else None, _p(proof))
        return rc, proof

    def witness_fill(self, wires, only_advice=False):
        """gl_witness_oracle.c: every row-local generator once, on a copy of `wires` [num_wires][n]."""
        w = _a(wires).copy()
        lib().glo_witness_fill(C.byref(self.s), _p(w), C.c_int(1 if only_advice else 0))
        return w

    def verify(self, proof):
        with _Hasher(self.hasher):
            return lib().glo_verify(C.byref(self.s), _p(self.cs_cap), _p(_a(proof)))

    def proof_to_bytes(self, proof):
        """gl_proof_bytes.c: `ProofWithPublicInputs::to_bytes()` restated along the Rust writer's call tree (recalled, unpinned)."""
        L = lib()
        L.glo_proof_to_bytes.restype = C.c_size_t
        w = _a(proof)
        assert w.size == self.proof_words
        with _Hasher(self.hasher):
            n = L.glo_proof_to_bytes(C.byref(self.s), _p(w), None, C.c_size_t(0))
            out = np.empty(n, np.uint8)
            assert L.glo_proof_to_bytes(C.byref(self.s), _p(w), out.ctypes.data_as(C.c_void_p), C.c_size_t(n)) == n
        return out.tobytes()


def circuit_digest(constants_sigmas_cap, degree_bits, hasher=0):
    """plonk/circuit_builder.rs build(): C::Hasher::hash_no_pad(cap.flatten() ++ hash_pad(domain_separator = []).to_vec() ++ [degree_bits]).
    With KeccakHash<25> a hash "flattens" to its four 7-byte chunks (BytesHash::to_vec) and hash_pad pads to the sponge width 12 as for Poseidon."""
    if hasher == 0:
        ds = hash_pad(np.zeros(0, np.uint64))
        parts = np.concatenate([_a(constants_sigmas_cap).reshape(-1), ds, np.array([degree_bits], np.uint64)])
        return hash_no_pad(parts)
    pad = np.array([1] + [0] * 10 + [1], np.uint64)                      # hash_pad([]): 1, zeros up to width - 1, 1
    ds = keccak_hash_to_elements(keccak_hash_no_pad(pad))
    cap_e = np.concatenate([keccak_hash_to_elements(h) for h in _a(constants_sigmas_cap).reshape(-1, 4)])
    return keccak_hash_no_pad(np.concatenate([cap_e, ds, np.array([degree_bits], np.uint64)]))
