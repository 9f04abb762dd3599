"""ctypes binding of libglprover.so (C ABI: include/glp.h).

Mirrors the plonky2 objects the reference touches on the prove() path:
  Batch            <-> `PolynomialBatch`       (fri/oracle.rs)   from_values / from_coeffs / cap / prove / get
  Context.fft/ifft <-> `PolynomialCoeffs::fft`, `PolynomialValues::ifft`
Errors surface as GlpError carrying glp_last_error(), the analogue of the `anyhow::Error` that
`data.prove(pw)` returns [REF src/ecdsa/gadgets/ecdsa.rs:349].
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np

P = 0xFFFFFFFF00000001
_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SO = os.path.join(_HERE, "libglprover.so")
_lib = None


class GlpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("glp error %d: %s" % (code, msg))
        self.code = code


class _Gate(C.Structure):
    _fields_ = [(f, C.c_uint32) for f in ("type", "selector_index", "group_start", "group_end", "row",
                                          "num_constraints", "p0", "p1")]


class _CircuitDesc(C.Structure):
    _fields_ = ([(f, C.c_uint32) for f in ("degree_bits", "num_wires", "num_routed_wires", "num_constants",
                                           "num_selectors", "num_challenges", "quotient_degree_factor",
                                           "num_partial_products", "num_gate_constraints", "rate_bits", "cap_height",
                                           "proof_of_work_bits", "num_query_rounds", "num_reductions")] +
                [("reduction_arity_bits", C.c_uint32 * 16), ("num_gates", C.c_uint32), ("num_public_inputs", C.c_uint32),
                 ("gates", C.POINTER(_Gate)), ("k_is", C.c_void_p), ("circuit_digest", C.c_uint64 * 4),
                 ("constants", C.c_void_p), ("sigmas", C.c_void_p), ("hasher", C.c_uint32)])


def library_path():
    return _SO


def build_library(force=False):
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return _SO


def exported_symbols():
    """Function names declared in include/glp.h."""
    hdr = open(os.path.join(_ROOT, "include", "glp.h")).read()
    return re.findall(r"GLP_API [^;(]*?(glp_\w+)\(", hdr)


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise GlpError(-4, "libglprover.so is not built (run __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(_SO)
    vp, u32, u64, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t
    L.glp_last_error.restype = C.c_char_p
    L.glp_version.restype = C.c_char_p
    L.glp_ctx_stream.restype = vp
    L.glp_ctx_stream.argtypes = [vp]
    L.glp_batch_num_digests.restype = sz
    L.glp_batch_num_digests.argtypes = [vp]
    sigs = {
        "glp_ctx_create": [C.c_int, C.POINTER(vp)],
        "glp_ctx_destroy": [vp],
        "glp_ctx_synchronize": [vp],
        "glp_ctx_set_profiling": [vp, C.c_int],
        "glp_ctx_stage_reset": [vp],
        "glp_ctx_stage_count": [vp],
        "glp_ctx_stage_get": [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_double)],
        "glp_poseidon_permute": [vp, vp, sz],
        "glp_fill_random_device": [vp, vp, sz, u64],
        "glp_fft": [vp, vp, u32, u32],
        "glp_ifft": [vp, vp, u32, u32],
        "glp_lde": [vp, vp, u32, u32, u32, u64, vp],
        "glp_batch_from_values": [vp, vp, u32, u32, u32, u32, C.POINTER(vp)],
        "glp_batch_from_values_device": [vp, vp, u32, u32, u32, u32, C.POINTER(vp)],
        "glp_batch_from_coeffs": [vp, vp, u32, u32, u32, u32, C.POINTER(vp)],
        "glp_batch_from_coeffs_device": [vp, vp, u32, u32, u32, u32, C.POINTER(vp)],
        "glp_batch_from_values_h": [vp, vp, u32, u32, u32, u32, u32, C.POINTER(vp)],
        "glp_batch_from_coeffs_h": [vp, vp, u32, u32, u32, u32, u32, C.POINTER(vp)],
        "glp_keccak256": [vp, vp, sz, sz, vp],
        "glp_batch_free": [vp],
        "glp_batch_info": [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)],
        "glp_batch_cap": [vp, vp],
        "glp_batch_coeffs": [vp, u32, u32, vp],
        "glp_batch_leaf": [vp, u64, vp],
        "glp_batch_merkle_proof": [vp, u64, vp],
        "glp_batch_digests": [vp, vp],
    }
    L.glp_proof_words.restype = sz
    L.glp_proof_words.argtypes = [vp]
    L.glp_proof_bytes_len.restype = sz
    L.glp_proof_bytes_len.argtypes = [vp]
    sigs.update({
        "glp_circuit_create": [vp, C.POINTER(_CircuitDesc), C.POINTER(vp)],
        "glp_circuit_free": [vp],
        "glp_circuit_digest": [vp, vp],
        "glp_circuit_constants_sigmas_cap": [vp, vp],
        "glp_prove": [vp, vp, vp, vp, vp],
        "glp_proof_to_bytes": [vp, vp, vp, sz],
        "glp_proof_from_bytes": [vp, vp, sz, vp],
        "glp_prove_device": [vp, vp, vp, vp, vp],
        "glp_session_begin": [vp, vp, vp, C.c_int, vp, C.POINTER(vp), vp, vp],
        "glp_session_partial_products": [vp, vp, vp, vp],
        "glp_session_quotient": [vp, vp, vp],
        "glp_session_open": [vp, vp, vp],
        "glp_session_fri_combine": [vp, vp],
        "glp_session_fri_commit": [vp, vp],
        "glp_session_fri_fold": [vp, vp],
        "glp_session_fri_final_poly": [vp, vp],
        "glp_pow_search": [vp, vp, vp, C.c_uint32, C.c_uint32, vp],
        "glp_pow_search_h": [vp, C.c_uint32, vp, vp, C.c_uint32, C.c_uint32, vp],
        "glp_session_queries": [vp, C.c_uint64, vp, C.c_uint32],
        "glp_session_proof": [vp, vp],
        "glp_session_end": [vp],
        "glp_verify": [vp, vp],
        "glp_verify_n": [vp, vp, sz],
        "glp_prove_batch": [vp, vp, u32, vp, C.c_int, vp, vp],
        "glp_verify_batch": [vp, vp, u32, vp, vp, vp],
        "glp_witness_fill": [vp, vp, vp, C.c_int],
        "glp_witness_columns": [vp, u32, vp],
        "glp_host_alloc": [vp, sz, C.POINTER(vp)],
        "glp_host_free": [vp, vp],
        "glp_witness_stage": [vp, vp, vp, u32, C.POINTER(vp)],
        "glp_prove_staged": [vp, vp, vp, vp, vp],
        "glp_witness_free": [vp],
        "glp_dev_alloc": [vp, sz, C.POINTER(vp)],
        "glp_dev_free": [vp, vp],
        "glp_dev_upload": [vp, vp, vp, sz],
        "glp_dev_download": [vp, vp, vp, sz],
    })
    sigs.update({
        "glp_circuit_file_write": [C.c_char_p, C.POINTER(_CircuitDesc), vp, vp],
        "glp_circuit_file_open": [C.c_char_p, C.c_int, C.POINTER(vp)],
        "glp_circuit_file_close": [vp],
    })
    L.glp_circuit_file_desc.restype = C.POINTER(_CircuitDesc)
    L.glp_circuit_file_desc.argtypes = [vp]
    for name in ("glp_circuit_file_wires", "glp_circuit_file_public_inputs"):
        getattr(L, name).restype = vp
        getattr(L, name).argtypes = [vp]
    for name in ("glp_num_openings", "glp_final_poly_len"):
        getattr(L, name).restype = sz
        getattr(L, name).argtypes = [vp]
    for name, argtypes in sigs.items():
        getattr(L, name).argtypes = argtypes
    L.glp_ctx_destroy.restype = None
    L.glp_batch_free.restype = None
    L.glp_circuit_free.restype = None
    L.glp_session_end.restype = None
    L.glp_circuit_file_close.restype = None
    L.glp_witness_free.restype = None
    _lib = L
    return L


def splitmix_field(seed, count, offset=0):
    """numpy twin of glp_fill_random_device (same values for the same seed)."""
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
        return np.where(z >= np.uint64(P), z - np.uint64(P), z)


def _chk(rc):
    if rc != 0:
        raise GlpError(rc, load_library().glp_last_error().decode())


def _a(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU + one HIP stream (glp_ctx)."""

    def __init__(self, device=0):
        L = load_library()
        self._h = C.c_void_p()
        _chk(L.glp_ctx_create(int(device), C.byref(self._h)))
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            load_library().glp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def synchronize(self):
        _chk(load_library().glp_ctx_synchronize(self._h))

    @property
    def stream(self):
        return load_library().glp_ctx_stream(self._h)

    # -- stage timers
    def set_profiling(self, on=True):
        _chk(load_library().glp_ctx_set_profiling(self._h, 1 if on else 0))

    def stage_reset(self):
        _chk(load_library().glp_ctx_stage_reset(self._h))

    def stages(self):
        L = load_library()
        out = []
        for i in range(L.glp_ctx_stage_count(self._h)):
            name, ms, by = C.c_char_p(), C.c_float(), C.c_double()
            _chk(L.glp_ctx_stage_get(self._h, i, C.byref(name), C.byref(ms), C.byref(by)))
            out.append((name.value.decode(), ms.value, by.value))
        return out

    # -- primitives
    def poseidon_permute(self, states):
        s = _a(states).reshape(-1, 12).copy()
        _chk(load_library().glp_poseidon_permute(self._h, _p(s), s.shape[0]))
        return s

    def fft(self, cols):
        a = np.atleast_2d(_a(cols)).copy()
        _chk(load_library().glp_fft(self._h, _p(a), a.shape[0], int(a.shape[1]).bit_length() - 1))
        return a

    def ifft(self, cols):
        a = np.atleast_2d(_a(cols)).copy()
        _chk(load_library().glp_ifft(self._h, _p(a), a.shape[0], int(a.shape[1]).bit_length() - 1))
        return a

    def lde(self, coeffs, rate_bits=3, shift=7):
        a = np.atleast_2d(_a(coeffs))
        out = np.empty((a.shape[0], a.shape[1] << rate_bits), np.uint64)
        _chk(load_library().glp_lde(self._h, _p(a), a.shape[0], int(a.shape[1]).bit_length() - 1, rate_bits, shift, _p(out)))
        return out

    def dev_alloc(self, nbytes):
        """Device buffer from the context's pool (for the *_device entry points); returns the pointer as an int."""
        p = C.c_void_p()
        _chk(load_library().glp_dev_alloc(self._h, int(nbytes), C.byref(p)))
        return p.value

    def dev_free(self, ptr):
        _chk(load_library().glp_dev_free(self._h, C.c_void_p(ptr)))

    def dev_upload(self, ptr, array):
        a = np.ascontiguousarray(array)
        _chk(load_library().glp_dev_upload(self._h, C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def dev_download(self, ptr, array):
        _chk(load_library().glp_dev_download(self._h, array.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), array.nbytes))

    def host_alloc(self, shape):
        """Page-locked host memory (glp_host_alloc) as a uint64 numpy array of `shape`: what witness generation should write into
        so that glp_witness_stage's copy overlaps the proof in flight.  Free with host_free(array)."""
        n = int(np.prod(shape))
        p = C.c_void_p()
        _chk(load_library().glp_host_alloc(self._h, max(n, 1) * 8, C.byref(p)))
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(max(n, 1),))[:n].reshape(shape)
        self.__dict__.setdefault("_pinned", {})[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        p = self.__dict__.get("_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None and getattr(self, "_h", None):
            _chk(load_library().glp_host_free(self._h, C.c_void_p(p)))

    def fill_random_device(self, dev_ptr, count, seed):
        _chk(load_library().glp_fill_random_device(self._h, C.c_void_p(dev_ptr), count, seed))

    # -- PolynomialBatch
    def batch_from_values(self, values, rate_bits=3, cap_height=4, hasher=0):
        return Batch._make(self, "glp_batch_from_values_h", values, rate_bits, cap_height, hasher)

    def batch_from_coeffs(self, coeffs, rate_bits=3, cap_height=4, hasher=0):
        return Batch._make(self, "glp_batch_from_coeffs_h", coeffs, rate_bits, cap_height, hasher)

    def keccak256(self, msgs):
        """Keccak-256 of equal-length byte strings on the GPU (glp_keccak256): list of bytes -> list of 32-byte digests."""
        msgs = [bytes(m) for m in msgs]
        if not msgs:
            return []
        n = len(msgs[0])
        if any(len(m) != n for m in msgs):
            raise GlpError(-1, "glp_keccak256 takes messages of one length")
        buf = np.frombuffer(b"".join(msgs) or b"\0", dtype=np.uint8)
        out = np.zeros((len(msgs), 32), np.uint8)
        _chk(load_library().glp_keccak256(self._h, buf.ctypes.data_as(C.c_void_p), len(msgs), n, out.ctypes.data_as(C.c_void_p)))
        return [bytes(r) for r in out]

    def batch_from_values_device(self, dev_ptr, ncols, log_n, rate_bits=3, cap_height=4):
        return Batch._make_dev(self, "glp_batch_from_values_device", dev_ptr, ncols, log_n, rate_bits, cap_height)

    def batch_from_coeffs_device(self, dev_ptr, ncols, log_n, rate_bits=3, cap_height=4):
        return Batch._make_dev(self, "glp_batch_from_coeffs_device", dev_ptr, ncols, log_n, rate_bits, cap_height)


class Batch:
    """plonky2 `PolynomialBatch` resident on the GPU."""

    def __init__(self, ctx, handle, ncols, log_n, rate_bits, cap_height):
        self.ctx, self._h = ctx, handle
        self.ncols, self.log_n, self.rate_bits, self.cap_height = ncols, log_n, rate_bits, cap_height

    @classmethod
    def _make(cls, ctx, fn, arr, rate_bits, cap_height, hasher=0):
        a = _a(arr)
        if a.ndim != 2:
            raise GlpError(-1, "expected a [ncols][n] array")
        ncols, n = a.shape
        if n & (n - 1) or n == 0:
            raise GlpError(-1, "n must be a power of two")
        h = C.c_void_p()
        _chk(getattr(load_library(), fn)(ctx._h, _p(a), ncols, n.bit_length() - 1, rate_bits, cap_height, int(hasher), C.byref(h)))
        return cls(ctx, h, ncols, n.bit_length() - 1, rate_bits, cap_height)

    @classmethod
    def _make_dev(cls, ctx, fn, dev_ptr, ncols, log_n, rate_bits, cap_height):
        h = C.c_void_p()
        _chk(getattr(load_library(), fn)(ctx._h, C.c_void_p(dev_ptr), ncols, log_n, rate_bits, cap_height, C.byref(h)))
        return cls(ctx, h, ncols, log_n, rate_bits, cap_height)

    def free(self):
        if self._h:
            if getattr(self.ctx, "_h", None):                # never touch a handle whose context is already gone
                load_library().glp_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @property
    def num_leaves(self):
        return 1 << (self.log_n + self.rate_bits)

    def cap(self):
        out = np.empty((1 << self.cap_height, 4), np.uint64)
        _chk(load_library().glp_batch_cap(self._h, _p(out)))
        return out

    def coeffs(self, col_begin=0, ncols=None):
        ncols = self.ncols - col_begin if ncols is None else ncols
        out = np.empty((ncols, 1 << self.log_n), np.uint64)
        _chk(load_library().glp_batch_coeffs(self._h, col_begin, ncols, _p(out)))
        return out

    def leaf(self, index):
        out = np.empty(self.ncols, np.uint64)
        _chk(load_library().glp_batch_leaf(self._h, int(index), _p(out)))
        return out

    def prove(self, index):
        depth = self.log_n + self.rate_bits - self.cap_height
        out = np.empty((depth, 4), np.uint64)
        _chk(load_library().glp_batch_merkle_proof(self._h, int(index), _p(out)))
        return out

    def digests(self):
        n = load_library().glp_batch_num_digests(self._h)
        out = np.empty((n, 4), np.uint64)
        _chk(load_library().glp_batch_digests(self._h, _p(out)))
        return out


def _desc_to_c(desc):
    """synth.Circuit-like attribute bag -> (glp_circuit_desc, objects that must stay alive while it is used)."""
    gates = (_Gate * len(desc.gates))()
    for i, g in enumerate(desc.gates):
        for f, _ in _Gate._fields_:
            setattr(gates[i], f, int(g[f]))
    k, const, sig = _a(desc.k_is), _a(desc.constants), _a(desc.sigmas)
    # the C ABI takes bare pointers: every array's size is checked here so that a wrong shape is a GlpError, not a
    # host over-read
    n = 1 << int(desc.degree_bits)
    if k.size != int(desc.num_routed_wires):
        raise GlpError(-1, "k_is has %d entries, num_routed_wires is %d" % (k.size, desc.num_routed_wires))
    if const.size != int(desc.num_constants) * n:
        raise GlpError(-1, "constants has %d elements, expected num_constants * 2^degree_bits = %d" % (const.size, int(desc.num_constants) * n))
    if sig.size != int(desc.num_routed_wires) * n:
        raise GlpError(-1, "sigmas has %d elements, expected num_routed_wires * 2^degree_bits = %d" % (sig.size, int(desc.num_routed_wires) * n))
    if len(desc.reduction_arity_bits) > 16:
        raise GlpError(-1, "more than 16 FRI reductions")
    d = _CircuitDesc()
    for f in ("degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_selectors", "num_challenges",
              "quotient_degree_factor", "num_partial_products", "num_gate_constraints", "rate_bits", "cap_height",
              "proof_of_work_bits", "num_query_rounds"):
        setattr(d, f, int(getattr(desc, f)))
    d.num_reductions = len(desc.reduction_arity_bits)
    for i, ab in enumerate(desc.reduction_arity_bits):
        d.reduction_arity_bits[i] = int(ab)
    d.num_gates, d.num_public_inputs = len(desc.gates), int(len(desc.public_inputs))
    d.gates = C.cast(gates, C.POINTER(_Gate))
    d.k_is, d.constants, d.sigmas = k.ctypes.data, const.ctypes.data, sig.ctypes.data
    dig = getattr(desc, "circuit_digest", None)
    if dig is not None:
        for i in range(4):
            d.circuit_digest[i] = int(dig[i])
    d.hasher = int(getattr(desc, "hasher", 0))
    return d, (gates, k, const, sig)


def write_circuit_file(path, desc, with_witness=True):
    """glp_circuit_file_write: the hand-off file a machine with the Rust builder produces (include/glp.h)."""
    d, keep = _desc_to_c(desc)
    w = pi = None
    if with_witness:
        w, pi = _a(desc.wires), _a(desc.public_inputs)
        if w.size != int(desc.num_wires) << int(desc.degree_bits):
            raise GlpError(-1, "wires has %d elements, expected num_wires * 2^degree_bits" % w.size)
    _chk(load_library().glp_circuit_file_write(os.fsencode(path), C.byref(d), _p(w) if w is not None else None,
                                               _p(pi) if (pi is not None and pi.size) else None))
    del keep


class _FileDesc:
    pass


class CircuitFile:
    """A mapped circuit hand-off file; `.desc` is an attribute bag `Circuit(ctx, cf.desc)` accepts (numpy views into the
    mapping, valid until close())."""

    def __init__(self, path, verify_checksum=True):
        L = load_library()
        self._h = C.c_void_p()
        _chk(L.glp_circuit_file_open(os.fsencode(path), 1 if verify_checksum else 0, C.byref(self._h)))
        cd = L.glp_circuit_file_desc(self._h).contents
        d = _FileDesc()
        for f in ("degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_selectors", "num_challenges",
                  "quotient_degree_factor", "num_partial_products", "num_gate_constraints", "rate_bits", "cap_height",
                  "proof_of_work_bits", "num_query_rounds"):
            setattr(d, f, int(getattr(cd, f)))
        d.reduction_arity_bits = [int(cd.reduction_arity_bits[i]) for i in range(cd.num_reductions)]
        d.gates = [{f: int(getattr(cd.gates[i], f)) for f, _ in _Gate._fields_} for i in range(cd.num_gates)]
        n = 1 << d.degree_bits

        def view(ptr, shape):
            cnt = int(np.prod(shape))
            if not ptr or cnt == 0:
                return np.zeros(shape, np.uint64)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(cnt,)).reshape(shape)
        d.k_is = view(cd.k_is, (d.num_routed_wires,))
        d.constants = view(cd.constants, (d.num_constants, n))
        d.sigmas = view(cd.sigmas, (d.num_routed_wires, n))
        dig = [int(cd.circuit_digest[i]) for i in range(4)]
        d.circuit_digest = np.array(dig, np.uint64) if any(dig) else None
        d.hasher = int(cd.hasher)
        wp = L.glp_circuit_file_wires(self._h)
        self.has_witness = bool(wp)
        d.wires = view(wp, (d.num_wires, n)) if wp else None
        d.public_inputs = view(L.glp_circuit_file_public_inputs(self._h), (int(cd.num_public_inputs),)) if wp else \
            np.zeros(int(cd.num_public_inputs), np.uint64)
        self.desc = d

    def close(self):
        if getattr(self, "_h", None):
            self.desc = None
            load_library().glp_circuit_file_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Circuit:
    """Device-resident circuit data: what `builder.build::<C>()` hands the prover
    [REF src/ecdsa/gadgets/ecdsa.rs:298].  `desc` is an attribute bag like synth.Circuit."""

    def __init__(self, ctx, desc):
        L = load_library()
        self.ctx, self.desc = ctx, desc
        d, keep = _desc_to_c(desc)
        n = 1 << int(desc.degree_bits)
        self._wire_elems = int(desc.num_wires) * n
        self._num_pis = int(len(desc.public_inputs))
        self._h = C.c_void_p()
        _chk(L.glp_circuit_create(ctx._h, C.byref(d), C.byref(self._h)))
        del keep
        self.proof_words = L.glp_proof_words(self._h)

    def free(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):                # never touch a handle whose context is already gone
                load_library().glp_circuit_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def digest(self):
        out = np.empty(4, np.uint64)
        _chk(load_library().glp_circuit_digest(self._h, _p(out)))
        return out

    def constants_sigmas_cap(self):
        out = np.empty((1 << self.desc.cap_height, 4), np.uint64)
        _chk(load_library().glp_circuit_constants_sigmas_cap(self._h, _p(out)))
        return out

    def prove(self, wires=None, public_inputs=None):
        """`data.prove(pw)` after witness generation: full witness in, proof words out (include/glp.h)."""
        w = _a(self.desc.wires if wires is None else wires)
        pi = _a(self.desc.public_inputs if public_inputs is None else public_inputs)
        self._check_witness(w, pi)
        proof = np.zeros(self.proof_words, np.uint64)
        _chk(load_library().glp_prove(self.ctx._h, self._h, _p(w), _p(pi) if pi.size else None, _p(proof)))
        return proof

    def _check_witness(self, w, pi):
        if w is not None and w.size != self._wire_elems:
            raise GlpError(-1, "wires has %d elements, expected num_wires * 2^degree_bits = %d" % (w.size, self._wire_elems))
        if pi.size != self._num_pis:
            raise GlpError(-1, "%d public inputs, the circuit has %d" % (pi.size, self._num_pis))

    def _check_proof(self, proof_words):
        a = _a(proof_words)
        if a.size != self.proof_words:
            raise GlpError(-1, "proof has %d words, a proof of this circuit has %d" % (a.size, self.proof_words))
        return a

    def verify(self, proof_words):
        """`data.verify(proof)`: True if accepted; raises nothing on rejection (reason: `last_error()`)."""
        L = load_library()
        a = _a(proof_words)
        rc = L.glp_verify_n(self._h, _p(a), a.size)
        if rc == 0:
            return True
        if rc == -5:          # GLP_ERR_PROVE: a well-formed call, the proof is rejected
            return False
        _chk(rc)

    def verify_batch(self, proofs, reasons=False):
        """glp_verify_batch: proofs [K][proof_words] -> bool array [K] (and the list of rejection reasons if asked); the query
        rounds of all proofs run in one launch on the GPU."""
        a = _a(proofs)
        if a.ndim != 2 or a.shape[1] != self.proof_words:
            raise GlpError(-1, "proofs must be [K][proof_words]")
        K = a.shape[0]
        status = np.zeros(K, np.int32)
        buf = C.create_string_buffer(K * 160) if reasons else None
        _chk(load_library().glp_verify_batch(self.ctx._h, self._h, K, _p(a), status.ctypes.data_as(C.c_void_p), buf))
        ok = status == 0
        if reasons:
            return ok, [buf.raw[160 * k:160 * (k + 1)].split(b"\0", 1)[0].decode() for k in range(K)]
        return ok

    def proof_to_bytes(self, proof_words):
        """`ProofWithPublicInputs::to_bytes()`."""
        L = load_library()
        n = L.glp_proof_bytes_len(self._h)
        out = np.empty(n, np.uint8)
        _chk(L.glp_proof_to_bytes(self._h, _p(self._check_proof(proof_words)), out.ctypes.data_as(C.c_void_p), n))
        return out.tobytes()

    def proof_from_bytes(self, data):
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        words = np.zeros(self.proof_words, np.uint64)
        _chk(load_library().glp_proof_from_bytes(self._h, buf.ctypes.data_as(C.c_void_p), buf.size, _p(words)))
        return words

    def prove_batch(self, wires, public_inputs=None, out=None):
        """glp_prove_batch: wires [K][num_wires][n] (host array) -> proofs [K][proof_words] (into `out` if given: the library
        writes every word, so a caller that proves batch after batch can reuse one buffer)."""
        w = _a(wires)
        if w.ndim != 3 or w[0].size != self._wire_elems:
            raise GlpError(-1, "wires must be [K][num_wires][2^degree_bits]")
        K = w.shape[0]
        pi = _a(np.zeros((K, 0), np.uint64) if public_inputs is None else public_inputs).reshape(K, -1)
        if pi.shape[1] != self._num_pis:
            raise GlpError(-1, "%d public inputs per proof, the circuit has %d" % (pi.shape[1], self._num_pis))
        if out is None:
            out = np.empty((K, self.proof_words), np.uint64)
        elif out.dtype != np.uint64 or out.shape != (K, self.proof_words) or not out.flags.c_contiguous:
            raise GlpError(-1, "out must be a C-contiguous uint64 array [K][proof_words]")
        _chk(load_library().glp_prove_batch(self.ctx._h, self._h, K, _p(w), 0, _p(pi) if pi.size else None, _p(out)))
        return out

    def prove_batch_device(self, dev_wires_ptr, K, public_inputs=None):
        pi = _a(np.zeros((K, 0), np.uint64) if public_inputs is None else public_inputs).reshape(K, -1)
        if pi.shape[1] != self._num_pis:
            raise GlpError(-1, "%d public inputs per proof, the circuit has %d" % (pi.shape[1], self._num_pis))
        out = np.empty((K, self.proof_words), np.uint64)
        _chk(load_library().glp_prove_batch(self.ctx._h, self._h, K, C.c_void_p(dev_wires_ptr), 1, _p(pi) if pi.size else None, _p(out)))
        return out

    def witness_fill(self, dev_wires_ptr, only_advice=False):
        """Row-local witness generation in place on an HBM-resident witness (include/glp.h, glp_witness_fill)."""
        _chk(load_library().glp_witness_fill(self.ctx._h, self._h, C.c_void_p(dev_wires_ptr), 1 if only_advice else 0))

    def witness_columns(self, gate_index):
        """uint8 [num_wires]: 1 = written by witness_fill on rows of that gate, 2 = read as generator input, 0 = untouched."""
        out = np.zeros(int(self.desc.num_wires), np.uint8)
        _chk(load_library().glp_witness_columns(self._h, int(gate_index), out.ctypes.data_as(C.c_void_p)))
        return out

    def stage_witness(self, host_wires, routed_only=False):
        """glp_witness_stage: start uploading a witness on the context's copy stream ([num_wires][n], or with routed_only the routed
        columns [num_routed_wires][n] -- the advice columns are then derived on the GPU); returns a StagedWitness for prove_staged."""
        w = host_wires if (isinstance(host_wires, np.ndarray) and host_wires.dtype == np.uint64 and host_wires.flags.c_contiguous) else _a(host_wires)
        n = 1 << int(self.desc.degree_bits)
        need = (int(self.desc.num_routed_wires) if routed_only else int(self.desc.num_wires)) * n
        if w.size < need or (not routed_only and w.size != need):
            raise GlpError(-1, "staged witness has %d elements, expected %d" % (w.size, need))
        h = C.c_void_p()
        _chk(load_library().glp_witness_stage(self.ctx._h, self._h, _p(w), 1 if routed_only else 0, C.byref(h)))
        return StagedWitness(self, h, w)

    def prove_staged(self, staged, public_inputs=None):
        pi = _a(self.desc.public_inputs if public_inputs is None else public_inputs)
        self._check_witness(None, pi)
        proof = np.zeros(self.proof_words, np.uint64)
        _chk(load_library().glp_prove_staged(self.ctx._h, self._h, staged._h, _p(pi) if pi.size else None, _p(proof)))
        return proof

    def prove_device(self, dev_wires_ptr, public_inputs=None):
        pi = _a(self.desc.public_inputs if public_inputs is None else public_inputs)
        self._check_witness(None, pi)
        proof = np.zeros(self.proof_words, np.uint64)
        _chk(load_library().glp_prove_device(self.ctx._h, self._h, C.c_void_p(dev_wires_ptr), _p(pi) if pi.size else None,
                                             _p(proof)))
        return proof


class StagedWitness:
    """A witness on its way to (or in) HBM (glp_witness): keeps the host array alive until the upload has been consumed."""

    def __init__(self, circuit, handle, host_array):
        self.circuit, self._h, self._host = circuit, handle, host_array

    def free(self):
        if getattr(self, "_h", None):
            if getattr(self.circuit.ctx, "_h", None):
                load_library().glp_witness_free(self._h)
            self._h = None
            self._host = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Session:
    """One proof stepped by the caller's transcript (include/glp.h, glp_session_*): every method returns what the
    Rust prover's `Challenger` observes next and takes the challenges it draws next
    [UPSTREAM plonky2 plonk/prover.rs `prove_with_partition_witness`; reached from REF src/ecdsa/gadgets/ecdsa.rs:349]."""

    def __init__(self, circuit, wires=None, public_inputs=None, dev_wires_ptr=None):
        L = load_library()
        self.circuit, d = circuit, circuit.desc
        self._capn = 1 << d.cap_height
        pi = _a(d.public_inputs if public_inputs is None else public_inputs)
        self._h = C.c_void_p()
        self.wires_cap = np.empty((self._capn, 4), np.uint64)
        self.public_inputs_hash = np.empty(4, np.uint64)
        if dev_wires_ptr is not None:
            circuit._check_witness(None, pi)
            wp, on_dev = C.c_void_p(dev_wires_ptr), 1
        else:
            self._w = _a(d.wires if wires is None else wires)     # keep the host array alive
            circuit._check_witness(self._w, pi)
            wp, on_dev = _p(self._w), 0
        _chk(L.glp_session_begin(circuit.ctx._h, circuit._h, wp, on_dev, _p(pi) if pi.size else None, C.byref(self._h),
                                 _p(self.wires_cap), _p(self.public_inputs_hash)))

    def _cap(self, fn, *args):
        out = np.empty((self._capn, 4), np.uint64)
        _chk(fn(self._h, *args, _p(out)))
        return out

    def partial_products(self, betas, gammas):
        self._b, self._g = _a(betas), _a(gammas)
        return self._cap(load_library().glp_session_partial_products, _p(self._b), _p(self._g))

    def quotient(self, alphas):
        self._al = _a(alphas)
        return self._cap(load_library().glp_session_quotient, _p(self._al))

    def open(self, zeta):
        L = load_library()
        out = np.empty((L.glp_num_openings(self.circuit._h), 2), np.uint64)
        _chk(L.glp_session_open(self._h, _p(_a(zeta)), _p(out)))
        return out

    def fri_combine(self, alpha):
        _chk(load_library().glp_session_fri_combine(self._h, _p(_a(alpha))))

    def fri_commit(self):
        return self._cap(load_library().glp_session_fri_commit)

    def fri_fold(self, beta):
        _chk(load_library().glp_session_fri_fold(self._h, _p(_a(beta))))

    def fri_final_poly(self):
        L = load_library()
        out = np.empty((L.glp_final_poly_len(self.circuit._h), 2), np.uint64)
        _chk(L.glp_session_fri_final_poly(self._h, _p(out)))
        return out

    def pow_search(self, sponge_state, pending_inputs, bits):
        st, pend = _a(sponge_state), _a(pending_inputs)
        w = C.c_uint64()
        _chk(load_library().glp_pow_search_h(self.circuit.ctx._h, int(getattr(self.circuit.desc, "hasher", 0)), _p(st),
                                             _p(pend) if pend.size else None, pend.size, int(bits), C.byref(w)))
        return int(w.value)

    def queries(self, pow_witness, indices):
        idx = _a(indices)
        _chk(load_library().glp_session_queries(self._h, C.c_uint64(int(pow_witness)), _p(idx), idx.size))

    def proof(self):
        out = np.zeros(self.circuit.proof_words, np.uint64)
        _chk(load_library().glp_session_proof(self._h, _p(out)))
        return out

    def end(self):
        if getattr(self, "_h", None):
            if getattr(self.circuit.ctx, "_h", None):        # a session must not outlive its context: then it is only dropped
                load_library().glp_session_end(self._h)
            self._h = None

    def __del__(self):
        try:
            self.end()
        except Exception:
            pass
