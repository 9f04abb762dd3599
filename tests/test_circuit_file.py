"""Circuit hand-off file (include/glp.h glp_circuit_file_*): host-only code, so these run without
a GPU.  Round trip for every circuit family, the committed sample file, and the error behaviour on damaged files."""
import os
import struct

import numpy as np
import pytest

import plonky2_lib_amd as glp
import plonky2_lib_amd.synth as synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
SCALARS = ("degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_selectors", "num_challenges",
           "quotient_degree_factor", "num_partial_products", "num_gate_constraints", "rate_bits", "cap_height",
           "proof_of_work_bits", "num_query_rounds")


def _same(a, b, with_witness=True):
    for f in SCALARS:
        assert int(getattr(a, f)) == int(getattr(b, f)), f
    assert list(a.reduction_arity_bits) == list(b.reduction_arity_bits)
    assert [{k: int(v) for k, v in g.items()} for g in a.gates] == [{k: int(g[k]) for k in a.gates[0]} for g in b.gates]
    assert (np.asarray(a.k_is) == np.asarray(b.k_is)).all()
    assert (a.constants == b.constants).all() and (a.sigmas == b.sigmas).all()
    if with_witness:
        assert (a.wires == b.wires).all() and (np.asarray(a.public_inputs) == np.asarray(b.public_inputs)).all()


@pytest.mark.parametrize("make", [lambda: synth.ecdsa_shape_circuit(7), lambda: synth.keccak_shape_circuit(6), lambda: synth.smt_shape_circuit(5),
                                  lambda: synth.zkdsa_circuit(3), lambda: synth.arith_circuit(9, synth.Config(135, 80, arity_bits=1, final_poly_bits=3))])
def test_round_trip(tmp_path, make):
    desc = make()
    path = str(tmp_path / "c.glpc")
    glp.write_circuit_file(path, desc)
    n = 1 << desc.degree_bits
    expect = 192 + 32 * len(desc.gates) + 8 * (desc.num_routed_wires + (desc.num_constants + desc.num_routed_wires + desc.num_wires) * n +
                                              len(desc.public_inputs))
    assert os.path.getsize(path) == expect
    with glp.CircuitFile(path) as cf:
        assert cf.has_witness
        _same(cf.desc, desc)
    glp.write_circuit_file(path, desc, with_witness=False)          # circuit only: what `build()` alone produces
    with glp.CircuitFile(path) as cf:
        assert not cf.has_witness and cf.desc.wires is None
        _same(cf.desc, desc, with_witness=False)


def test_hasher_field(tmp_path):
    """Header word 148: GenericConfig::Hasher of the circuit (0 Poseidon, 1 KeccakHash<25>); anything else is a malformed header."""
    import struct
    desc = synth.zkdsa_circuit(3)
    path = str(tmp_path / "h.glpc")
    glp.write_circuit_file(path, desc)
    raw = bytearray(open(path, "rb").read())
    assert struct.unpack_from("<I", raw, 148)[0] == 0
    with glp.CircuitFile(path) as cf:
        assert cf.desc.hasher == 0
    desc.hasher = 1
    glp.write_circuit_file(path, desc)
    raw = bytearray(open(path, "rb").read())
    assert struct.unpack_from("<I", raw, 148)[0] == 1
    with glp.CircuitFile(path) as cf:
        assert cf.desc.hasher == 1
    struct.pack_into("<I", raw, 148, 7)
    open(path, "wb").write(raw)
    with pytest.raises(glp.GlpError):
        glp.CircuitFile(path)


def test_committed_sample_file():
    """tests/golden/zkdsa_2_3.glpc (written by make_golden.py): the reference's simple-signature circuit
    [REF src/zkdsa/circuits/mod.rs:24-43] with its witness; the header fields are checked byte by byte against the layout
    documented in plonky2-lib_amd/csrc/circuit_file.hip."""
    path = os.path.join(GOLDEN, "zkdsa_2_3.glpc")
    raw = open(path, "rb").read()
    assert raw[:8] == b"GLPCIRC1"
    version, header_bytes = struct.unpack_from("<II", raw, 8)
    assert (version, header_bytes) == (1, 192)
    sc = struct.unpack_from("<14I", raw, 16)
    desc = synth.zkdsa_circuit(3)
    assert sc[:4] == (3, 135, 80, desc.num_constants) and sc[9:13] == (3, 4, 16, 28)
    num_gates, num_pis, has_w = struct.unpack_from("<III", raw, 136)
    assert (num_gates, num_pis, has_w) == (len(desc.gates), 12, 1)
    first_gate = struct.unpack_from("<8I", raw, 192)
    assert first_gate == tuple(desc.gates[0][k] for k in ("type", "selector_index", "group_start", "group_end", "row", "num_constraints", "p0", "p1"))
    # FNV-1a 64 over everything after the header
    h = 0xcbf29ce484222325
    for b in raw[192:]:
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    assert struct.unpack_from("<Q", raw, 184)[0] == h
    with glp.CircuitFile(path) as cf:
        _same(cf.desc, desc)
        g = np.load(os.path.join(GOLDEN, "proof_zkdsa_2_3.npz"))
        assert (cf.desc.public_inputs == g["public_inputs"]).all()


def test_committed_sample_file_current_version():
    """tests/golden/zkdsa_2_3_v2.glpc: the same circuit and witness in the version the library writes now -- the checksum covers the
    184 header bytes in front of it as well as the sections (a header field changed in transit is then an error too)."""
    raw = open(os.path.join(GOLDEN, "zkdsa_2_3_v2.glpc"), "rb").read()
    old = open(os.path.join(GOLDEN, "zkdsa_2_3.glpc"), "rb").read()
    assert struct.unpack_from("<II", raw, 8) == (2, 192) and raw[192:] == old[192:] and raw[12:184] == old[12:184]
    h = 0xcbf29ce484222325
    for b in raw[:184] + raw[192:]:
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    assert struct.unpack_from("<Q", raw, 184)[0] == h
    with glp.CircuitFile(os.path.join(GOLDEN, "zkdsa_2_3_v2.glpc")) as cf:
        _same(cf.desc, synth.zkdsa_circuit(3))


def test_damaged_files_are_errors(tmp_path):
    desc = synth.arith_circuit(5, synth.Config.standard_recursion_config(), seed=2)
    path = str(tmp_path / "c.glpc")
    glp.write_circuit_file(path, desc)
    raw = bytearray(open(path, "rb").read())

    def expect_error(data, what):
        p = str(tmp_path / "bad.glpc")
        open(p, "wb").write(bytes(data))
        with pytest.raises(glp.GlpError) as e:
            glp.CircuitFile(p)
        assert what in str(e.value), str(e.value)
    bad = bytearray(raw); bad[5000] ^= 0x40
    expect_error(bad, "checksum")
    expect_error(raw[:-8], "truncated")
    expect_error(raw + b"\0" * 8, "truncated or padded")
    bad = bytearray(raw); bad[0] = ord("X")
    expect_error(bad, "magic")
    bad = bytearray(raw); bad[8] = 3
    expect_error(bad, "version")
    assert struct.unpack_from("<I", raw, 8)[0] == 2                   # the writer's version: its checksum covers the header too
    bad = bytearray(raw); struct.pack_into("<I", bad, 60, 27)          # num_query_rounds 28 -> 27: only the header changes
    expect_error(bad, "checksum")
    # a word >= p in a field section is an error even with a consistent checksum (the kernels assume canonical inputs)
    bad = bytearray(raw)
    off = 192 + 32 * len(desc.gates) + 8 * desc.num_routed_wires + 8 * 77          # constants[0][77]
    struct.pack_into("<Q", bad, off, glp.P + 5)
    h = 0xcbf29ce484222325
    for b in bytes(bad[:184]) + bytes(bad[192:]):
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    struct.pack_into("<Q", bad, 184, h)
    expect_error(bad, "constants[77]")
    bad = bytearray(raw); struct.pack_into("<I", bad, 16, 40)        # degree_bits = 40
    expect_error(bad, "out of range")
    expect_error(raw[:100], "shorter than")
    with pytest.raises(glp.GlpError):
        glp.CircuitFile(str(tmp_path / "missing.glpc"))
    # a flipped byte is still readable when the caller waives the checksum (mapping a multi-GB file without a full read)
    bad = bytearray(raw); bad[5000] ^= 0x40
    p = str(tmp_path / "nock.glpc")
    open(p, "wb").write(bytes(bad))
    with glp.CircuitFile(p, verify_checksum=False) as cf:
        assert cf.desc.degree_bits == 5
    with pytest.raises(glp.GlpError):
        glp.write_circuit_file(str(tmp_path / "nodir" / "x.glpc"), desc)
