"""Mathematical cross-checks of the oracle's field / FFT / Merkle / batch code against definitions
computed with Python integers (independent of the C restatement)."""
import numpy as np

P = 0xFFFFFFFF00000001


def test_field_ops_vs_python(oracle):
    rng = np.random.default_rng(1)
    a = oracle.rand_field(rng, 2000); b = oracle.rand_field(rng, 2000)
    edge = np.array([0, 1, 2, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000], np.uint64)
    a[:8] = edge; b[:8] = edge[::-1]
    m = oracle.vec_mul(a, b); s = oracle.vec_add(a, b); d = oracle.vec_sub(a, b)
    for i in range(2000):
        x, y = int(a[i]), int(b[i])
        assert int(m[i]) == x * y % P and int(s[i]) == (x + y) % P and int(d[i]) == (x - y) % P
    assert oracle.mul(oracle.inv(12345), 12345) == 1
    assert oracle.fpow(7, P - 1) == 1


def test_roots_of_unity(oracle):
    # POWER_OF_TWO_GENERATOR = 7^((p-1)/2^32); primitive_root_of_unity(k) has exact order 2^k
    assert oracle.root_of_unity(32) == pow(7, (P - 1) >> 32, P) == 1753635133440165772
    for k in (1, 3, 12, 20, 23):
        w = oracle.root_of_unity(k)
        assert pow(w, 1 << k, P) == 1 and pow(w, 1 << (k - 1), P) == P - 1
    # every 64th root of unity is +-2^j: basis of the shift-only butterflies in the HIP kernels
    w64 = oracle.root_of_unity(6)
    assert any(pow(2, 3 * t, P) == w64 for t in range(64))


def _horner(c, x):
    r = 0
    for v in reversed(c):
        r = (r * x + int(v)) % P
    return r


def test_fft_is_evaluation(oracle):
    rng = np.random.default_rng(2)
    for lg in (0, 1, 4, 7):
        n = 1 << lg
        c = oracle.rand_field(rng, n)
        v = oracle.fft(c)
        w = oracle.root_of_unity(lg)
        for i in range(n):
            assert int(v[i]) == _horner(c, pow(w, i, P))
        assert (oracle.ifft(v) == c).all()
        cv = oracle.coset_fft(c, 7)
        for i in range(0, n, max(1, n // 8)):
            assert int(cv[i]) == _horner(c, 7 * pow(w, i, P) % P)
        assert (oracle.coset_ifft(cv, 7) == c).all()


def test_lde_is_coset_evaluation(oracle):
    rng = np.random.default_rng(3)
    lg, rb = 6, 3
    c = oracle.rand_field(rng, 1 << lg)
    v = oracle.lde(c, rb)
    W = oracle.root_of_unity(lg + rb)
    for i in (0, 1, 5, 100, 511):
        assert int(v[i]) == _horner(c, 7 * pow(W, i, P) % P)


def test_merkle_paths_and_cap(oracle):
    rng = np.random.default_rng(4)
    for (nl, ll, ch) in ((64, 9, 2), (16, 3, 0), (8, 20, 3), (32, 135, 4)):
        leaves = oracle.rand_field(rng, (nl, ll))
        dig, cap = oracle.merkle_build(leaves, ch)
        assert cap.shape == (1 << ch, 4)
        for idx in (0, 1, nl // 2, nl - 1):
            sib = oracle.merkle_prove(dig, nl, ch, idx)
            assert sib.shape[0] == (nl.bit_length() - 1) - ch
            assert oracle.merkle_verify(leaves[idx], idx, cap, sib)
            bad = leaves[idx].copy(); bad[0] ^= np.uint64(1)
            assert not oracle.merkle_verify(bad, idx, cap, sib)
    # cap_height == log2(leaves): the cap is the leaf digests themselves
    leaves = oracle.rand_field(rng, (4, 2))
    dig, cap = oracle.merkle_build(leaves, 2)
    assert (cap[:, :2] == leaves).all() and (cap[:, 2:] == 0).all()


def test_batch_from_values_layout(oracle):
    rng = np.random.default_rng(5)
    ncols, lg, rb, ch = 5, 5, 3, 2
    vals = oracle.rand_field(rng, (ncols, 1 << lg))
    b = oracle.batch_from_values(vals, rb, ch)
    br = oracle.bitrev_perm(lg + rb)
    for c in range(ncols):
        assert (b.coeffs[c] == oracle.ifft(vals[c])).all()
        assert (b.leaves[:, c] == oracle.lde(b.coeffs[c], rb)[br]).all()
    # LDE restricted to every 8th point of the coset g*<W> is NOT the trace (coset), but the
    # polynomial is the same: fft(coeffs) must return the trace values
    assert (oracle.fft(b.coeffs[0]) == vals[0]).all()
    dig, cap = oracle.merkle_build(b.leaves, ch)
    assert (cap == b.cap).all() and (dig == b.digests).all()


def test_challenger_duplex(oracle):
    ch = oracle.Challenger()
    ch.observe([1, 2, 3])
    st = np.zeros(12, np.uint64); st[:3] = [1, 2, 3]
    st = oracle.poseidon_permute(st)
    # challenges pop from the end of the squeezed rate portion
    assert ch.get() == int(st[7]) and ch.get() == int(st[6])
    ch.observe([9])           # invalidates buffered outputs
    st[0] = 9
    st = oracle.poseidon_permute(st)
    assert ch.get() == int(st[7])
    # 8 observed elements trigger an immediate duplexing
    ch2 = oracle.Challenger(); ch2.observe(list(range(8)))
    st2 = oracle.poseidon_permute(np.array(list(range(8)) + [0] * 4, np.uint64))
    assert ch2.get() == int(st2[7])


def test_oracle_matches_committed_golden_fixture(oracle):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "batch_5x32.npz"))
    b = oracle.batch_from_values(g["values"], 3, 2)
    assert (b.coeffs == g["coeffs"]).all() and (b.leaves == g["leaves"]).all()
    assert (b.digests == g["digests"]).all() and (b.cap == g["cap"]).all()


def test_oracle_prover_matches_committed_golden_proof(oracle):
    """tests/golden/proof_zkdsa_2_3.npz (made by tests/golden/make_golden.py): every word of the zkdsa proof, the circuit
    digest and the constants/sigmas cap must be reproduced by today's oracle."""
    import os
    import plonky2_lib_amd.synth as synth
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "proof_zkdsa_2_3.npz"))
    desc = synth.zkdsa_circuit(3)
    oc = oracle.OracleCircuit(desc)
    rc, proof = oc.prove()
    assert rc == 0
    assert (np.asarray(desc.circuit_digest, np.uint64) == g["circuit_digest"]).all()
    assert (oc.cs_cap == g["constants_sigmas_cap"]).all()
    assert (np.asarray(desc.public_inputs, np.uint64) == g["public_inputs"]).all()
    assert (proof == g["proof"]).all()
    assert oc.verify(g["proof"]) == 0
