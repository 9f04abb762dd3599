"""KeccakGoldilocksConfig on the GPU, commitment half: the Keccak-256 primitive against the reference's known-answer pairs
[REF src/hash/keccak256.rs:196-212,256-277] and the `KeccakHash<25>` Merkle commitment (leaves, every digest, cap, paths) against the
oracle, through the C ABI."""
import numpy as np
import pytest

import plonky2_lib_amd as glp
from test_oracle_keccak import LONG_IN, LONG_OUT, SHORT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = glp.Context(0)
    yield c
    c.close()


def test_keccak256_reference_vectors_on_gpu(ctx, oracle):
    for msg, dig in SHORT + [(LONG_IN, LONG_OUT)]:
        assert ctx.keccak256([bytes.fromhex(msg)])[0].hex() == dig
    rng = np.random.default_rng(2)
    for n in (0, 1, 7, 8, 9, 135, 136, 137, 271, 272, 273, 1000):
        msgs = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for _ in range(70)]
        got = ctx.keccak256(msgs)
        assert got == [oracle.keccak256(m) for m in msgs], n


@pytest.mark.parametrize("ncols,lg,rb,ch", [
    (1, 3, 3, 2), (3, 4, 3, 4),            # hash_or_noop copies up to 3 elements (24 bytes fit a 25-byte digest) ...
    (4, 4, 3, 3),                          # ... and hashes 4
    (16, 5, 3, 2), (17, 5, 3, 2), (18, 5, 2, 0),       # around one 136-byte rate block (17 lanes): padding block of its own at 17
    (34, 6, 3, 4), (135, 8, 3, 4), (136, 10, 3, 4), (20, 12, 3, 4), (5, 14, 1, 3),
])
def test_keccak_batch_parity(ctx, oracle, ncols, lg, rb, ch):
    rng = np.random.default_rng(100 * ncols + lg)
    vals = oracle.rand_field(rng, (ncols, 1 << lg))
    ref = oracle.batch_from_values(vals, rb, ch, hasher=1)
    b = ctx.batch_from_values(vals, rb, ch, hasher=1)
    assert (b.coeffs() == ref.coeffs).all()
    assert (b.digests() == ref.digests).all()
    assert (b.cap() == ref.cap).all()
    nl = ref.leaves.shape[0]
    for j in (0, 1, nl // 3, nl - 1):
        assert (b.leaf(j) == ref.leaves[j]).all()
        sib = b.prove(j)
        assert (sib == ref.prove(j)).all()
        assert oracle.merkle_verify(ref.leaves[j], j, ref.cap, sib, hasher=1)
    # every digest is 25 bytes: the top 7 bytes of its last word are zero
    assert (b.digests()[:, 3] <= 0xFF).all()
    # and the Poseidon commitment of the same values is a different tree
    p = ctx.batch_from_values(vals, rb, ch)
    assert not (p.cap() == b.cap()).all()
    b.free(); p.free()


def test_keccak_batch_from_coeffs(ctx, oracle):
    rng = np.random.default_rng(9)
    co = oracle.rand_field(rng, (7, 1 << 9))
    ref = oracle.batch_from_coeffs(co, 3, 4, hasher=1)
    b = ctx.batch_from_coeffs(co, 3, 4, hasher=1)
    assert (b.cap() == ref.cap).all() and (b.digests() == ref.digests).all()
    b.free()
    with pytest.raises(glp.GlpError):
        ctx.batch_from_values(co, 3, 4, hasher=7)
