"""GPU parity tests for the commitment half of prove(): Poseidon, NTT/iNTT/LDE and
PolynomialBatch (from_values / from_coeffs -> coefficients, leaves, every Merkle digest, cap, paths),
HIP library (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import plonky2_lib_amd as glp

pytestmark = pytest.mark.gpu
P = glp.P
KAT = [4330397376401421145, 14124799381142128323, 8742572140681234676, 14345658006221440202]


@pytest.fixture(scope="module")
def ctx():
    c = glp.Context(0)
    yield c
    c.close()


def test_poseidon_kat_on_gpu(ctx):
    # [REF src/zkdsa/circuits/mod.rs:85-101]
    out = ctx.poseidon_permute(np.zeros((3, 12), np.uint64))
    for row in out:
        assert [int(x) for x in row[:4]] == KAT


def test_poseidon_parity(ctx, oracle):
    rng = np.random.default_rng(11)
    st = oracle.rand_field(rng, (1000, 12))
    st[0] = P - 1
    st[1] = 0xFFFFFFFF
    st[2] = 0xFFFFFFFF00000000
    st[3, :] = [0, 1, 2, P - 1, P - 2, 1 << 32, (1 << 32) - 1, (1 << 63), 3, 5, 7, 11]
    got = ctx.poseidon_permute(st)
    for i in range(st.shape[0]):
        assert (got[i] == oracle.poseidon_permute(st[i])).all(), i


def test_poseidon_parity_edge_value_soak(ctx, oracle):
    """The device permutation keeps values non-canonical between layers and folds with hand-placed carry handling
    (inline asm): soak it with states built from boundary values of every limb, plus a large random batch."""
    special = np.array([0, 1, 2, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, 0xFFFFFFFF00000000, 0xFFFFFFFE00000001,
                        0xFFFFFFFEFFFFFFFF, 1 << 63, (1 << 63) - 1, 0x8000000080000000 % P, 0x00000000FFFFFFFE, 0xFFFFFFFF], dtype=np.uint64)
    rng = np.random.default_rng(2024)
    edge = special[rng.integers(0, len(special), size=(20000, 12))]
    mixed = oracle.rand_field(rng, (20000, 12))
    mask = rng.random((20000, 12)) < 0.5
    mixed[mask] = special[rng.integers(0, len(special), size=int(mask.sum()))]
    rnd = oracle.rand_field(rng, (100000, 12))
    st = np.concatenate([edge, mixed, rnd])
    got = ctx.poseidon_permute(st)
    ref = np.stack([oracle.poseidon_permute(r) for r in st])
    bad = np.nonzero((got != ref).any(axis=1))[0]
    assert bad.size == 0, "first mismatching state %d: %s" % (bad[0], [hex(int(x)) for x in st[bad[0]]])


@pytest.mark.parametrize("lg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 15, 16, 17, 18, 19, 20])
def test_fft_ifft_parity(ctx, oracle, lg):
    rng = np.random.default_rng(100 + lg)
    ncols = 3
    a = oracle.rand_field(rng, (ncols, 1 << lg))
    f = ctx.fft(a)
    i = ctx.ifft(a)
    for c in range(ncols):
        assert (f[c] == oracle.fft(a[c])).all()
        assert (i[c] == oracle.ifft(a[c])).all()
    assert (ctx.ifft(f) == a).all()


@pytest.mark.parametrize("lg,rb,shift", [(0, 3, 7), (1, 2, 7), (2, 4, 49), (3, 3, 7), (4, 1, 7 ** 4), (4, 3, 7), (5, 4, 7), (6, 1, 7), (7, 3, 7), (8, 4, 49), (8, 0, 7), (9, 3, 7 ** 16 % P), (12, 3, 7), (13, 3, 7), (14, 2, 49), (16, 3, 7),
                                         (17, 3, 7), (18, 1, 7), (19, 2, 49), (20, 3, 7)])
def test_lde_parity(ctx, oracle, lg, rb, shift):
    rng = np.random.default_rng(200 + lg)
    c = oracle.rand_field(rng, (2, 1 << lg))
    got = ctx.lde(c, rb, shift)
    for k in range(2):
        assert (got[k] == oracle.lde(c[k], rb, shift)).all()


def _check_batch(oracle, b, ref, probe):
    assert (b.cap() == ref.cap).all()
    assert (b.coeffs() == ref.coeffs).all()
    assert (b.digests() == ref.digests).all()
    nl = ref.leaves.shape[0]
    for j in probe:
        j %= nl
        assert (b.leaf(j) == ref.leaves[j]).all()
        sib = b.prove(j)
        assert (sib == ref.prove(j)).all()
        assert oracle.merkle_verify(ref.leaves[j], j, ref.cap, sib)


@pytest.mark.parametrize("ncols,lg,rb,ch", [
    (3, 4, 3, 4),      # <= 4 columns: hash_or_noop copies the leaf
    (4, 2, 3, 5),      # cap_height == log2(leaves): cap = leaf digests
    (5, 5, 3, 2),      # one partial sponge chunk
    (8, 6, 3, 4),      # exactly one full chunk
    (9, 3, 3, 0),      # cap_height 0 -> single root
    (135, 10, 3, 4),   # standard_recursion_config wire count
    (136, 13, 3, 4),   # standard_ecc_config wire count, two-pass NTT
    (20, 14, 3, 4),    # zs + partial products batch
    (16, 12, 3, 4),    # quotient chunks batch
    (2, 15, 1, 3),
])
def test_batch_from_values_parity(ctx, oracle, ncols, lg, rb, ch):
    rng = np.random.default_rng(1000 * ncols + lg)
    vals = oracle.rand_field(rng, (ncols, 1 << lg))
    ref = oracle.batch_from_values(vals, rb, ch)
    b = ctx.batch_from_values(vals, rb, ch)
    _check_batch(oracle, b, ref, [0, 1, 7, 12345, (1 << (lg + rb)) - 1])
    b.free()


@pytest.mark.parametrize("ncols,lg,rb,ch", [
    (1, 0, 3, 0),      # a single constant polynomial: n = 1
    (5, 0, 3, 3),      # n = 1, cap = all 8 leaves
    (2, 1, 1, 2),      # n = 2, LDE x2, cap height = log2(leaves)
    (300, 4, 3, 4),    # many columns: 38 sponge absorptions per leaf, 2400 planes
    (8, 7, 0, 3),      # rate_bits = 0: the "LDE" is the coset evaluation itself
])
def test_batch_edge_shapes(ctx, oracle, ncols, lg, rb, ch):
    rng = np.random.default_rng(ncols * 31 + lg)
    vals = oracle.rand_field(rng, (ncols, 1 << lg))
    ref = oracle.batch_from_values(vals, rb, ch)
    b = ctx.batch_from_values(vals, rb, ch)
    _check_batch(oracle, b, ref, [0, 1, 3, (1 << (lg + rb)) - 1])
    b.free()


@pytest.mark.parametrize("form", ["lane", "quad", "coop"])
def test_leaf_hash_forms(oracle, form):
    """The three forms of the Poseidon leaf hash (one sponge per lane / per quad of lanes / per 12 of 16 lanes; merkle.hip), each forced
    over the edge shapes by the thresholds a context reads at creation: identical trees."""
    import os
    env = {"lane": ("0", "0"), "quad": ("0", str(1 << 40)), "coop": (str(1 << 40), str(1 << 40))}[form]
    keys = ("GLP_MERKLE_COOP_MAX", "GLP_MERKLE_QUAD_MAX")
    old = {k: os.environ.get(k) for k in keys}
    os.environ.update(dict(zip(keys, env)))
    try:
        c2 = glp.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    try:
        for ncols, lg, rb, ch in [(3, 4, 3, 4), (5, 0, 3, 3), (9, 3, 3, 0), (8, 6, 3, 4), (17, 2, 1, 1), (300, 4, 3, 4), (135, 7, 3, 4)]:
            rng = np.random.default_rng(ncols * 131 + lg)
            vals = oracle.rand_field(rng, (ncols, 1 << lg))
            ref = oracle.batch_from_values(vals, rb, ch)
            b = c2.batch_from_values(vals, rb, ch)
            _check_batch(oracle, b, ref, [0, 1, 3, 77, (1 << (lg + rb)) - 1])
            b.free()
    finally:
        c2.close()


@pytest.mark.parametrize("ncols,lg,rb", [(1, 5, 0), (33, 5, 0), (7, 6, 1), (5, 8, 0), (9, 7, 2)])
def test_mid_size_transforms_ragged_column_counts(ctx, oracle, ncols, lg, rb):
    """k_lde_mid packs several 32..256-point columns into one workgroup when a column's cosets leave room: column counts that do not fill the last one"""
    rng = np.random.default_rng(ncols * 17 + lg)
    co = oracle.rand_field(rng, (ncols, 1 << lg))
    got = ctx.lde(co, rb, 7)
    for k in range(ncols):
        assert (got[k] == oracle.lde(co[k], rb, 7)).all()


def test_empty_batch_is_an_error(ctx):
    with pytest.raises(glp.GlpError):
        ctx.batch_from_values(np.zeros((0, 8), np.uint64), 3, 2)


@pytest.mark.parametrize("ncols,lg", [(16, 9), (7, 13)])
def test_batch_from_coeffs_parity(ctx, oracle, ncols, lg):
    rng = np.random.default_rng(77 + lg)
    co = oracle.rand_field(rng, (ncols, 1 << lg))
    ref = oracle.batch_from_coeffs(co, 3, 4)
    b = ctx.batch_from_coeffs(co, 3, 4)
    _check_batch(oracle, b, ref, [0, 3, 999, (1 << (lg + 3)) - 1])
    b.free()


def _check_transforms(ctx, oracle, lg, lde_rb):
    rng = np.random.default_rng(lg)
    a = oracle.rand_field(rng, (2, 1 << lg))
    f = ctx.fft(a)
    assert (f[0] == oracle.fft(a[0])).all() and (f[1] == oracle.fft(a[1])).all()
    i = ctx.ifft(a)
    assert (i[1] == oracle.ifft(a[1])).all()
    assert (ctx.ifft(f) == a).all()
    if lde_rb:
        got = ctx.lde(a[:1], lde_rb, 7)
        assert (got[0] == oracle.lde(a[0], lde_rb, 7)).all()


@pytest.mark.parametrize("lg,lde_rb", [(21, 3), (22, 1), (23, 0)])
def test_large_transforms(ctx, oracle, lg, lde_rb):
    """2^21 and 2^22 points: still TWO passes, the strided one on 512- / 1024-row tiles in the 160 KB LDS (k_strided32);
    2^23: three passes (k_outer over 2^20-point blocks)."""
    _check_transforms(ctx, oracle, lg, lde_rb)


@pytest.mark.parametrize("env", [{"GLP_NTT_2PASS_LG": "20"}, {"GLP_NTT_STRIDED32_LW": "3"}])
def test_large_transform_switches(oracle, env):
    """The tuning switches a context reads at creation (include/glp.h): the three-pass path at 2^21 (what round 2 shipped, kept
    for the A/B rows in profiles/) and the 8-column (64-byte row segment) form of the k_strided32 tile."""
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        c2 = glp.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    try:
        _check_transforms(c2, oracle, 21, 1)
    finally:
        c2.close()


def test_large_batch_2_21(ctx, oracle):
    rng = np.random.default_rng(77)
    vals = oracle.rand_field(rng, (5, 1 << 21))
    ref = oracle.batch_from_values(vals, 3, 4)
    b = ctx.batch_from_values(vals, 3, 4)
    assert (b.cap() == ref.cap).all()
    assert (b.coeffs(3, 1)[0] == ref.coeffs[3]).all()
    for j in (0, 12345678, (1 << 24) - 1):
        assert (b.leaf(j) == ref.leaves[j]).all() and (b.prove(j) == ref.prove(j)).all()
    b.free()


def test_golden_fixture(ctx):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "batch_5x32.npz"))
    b = ctx.batch_from_values(g["values"], 3, 2)
    assert (b.cap() == g["cap"]).all() and (b.coeffs() == g["coeffs"]).all()
    assert (b.digests() == g["digests"]).all()
    b.free()


def test_headline_shape_properties(ctx, oracle):
    """136 columns x 2^16 rows is past what the oracle checks in seconds per digest, so use
    size-independent properties: iNTT then NTT is the identity, sampled LDE points equal a Horner
    evaluation, sampled Merkle paths recompute to the cap."""
    rng = np.random.default_rng(5)
    ncols, lg = 136, 16
    vals = oracle.rand_field(rng, (ncols, 1 << lg))
    b = ctx.batch_from_values(vals, 3, 4)
    co = b.coeffs(0, 2)
    assert (ctx.fft(co) == vals[:2]).all()
    cap = b.cap()
    br = oracle.bitrev_perm(lg + 3)
    W = oracle.root_of_unity(lg + 3)
    for j in (0, 5, 77777, (1 << (lg + 3)) - 1):
        leaf = b.leaf(j)
        assert oracle.merkle_verify(leaf, j, cap, b.prove(j))
        x = 7 * pow(W, int(br[j]), P) % P
        for c in (0, 1):
            acc = 0
            for v in co[c][::-1]:
                acc = (acc * x + int(v)) % P
            assert acc == int(leaf[c])
    b.free()


def test_errors(ctx):
    with pytest.raises(glp.GlpError):
        ctx.batch_from_values(np.zeros((2, 8), np.uint64), 3, 7)      # cap_height > log2(leaves)
    with pytest.raises(glp.GlpError):
        ctx.batch_from_values(np.zeros((2, 12), np.uint64), 3, 1)     # not a power of two
    with pytest.raises(glp.GlpError) as e:
        glp.Batch._make_dev(ctx, "glp_batch_from_values_device", 8, 1, 25, 3, 4)      # > 2^24 rows
    assert e.value.code == -3


def test_lde_plan_cache_is_bounded_and_stays_correct(ctx, oracle):
    """glp_lde keys its coset tables by the caller's shift (unbounded growth).  More distinct shifts than the cache
    holds: results stay right before, across and after evictions, including a shift whose plan was evicted and rebuilt."""
    rng = np.random.default_rng(60)
    c = oracle.rand_field(rng, (1, 1 << 13))
    shifts = [int(x) for x in oracle.rand_field(rng, (70,)) if int(x) != 0]
    first = ctx.lde(c, 2, shifts[0])
    assert (first[0] == oracle.lde(c[0], 2, shifts[0])).all()
    for s in shifts[1:]:
        got = ctx.lde(c, 2, s)
        assert got[0][123] == oracle.lde(c[0], 2, s)[123]
    again = ctx.lde(c, 2, shifts[0])                 # evicted by now (70 > LDE_PLAN_CACHE_MAX = 48): rebuilt
    assert (again == first).all()
    big = oracle.rand_field(rng, (1, 1 << 21))        # a three-pass plan pins its inner plan while others are evicted around it
    assert (ctx.lde(big, 1, 7)[0][::4097] == oracle.lde(big[0], 1, 7)[::4097]).all()
    for s in shifts[:50]:
        ctx.lde(c, 1, s)
    assert (ctx.lde(big, 1, 7)[0][::4097] == oracle.lde(big[0], 1, 7)[::4097]).all()
