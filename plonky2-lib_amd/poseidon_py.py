"""Poseidon over Goldilocks on Python integers, recording the S-box inputs of every round: host-side
witness generation for PoseidonGate rows of synthetic circuits (what plonky2's `PoseidonGenerator`
does inside `generate_partial_witness`).  Not on the prove() path."""
from .tools.gen_poseidon_constants import all_round_constants

P = 0xFFFFFFFF00000001
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
_RC = None


def _rc():
    global _RC
    if _RC is None:
        _RC = all_round_constants()
    return _RC


def _mds(s):
    return [(sum(s[(i + r) % 12] * CIRC[i] for i in range(12)) + (8 * s[0] if r == 0 else 0)) % P for r in range(12)]


def permute_trace(state):
    """Returns (output state, full_sbox_0 [3][12], partial_sbox [22], full_sbox_1 [4][12])."""
    rc, s, k = _rc(), [int(x) % P for x in state], 0
    f0, part, f1 = [], [], []
    for r in range(4):
        s = [(s[i] + rc[k + i]) % P for i in range(12)]; k += 12
        if r:
            f0.append(list(s))
        s = _mds([pow(x, 7, P) for x in s])
    for r in range(22):
        s = [(s[i] + rc[k + i]) % P for i in range(12)]; k += 12
        part.append(s[0])
        s[0] = pow(s[0], 7, P)
        s = _mds(s)
    for r in range(4):
        s = [(s[i] + rc[k + i]) % P for i in range(12)]; k += 12
        f1.append(list(s))
        s = _mds([pow(x, 7, P) for x in s])
    return s, f0, part, f1


def permute(state):
    return permute_trace(state)[0]


def hash_no_pad(xs):
    st = [0] * 12
    xs = [int(x) for x in xs]
    for off in range(0, len(xs), 8):
        chunk = xs[off:off + 8]
        st[:len(chunk)] = chunk
        st = permute(st)
    return st[:4]
