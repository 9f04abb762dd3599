"""Regenerates tests/golden/*.npz from the CPU oracle (seeded).  The oracle itself is pinned by the
reference's Poseidon known-answer vector; see tests/test_oracle_poseidon.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as o  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    o.build()
    rng = np.random.default_rng(0x5EED0001)
    vals = o.rand_field(rng, (5, 32))
    b = o.batch_from_values(vals, 3, 2)
    np.savez_compressed(os.path.join(HERE, "batch_5x32.npz"), values=vals, coeffs=b.coeffs, leaves=b.leaves,
                        digests=b.digests, cap=b.cap)


if __name__ == "__main__":
    main()
