"""Regenerates tests/golden/*.npz from the CPU oracle (seeded) and the sample circuit hand-off file zkdsa_2_3.glpc.  The oracle itself is pinned by the
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
    # one complete proof: the reference's simple-signature circuit [REF src/zkdsa/circuits/mod.rs:24-43] as rebuilt by
    # plonky2_lib_amd.synth.zkdsa_circuit (seeded), proved by the oracle prover
    import plonky2_lib_amd.synth as synth
    desc = synth.zkdsa_circuit(3)
    oc = o.OracleCircuit(desc)
    rc, proof = oc.prove()
    assert rc == 0 and oc.verify(proof) == 0
    np.savez_compressed(os.path.join(HERE, "proof_zkdsa_2_3.npz"), proof=proof, circuit_digest=np.asarray(desc.circuit_digest, np.uint64),
                        constants_sigmas_cap=oc.cs_cap, public_inputs=np.asarray(desc.public_inputs, np.uint64))
    # the same circuit + witness as a circuit hand-off file (include/glp.h, glp_circuit_file_*): what a machine with the Rust
    # builder would ship to the GPU box.  Written by the PRODUCT library's writer (host code, no GPU needed).
    # zkdsa_2_3.glpc is a VERSION 1 file (checksum over the sections only), written by the round-2 library and kept as the
    # fixture of the reader's compatibility path: it is not regenerated.  zkdsa_2_3_v2.glpc is the same content in the current
    # version (checksum over header and sections).
    import plonky2_lib_amd as glp
    glp.write_circuit_file(os.path.join(HERE, "zkdsa_2_3_v2.glpc"), desc, with_witness=True)


if __name__ == "__main__":
    main()
