"""Kernel timeline of ONE glp_prove_batch call (no GLP_BATCH_TRACE, so no stream synchronisations between the stages):
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 profiles/batch_timeline.py [zkdsa | smt | keccak]      (256 / 256 / 16 proofs)
    python profiles/batch_timeline.py summarize <kernel_trace.csv>
prints every kernel of the last batch with its start offset, the idle gap in front of it and its duration."""
import csv, os, sys, time
if len(sys.argv) > 2 and sys.argv[1] == "summarize":
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[2]))]
    rows.sort()
    # the last batch: around the last k_pow_batch2, bounded by idle gaps of more than 0.1 ms on either side
    last = max(i for i, r in enumerate(rows) if "k_pow_batch2" in r[2])
    lo = hi = last
    while lo > 0 and rows[lo][0] - rows[lo - 1][1] < 100000: lo -= 1
    while hi + 1 < len(rows) and rows[hi + 1][0] - rows[hi][1] < 100000: hi += 1
    rows = rows[lo:hi + 1]
    t0, busy, prev = rows[0][0], 0, rows[0][0]
    for s, e, n in rows:
        print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n[:70]))
        busy += e - s
        prev = max(prev, e)
    print("span %.3f ms, kernels %d, sum of durations %.3f ms" % ((prev - t0) / 1e6, len(rows), busy / 1e6))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plonky2_lib_amd as glp, plonky2_lib_amd.synth as synth
what = sys.argv[1] if len(sys.argv) > 1 else "zkdsa"
rng = np.random.default_rng(1)
if what == "smt":
    from plonky2_lib_amd import gadgets
    tree = gadgets.SparseMerkleTree()
    keys = [tuple(int(x) for x in rng.integers(0, 1 << 32, 4)) for _ in range(128)]
    for k in keys:
        tree.insert(k, tuple(int(x) for x in rng.integers(1, 1 << 32, 4)))
    pick = [k for k in (keys[int(rng.integers(0, 128))] for _ in range(2000)) if len(tree.find(k)["siblings"]) < 16][:256]
    descs = [gadgets.smt_inclusion_circuit(tree, k, public=True) for k in pick]
elif what == "keccak":
    from plonky2_lib_amd import gadgets
    descs = [gadgets.keccak256_circuit(bytes(rng.integers(0, 256, int(rng.integers(0, 136)), dtype=np.uint8))) for _ in range(16)]
else:
    descs = [synth.zkdsa_circuit(3, seed=5, private_key=synth.gl.rand(rng, 4), message=synth.gl.rand(rng, 4)) for _ in range(256)]
ctx = glp.Context(0); gc = glp.Circuit(ctx, descs[0])
w = np.stack([d.wires for d in descs]); pis = np.stack([d.public_inputs for d in descs])
for _ in range(3):
    gc.prove_batch(w, pis)
    time.sleep(0.002)
t = time.perf_counter(); gc.prove_batch(w, pis); print("total ms", (time.perf_counter() - t) * 1e3)
