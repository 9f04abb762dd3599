import numpy as np, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky2_lib_amd as glp, plonky2_lib_amd.synth as synth
K = 256
rng = np.random.default_rng(1)
descs = [synth.zkdsa_circuit(3, seed=5, private_key=synth.gl.rand(rng, 4), message=synth.gl.rand(rng, 4)) for _ in range(K)]
ctx = glp.Context(0); gc = glp.Circuit(ctx, descs[0])
w = np.stack([d.wires for d in descs]); pis = np.stack([d.public_inputs for d in descs])
gc.prove_batch(w, pis)
os.environ["GLP_BATCH_TRACE"] = "1"
t=time.perf_counter(); gc.prove_batch(w, pis); print("total ms", (time.perf_counter()-t)*1e3)
p = gc.prove_batch(w, pis)
pw = p[:, gc.proof_words - len(descs[0].public_inputs) - 1].astype(np.float64)
print("proof-of-work candidates below the witnesses: sum(w + 1) = %.0f (expected K 2^16 = %.0f), max %.0f" % ((pw + 1).sum(), K * 65536.0, pw.max()))
for _ in range(3):
    t = time.perf_counter(); ok = gc.verify_batch(p); print("verify_batch total ms %.3f  all accepted %s" % ((time.perf_counter() - t) * 1e3, bool(ok.all())))
