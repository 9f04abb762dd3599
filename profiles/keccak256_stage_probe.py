import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import plonky2_lib_amd as glp
from plonky2_lib_amd import gadgets
ctx = glp.Context(0)
c = gadgets.keccak256_circuit(b"x" * 500, blocks_num=4)
gc = glp.Circuit(ctx, c)
w = np.ascontiguousarray(c.wires); d = ctx.dev_alloc(w.nbytes); ctx.dev_upload(d, w)
gc.prove_device(d, c.public_inputs)
t = time.perf_counter()
for _ in range(10): gc.prove_device(d, c.public_inputs)
print("keccak 4-block 2^15: %.2f ms" % ((time.perf_counter() - t) * 100))
ctx.set_profiling(True); ctx.stage_reset(); gc.prove_device(d, c.public_inputs); ctx.synchronize()
acc = {}
for k, ms, _ in ctx.stages(): acc[k] = acc.get(k, 0) + ms
print({k: round(v, 3) for k, v in acc.items()}, "sum %.2f" % sum(acc.values()))
