import time, numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import plonky2_lib_amd as glp, plonky2_lib_amd.synth as synth
ctx = glp.Context(0)
for name, desc in (("zkdsa 2^3", synth.zkdsa_circuit(3)), ("poseidon chain 2^12", synth.poseidon_chain_circuit(12)), ("keccak-shape 2^15", synth.keccak_shape_circuit(15))):
    gc = glp.Circuit(ctx, desc)
    gc.prove()
    t=time.perf_counter(); n=20
    for _ in range(n): gc.prove()
    dt=(time.perf_counter()-t)/n
    print(name, "%.2f ms/proof"%(dt*1e3))
    ctx.set_profiling(True); ctx.stage_reset(); gc.prove(); ctx.synchronize()
    print("  ", {k: round(ms,3) for k,ms,_ in ctx.stages() if ms>0.2})
    ctx.set_profiling(False)
    gc.free()
