"""Stage times of one 136-column x 2^20 PolynomialBatch commitment (iNTT, LDE, leaf hash, levels): the A/B harness for NTT kernel variants.
usage: python profiles/lde_probe.py [ncols] [log_n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky2_lib_amd as glp
ncols = int(sys.argv[1]) if len(sys.argv) > 1 else 136
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = glp.Context(0)
d = ctx.dev_alloc(8 * ncols << lg)
ctx.fill_random_device(d, ncols << lg, 12345)
b = ctx.batch_from_values_device(d, ncols, lg); b.free()
ctx.set_profiling(True); ctx.stage_reset()
for _ in range(3):
    b = ctx.batch_from_values_device(d, ncols, lg); b.free()
ctx.synchronize()
acc = {}
for name, ms, by in ctx.stages():
    acc.setdefault(name, []).append(ms)
print({k: round(min(v), 3) for k, v in acc.items()})
