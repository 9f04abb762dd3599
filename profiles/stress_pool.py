#!/usr/bin/env python3
"""Stress check (not a benchmark): many proofs of alternating sizes through one context, then a second context; free HBM
must return to its starting level apart from the pool the live contexts keep, and proofs must stay identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonky2_lib_amd as glp, plonky2_lib_amd.synth as synth

free0 = torch.cuda.mem_get_info()[0]
ctx = glp.Context(0)
descs = [synth.arith_circuit(lg, synth.Config.standard_ecc_config(), seed=lg) for lg in (10, 16, 12, 17)]
circs = [glp.Circuit(ctx, d) for d in descs]
first = [c.prove() for c in circs]
marks = []
for it in range(12):
    for c, f in zip(circs, first):
        assert (c.prove() == f).all()
    marks.append(torch.cuda.mem_get_info()[0])
print("free HBM after each sweep (GiB):", [round(m / 2**30, 2) for m in marks])
assert max(marks[2:]) - min(marks[2:]) < 64 << 20, "pool keeps growing"
for c in circs:
    c.free()
ctx.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("free before %.2f GiB, after closing the context %.2f GiB" % (free0 / 2**30, free1 / 2**30))
assert free0 - free1 < 256 << 20, "device memory not returned"
print("ok")
