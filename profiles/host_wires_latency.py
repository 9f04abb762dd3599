#!/usr/bin/env python3
"""glp_prove from HOST wires (the PCIe-inclusive path a Rust caller takes) against glp_prove_device (witness resident in
HBM, what bench.py times), headline shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonky2_lib_amd as glp, plonky2_lib_amd.synth as synth

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
desc = synth.ecdsa_shape_circuit(lg)
ctx = glp.Context(0)
c = glp.Circuit(ctx, desc)
w = np.ascontiguousarray(desc.wires)
dw = torch.from_numpy(w.view(np.int64)).cuda()
for name, fn in (("device wires", lambda: c.prove_device(dw.data_ptr())), ("host wires (pageable)", lambda: c.prove(w))):
    fn(); fn()
    t = time.time()
    for _ in range(4):
        p = fn()
    print("%-24s %.1f ms/proof" % (name, (time.time() - t) / 4 * 1e3), flush=True)
