"""`bench.perf_rs_variant` on its own for any number of signatures: the real secp256k1 circuit built on the host, proved a few times from an HBM-resident
witness with the stage timers on, the last proof verified.  20 signatures = the `perf.rs` batch (2^21 rows); 42 = a 2^22-row trace (1024-row LDS tiles).
usage: python profiles/perf_rs_probe.py [nsig=20] [steps=3]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import plonky2_lib_amd as glp
import bench
nsig = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_device(0)
ctx = glp.Context(0)
out = bench.perf_rs_variant(glp, ctx, torch, torch.device("cuda", 0), np, nsig=nsig, steps=steps)
print(json.dumps(out, indent=1))
