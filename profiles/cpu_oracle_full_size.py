"""The CPU oracle's prove() of the REAL headline circuit at full size (10 secp256k1 signatures, 2^20 rows x 136 wires), run to completion:
the measurement bench.py's `cpu_baseline` only estimates (commitments at full size + the other stages scaled from 2^17 rows).
usage: python profiles/cpu_oracle_full_size.py [log_n=20] > profiles/r03_cpu_oracle_full_size.txt   (offline: minutes of CPU, ~40 GB of RAM)"""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from plonky2_lib_amd import gadgets_ecdsa
import bench

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
oracle.build()
cores = max(1, min(oracle.max_threads(), bench.usable_cores()))
oracle.set_threads(cores)
nsig = ((1 << lg) - 7714 - 2) // 98687
t0 = time.perf_counter()
desc = gadgets_ecdsa.ecdsa_circuit(gadgets_ecdsa.random_signatures(nsig, seed=bench.SEED), min_log_n=lg)
t_build = time.perf_counter() - t0
print("circuit: %d signatures, 2^%d rows x %d wires, %d gates, built in %.1f s" % (nsig, desc.degree_bits, desc.num_wires, len(desc.gates), t_build), flush=True)
t0 = time.perf_counter()
oc = oracle.OracleCircuit(desc)                 # commits constants ++ sigmas (84 columns): the build()-time half, not part of prove()
t_cs = time.perf_counter() - t0
print("constants/sigmas commitment (build time, not prove): %.1f s" % t_cs, flush=True)
t0 = time.perf_counter()
rc, proof = oc.prove()
t_prove = time.perf_counter() - t0
print("oracle prove(): rc=%d, %.1f s wall on %d threads (%s)" % (rc, t_prove, cores, os.uname().nodename), flush=True)
t0 = time.perf_counter()
ok = oc.verify(proof) == 0
print("oracle verify(): %s, %.2f s" % ("accepted" if ok else "REJECTED", time.perf_counter() - t0))
print("peak RSS %.1f GB" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6))
print("proofs/sec = %.5f" % (1.0 / t_prove))
