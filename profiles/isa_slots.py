#!/usr/bin/env python3
"""Issue-slot count of one Poseidon permutation, from the gfx950 ISA hipcc emits for csrc/merkle.hip.
A slot = one full-rate VALU wave-instruction (2 cycles on a SIMD-32); per-opcode weights are calibrated by
profiles/r01_ubench_opcode_rates.txt (full-rate set below counts 1, everything else 1.75).  The kernel is compiled with every round
loop unrolled (-DGLP_POSEIDON_FLAT), so the static count of k_permute_states is the dynamic count.  Prints JSON."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Measured on gfx950 (profiles/r01_ubench_opcode_rates.txt): only plain VOP2 add/sub/logic/mov/right-shift issue at
# full rate; multiplies, multiply-adds, every carry-in/carry-out add, compares, selects with an SGPR mask, 64-bit
# and three-operand forms issue at about half of it.
HALF_WEIGHT = 1.75     # measured: ~57 vs ~100 lane-ops/clk/CU (profiles/r01_ubench_opcode_rates.txt)
FULL = {"v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_xor_b32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_mov_b32_e32",
        "v_lshrrev_b32_e32", "v_not_b32_e32", "v_cndmask_b32_e32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}


def main():
    src = os.path.join(ROOT, "plonky2-lib_amd", "csrc", "merkle.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "merkle.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DGLP_LDE_NO_HOIST", "-DGLP_POSEIDON_FLAT", "-S", "--cuda-device-only",
                               "-o", out, src], stderr=subprocess.DEVNULL)
        txt = open(out).read()
    body = txt[txt.index("_ZN3glp16k_permute_statesEPmm:"):]
    body = body[:body.index("s_endpgm")]
    # compiled with -DGLP_POSEIDON_FLAT every round loop is unrolled: the kernel is straight-line code (the bounds
    # check branches forward only), so the static instruction count IS the dynamic count of one permutation
    ops = []
    labels = set()
    for l in body.split("\n"):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels.add(m.group(1))
            continue
        m = re.match(r"^\s+([vs]_\w+|ds_\w+|global_\w+)(.*)", l)
        if m:
            if m.group(1).startswith("s_cbranch") and any(lab in m.group(2) for lab in labels):
                raise SystemExit("backward branch in the flat build: loops were not fully unrolled")
            ops.append(m.group(1))
    c = collections.Counter(ops)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    slots = sum(v * (1.0 if k in FULL else HALF_WEIGHT) for k, v in c.items() if k.startswith("v_"))
    out = {"build": "-DGLP_POSEIDON_FLAT (all round loops unrolled; the shipped build keeps them rolled: same instructions per round)",
           "mads_per_permutation": c["v_mad_u64_u32"], "valu_instructions": valu, "s_nop": c["s_nop"],
           "full_rate_instructions": sum(v for k, v in c.items() if k in FULL),
           "slots_per_permutation": slots, "top_opcodes": dict(c.most_common(14))}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
