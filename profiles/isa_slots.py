#!/usr/bin/env python3
"""Issue-slot count of one Poseidon permutation, from the gfx950 ISA hipcc emits for csrc/merkle.hip.
A slot = one full-rate VALU wave-instruction (2 cycles on a SIMD-32); per-opcode weights are calibrated by
profiles/r01_ubench_opcode_rates.txt (full-rate set below counts 1, everything else 1.75).  The permutation is three loops: 4 full rounds, 22 partial rounds, 4 full rounds; per-loop
bodies are recognised by their multiply count.  Prints JSON."""
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
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-o", out, src], stderr=subprocess.DEVNULL)
        txt = open(out).read()
    body = txt[txt.index("_ZN3glp16k_permute_statesEPmm:"):]
    body = body[:body.index("s_endpgm")]
    blocks, cur = [], []
    for l in body.split("\n"):
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append(cur); cur = []
        else:
            m = re.match(r"^\s+([vs]_\w+|ds_\w+|global_\w+)", l)
            if m:
                cur.append(m.group(1))
    blocks.append(cur)
    res = []
    for b in blocks:
        c = collections.Counter(b)
        mads = c["v_mad_u64_u32"]
        if mads < 100:
            continue
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        slots = sum(v * (1.0 if k in FULL else HALF_WEIGHT) for k, v in c.items() if k.startswith("v_"))
        res.append({"mads": mads, "valu_instructions": valu, "issue_slots": slots, "s_nop": c["s_nop"]})
    # loop bodies: the partial round has the fewest multiplies; a full-round body has 11 more S-boxes
    part_mads = min(r["mads"] for r in res)
    part = [r for r in res if r["mads"] == part_mads]
    full = [r for r in res if 1.4 * part_mads < r["mads"] < 2.6 * part_mads]
    per_full = min(r["issue_slots"] for r in full) if full else None
    per_part = min(r["issue_slots"] for r in part) if part else None
    out = {"loop_bodies": res, "full_round_slots": per_full, "partial_round_slots": per_part,
           "slots_per_permutation": (8 * per_full + 22 * per_part) if per_full and per_part else None}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
