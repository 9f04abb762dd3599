#!/usr/bin/env python3
"""Issue-slot count of one Poseidon permutation, from the gfx950 ISA hipcc emits for csrc/merkle.hip.
A slot = one full-rate VALU wave-instruction (2 cycles on a SIMD-32); per-opcode weights are calibrated by
profiles/r01_ubench_opcode_rates.txt (full-rate set below counts 1, everything else 1.75).  Every basic block of k_permute_states is
counted and weighted by its trip count (loops: 4 full rounds, 5 four-round partial blocks, 3 full rounds).  Prints JSON."""
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
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DGLP_LDE_NO_HOIST", "-S", "--cuda-device-only",
                               "-o", out, src], stderr=subprocess.DEVNULL)
        txt = open(out).read()
    body = txt[txt.index("_ZN3glp16k_permute_statesEPmm:"):]
    body = body[:body.index("s_endpgm")]
    # basic blocks in program order; a block that branches back to its own label is a loop body
    blocks, cur, label = [], [], None
    for l in body.split("\n"):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((label, cur)); cur, label = [], m.group(1)
            continue
        m = re.match(r"^\s+([vs]_\w+|ds_\w+|global_\w+)(.*)", l)
        if m:
            cur.append((m.group(1), m.group(2)))
            if m.group(1).startswith("s_cbranch") and label is not None and label in m.group(2):
                blocks.append((label, cur)); cur, label = [], None      # code after a back-branch is straight-line again
    blocks.append((label, cur))
    # permute(): 4 full rounds (loop x4), partial rounds 4..23 as 5 blocks of 4 (loop x5), rounds 24..25 as one
    # block of 2 (straight line), 3 full rounds (loop x3), last full round (straight line)
    trips = [4, 5, 3]
    res, total, loops = [], 0.0, 0
    for lab, b in blocks:
        ops = [o for o, _ in b]
        c = collections.Counter(ops)
        is_loop = lab is not None and any(o.startswith("s_cbranch") and lab in rest for o, rest in b)
        mads = c["v_mad_u64_u32"]
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        slots = sum(v * (1.0 if k in FULL else HALF_WEIGHT) for k, v in c.items() if k.startswith("v_"))
        if valu == 0:
            continue
        trip = 1
        if is_loop:
            if loops >= len(trips):
                raise SystemExit("unexpected loop structure in k_permute_states")
            trip = trips[loops]; loops += 1
        total += trip * slots
        res.append({"label": lab, "loop": is_loop, "trip_count": trip, "mads": mads, "valu_instructions": valu,
                    "issue_slots": slots, "s_nop": c["s_nop"]})
    if loops != len(trips):
        raise SystemExit("expected %d loops in k_permute_states, found %d" % (len(trips), loops))
    out = {"blocks": res, "mads_per_permutation": sum(r["mads"] * r["trip_count"] for r in res),
           "slots_per_permutation": total}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
