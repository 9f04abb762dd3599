"""merkle_leaves stage time of a PolynomialBatch commitment with the leaf hash forced to each of its three forms (one state per lane, per quad of
lanes, per 12 of 16 lanes): where the crossovers MERKLE_COOP_MAX_LEAVES / MERKLE_QUAD_MAX_LEAVES sit.  Needs a build that reads GLP_COOP_MAX / GLP_QUAD_MAX.
usage: python profiles/leaf_form_probe.py [ncols]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky2_lib_amd as glp
ncols = int(sys.argv[1]) if len(sys.argv) > 1 else 135
forms = {"lane": ("0", "0"), "quad": ("0", str(1 << 40)), "coop": (str(1 << 40), str(1 << 40))}
ctxs = {}
for f, (cm, qm) in forms.items():
    os.environ["GLP_MERKLE_COOP_MAX"], os.environ["GLP_MERKLE_QUAD_MAX"] = cm, qm
    ctxs[f] = glp.Context(0)
print("leaves   " + "  ".join("%9s" % f for f in forms) + "   (merkle_leaves stage, ms, %d columns, best of 5)" % ncols)
for lg in range(5, 17):
    caps, row = [], []
    for f, ctx in ctxs.items():
        d = ctx.dev_alloc(8 * ncols << lg)
        ctx.fill_random_device(d, ncols << lg, 12345)
        b = ctx.batch_from_values_device(d, ncols, lg); caps.append(b.cap().copy()); b.free()
        ctx.set_profiling(True); ctx.stage_reset()
        for _ in range(5):
            b = ctx.batch_from_values_device(d, ncols, lg); b.free()
        ctx.synchronize()
        row.append(min(ms for name, ms, by in ctx.stages() if name == "merkle_leaves"))
        ctx.set_profiling(False)
        ctx.dev_free(d)
    assert all((c == caps[0]).all() for c in caps), "the three forms disagree"
    print("2^%-2d  " % (lg + 3) + "  ".join("%9.3f" % v for v in row))
