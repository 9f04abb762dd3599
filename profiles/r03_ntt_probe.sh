#!/bin/bash
# usage (GPU box, repo root): bash profiles/r03_ntt_probe.sh > gpurun_out/r03_ntt_two_pass_probe.txt
# Stage times (min of 3) of a 136-column PolynomialBatch commitment at 2^20 / 2^21 / 2^22 rows: two passes with the
# k_strided32 LDS tiles (tiles per block = software-pipelining depth, 16- and 8-column forms) against the three-pass path of
# round 2 (GLP_NTT_2PASS_LG=20).
set -e
echo "== 2^20 (unchanged path)";                 python3 profiles/lde_probe.py 136 20
for LG in 21 22; do
  echo "== 2^$LG two-pass, default (16-col tiles, 8 tiles per block)";  python3 profiles/lde_probe.py 136 $LG
  echo "== 2^$LG two-pass, 1 tile per block (no pipelining)";   GLP_NTT_STRIDED32_TL=1 python3 profiles/lde_probe.py 136 $LG
  echo "== 2^$LG two-pass, 4 tiles per block";                  GLP_NTT_STRIDED32_TL=4 python3 profiles/lde_probe.py 136 $LG
  echo "== 2^$LG two-pass, 17 tiles per block";                 GLP_NTT_STRIDED32_TL=17 python3 profiles/lde_probe.py 136 $LG
  echo "== 2^$LG two-pass, 8-col tiles";                        GLP_NTT_STRIDED32_LW=3 python3 profiles/lde_probe.py 136 $LG
  echo "== 2^$LG three-pass (r02)";                             GLP_NTT_2PASS_LG=20 python3 profiles/lde_probe.py 136 $LG
done
