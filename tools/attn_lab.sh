#!/bin/bash
# DiT attention loop with one ingredient removed at a time (lab library: python -m flowmirror_hydravox_amd.build --lab attn -- -DHVX_LAB, then HVX_LIB_PATH=<printed path>; results of LAB != 0 are garbage, timing only)
# 0 product | 1 no global/stash | 2 no barrier | 3 = 1+2 | 7 = 3 + fragments from registers | 11 = 3 + no exp/cvt | 15 = 7 + no exp/cvt (MFMA only)
# 19 = 3 + no MFMA | 23 = 7 + no MFMA (VALU only)
for lab in 0 1 2 3 7 11 15 19 23; do
  echo "== attn_lab=$lab"
  ATTN_LAB=$lab ITERS=10 python tools/attn_probe.py 8
done
