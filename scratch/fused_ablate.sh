#!/bin/bash
for d in 0 1 2 3 4 8 12 7 11 15; do echo "== OMR_FUSED_DBG=$d"; OMR_FUSED_DBG=$d python tools/bwd_fused_shapes.py $1 2>/dev/null | grep "cout 32 cin 32\|apply-on" | head -2 | cut -c1-60; done
