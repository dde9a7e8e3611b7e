#!/bin/bash
for m in 0 1 2 3; do for st in 0 2 4 6; do echo "== mode $m stagger $st"; OMR_FUSED_STAGGER_MODE=$m OMR_FUSED_STAGGER=$st python tools/bwd_fused_shapes.py 2>/dev/null | grep "cout 32 cin 32\|cout 16" | cut -c1-60; done; done
