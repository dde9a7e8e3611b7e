#!/bin/bash
# Build libomr_hip.so in-tree (hipcc --offload-arch=gfx950); prints only errors.
make -C "$(dirname "$0")/omr_a2s_multimodal_transformer_amd/csrc" -j8 2>&1 | grep -E "error|Error|warning: unused" -A3 | head -30
ls -la --time-style=+%T "$(dirname "$0")/omr_a2s_multimodal_transformer_amd/libomr_hip.so" | awk '{print $6, $7}'
