#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for r in 1 2; do for lib in "$@"; do ARCLE_HIP_LIB=$R/$lib timeout 300 python tools/researchbench.py 2>&1 | grep -E "full|Error|error" | tail -2; done; done
