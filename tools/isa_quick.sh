#!/bin/bash
# Developer aid: registers / spills / scratch of the fused chains in a built .so
#   tools/isa_quick.sh genie2_amd/lib/abl/libgenie_fz_x.so
T=$(mktemp -d); cd $T
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input="$OLDPWD/$1" >/dev/null 2>&1
/opt/rocm/bin/roc-obj-ls "$OLDPWD/$1" 2>/dev/null | head -3
