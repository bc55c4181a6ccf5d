#!/bin/bash
# Build libntt_hip.so as of git revision REV into ab/libntt_NAME.so (for same-process A/B against the working tree):
#   tools/ab_build_rev.sh NAME REV ["-DEXTRA ..."]
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p ab
W=$(mktemp -d /tmp/abrev.XXXXXX)
git worktree add -f --detach "$W" "$2" > /dev/null 2>&1
make -C "$W/ntt_aie_amd/csrc" -j8 OBJDIR="$ROOT/ab/build_$1" OUT="$ROOT/ab/libntt_$1.so" EXTRA="$3" 2>&1 | grep -E "error|Error" || true
git worktree remove --force "$W"
ls -la "ab/libntt_$1.so"
