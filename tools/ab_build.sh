#!/bin/bash
# A/B builds of libntt_hip.so with extra -D defines:  tools/ab_build.sh NAME "-DNTT_X=1 ..."  -> ab/libntt_NAME.so
# (run an experiment with NTT_HIP_LIB=$PWD/ab/libntt_NAME.so; ab/ is git-ignored but travels with gpurun)
set -e
cd "$(dirname "$0")/.."
mkdir -p ab
make -C ntt_aie_amd/csrc -j8 OBJDIR="$PWD/ab/build_$1" OUT="$PWD/ab/libntt_$1.so" EXTRA="$2" 2>&1 | grep -E "error|Error" || true
ls -la "ab/libntt_$1.so"
