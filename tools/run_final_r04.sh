#!/bin/bash
# Last validation call of round 4 on the final tree: the reference-format series (profile/kerneltime, profile/exectime) re-made with
# this round's kernels, the determinism soak incl. the new kernel variants, the PCIe-inclusive rate.  -> gpurun_out/final_r04/
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
S=gpurun_out/final_r04
mkdir -p $S
python3 tests/profile_series.py gpurun_out/profiles_series > $S/profile_series.log 2>&1 || { tail -5 $S/profile_series.log; exit 1; }
tail -3 $S/profile_series.log
bash tools/kerneltime_rocprof.sh > $S/kerneltime_rocprof.log 2>&1 || true
python3 tools/race_soak.py 100 > $S/r04_race_soak.txt 2>&1 || { tail -5 $S/r04_race_soak.txt; exit 1; }
tail -8 $S/r04_race_soak.txt
python3 tools/pcie_inclusive.py > $S/r04_pcie_inclusive.txt 2>&1 || true
tail -3 $S/r04_pcie_inclusive.txt
