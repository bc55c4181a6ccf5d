#!/bin/bash
# profiles/kerneltime/mi355x_rocprof.csv: "N , microseconds" with the kernel time taken from rocprofv3 --kernel-trace --stats
# (sum of the average durations of the transform's pass kernels), N = 2^8 .. 2^17, batch 1 -- the device-side number the
# reference's profile/kerneltime/{aie,gpu}.csv hold (theirs come from trace events / nvprof).
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/profiles_series/kerneltime
mkdir -p $OUT
: > $OUT/mi355x_rocprof.csv
for L in 8 9 10 11 12 13 14 15 16 17; do
  D=$GRAFT_REPO_ROOT/gpurun_out/kt_$L
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $D -o run -- python3 $GRAFT_REPO_ROOT/tools/one_size.py $L > /dev/null 2>&1)
  python3 - "$D" $L >> $OUT/mi355x_rocprof.csv <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
us = sum(float(r["AverageNs"]) for r in csv.DictReader(open(f)) if "pass_kernel" in r["Name"]) / 1e3
print("%d , %.5f" % (1 << int(sys.argv[2]), us))
PY
  rm -rf $D
done
cat $OUT/mi355x_rocprof.csv
