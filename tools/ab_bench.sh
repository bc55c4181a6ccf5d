#!/bin/bash
# Same-box A/B of bench.py: tools/ab_bench.sh ROUNDS NAME [NAME ...]   (NAME = default | a library built by ab_build.sh)
# prints ms_per_step and the per-pass kernel times of every run, alternating the variants ROUNDS times
cd "$(dirname "$0")/.."
rounds=$1; shift
for i in $(seq 1 "$rounds"); do
  for name in "$@"; do
    if [ "$name" = default ]; then unset NTT_HIP_LIB; else export NTT_HIP_LIB=$PWD/ab/libntt_$name.so; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline ${AB_BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s' % '$name', round(d['ms_per_step'],4), [round(x,4) for x in d['roofline']['pass_ms']])"
  done
done
