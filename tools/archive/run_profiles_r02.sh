#!/bin/bash
# One GPU-box call that produces every r02 artefact under gpurun_out/profiles_r02/ (copy into profiles/).
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
S=gpurun_out/profiles_r02
mkdir -p $S
bash tools/collect_profiles.sh r02 > gpurun_out/collect.log 2>&1 || { tail -30 gpurun_out/collect.log; exit 1; }
tail -5 gpurun_out/collect.log
python3 tools/power_probe.py > $S/r02_power_probe.txt 2>&1
tail -6 $S/r02_power_probe.txt
python3 tools/ab_pass.py --rounds 7 r01_arithmetic=ab/libntt_base.so r02=ntt_aie_amd/libntt_hip.so > $S/r02_ab_same_process.txt 2>&1
python3 tools/ab_pass.py --rounds 3 --dbg 3 r02_valu_floor=ntt_aie_amd/libntt_hip_exp.so >> $S/r02_ab_same_process.txt 2>&1
cat $S/r02_ab_same_process.txt
CNT="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
cd /tmp
rocprofv3 --pmc $CNT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq_real -o run -- python3 $GRAFT_REPO_ROOT/tools/ab_pass.py --rounds 2 --reps 3 real=ntt_aie_amd/libntt_hip_exp.so > $GRAFT_REPO_ROOT/gpurun_out/sq_real.log 2>&1
rocprofv3 --pmc $CNT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq_floor -o run -- python3 $GRAFT_REPO_ROOT/tools/ab_pass.py --rounds 2 --reps 3 --dbg 3 floor=ntt_aie_amd/libntt_hip_exp.so > $GRAFT_REPO_ROOT/gpurun_out/sq_floor.log 2>&1
cd $GRAFT_REPO_ROOT
{ echo "# real kernels (experiment build, no debug flags), rocprofv3 --pmc $CNT"; python3 tools/sq_table.py $(find gpurun_out/sq_real -name '*counter_collection.csv' | head -1);
  echo; echo "# the same kernels with L2-resident loads and no stores (NTT_DEBUG_FLAGS=3): the VALU floor"; python3 tools/sq_table.py $(find gpurun_out/sq_floor -name '*counter_collection.csv' | head -1); } > $S/r02_sq_real_vs_floor.txt
cat $S/r02_sq_real_vs_floor.txt
rm -rf gpurun_out/sq_real gpurun_out/sq_floor
python3 tools/bench_configs.py > $S/r02_bench_all_configs.jsonl 2> gpurun_out/cfg.err
# second half of round 2 (linear tile copies, per-exchange syncs, tapered launch tail) against the library as of 59eb281, same process:
# ab/libntt_r02a.so = tools/ab_build_rev.sh r02a 59eb281
{ V="r02a_59eb281=ab/libntt_r02a.so final=ntt_aie_amd/libntt_hip.so"
  python3 tools/ab_latency.py $V
  python3 tools/ab_latency.py --inverse $V
  python3 tools/ab_latency.py --p 12289 --g 11 $V
  python3 tools/ab_latency.py --batch 65536 --k 10 $V
  python3 tools/ab_latency.py --p 998244353 --g 3 --batch 65536 --k 10 $V
  python3 tools/ab_latency.py --logn 8 --p 3329 --g 3 --batch 1048576 --k 10 $V
  python3 tools/ab_latency.py --logn 12 --p 18446744069414584321 --g 7 --word-bytes 8 --batch 16384 --k 10 $V
  python3 tools/ab_latency.py --logn 16 --p 18446744069414584321 --g 7 --word-bytes 8 --batch 4096 --k 10 $V
  python3 tools/ab_latency.py --logn 20 --p 18446744069414584321 --g 7 --word-bytes 8 --batch 512 --k 5 $V
} 2>&1 | grep -v amdgpu.ids > $S/r02_ab_second_half.txt
cat $S/r02_ab_second_half.txt
python3 tools/chunk_streams.py 128 512 > $S/r02_chunk_streams.txt 2>&1
cp gpurun_out/profiles_r02/../profiles_r02/* $S/ 2>/dev/null || true
ls -la $S
