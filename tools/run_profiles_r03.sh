#!/bin/bash
# One GPU-box call that produces the judged r03 evidence under gpurun_out/profiles_r03/ (copy into profiles/):
#   bench line, rocprofv3 kernel stats, HBM traffic + SQ counters keyed by the full PassCfg<...> argument list (forward and
#   inverse kernels apart), forward kernels real vs VALU floor with the held clock, power probe, every BASELINE config,
#   the general 64-bit modulus beside Goldilocks, the regression sweep against round 2's library (ab/libntt_r02.so:
#   tools/ab_build_rev.sh r02 35635cf).
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
S=gpurun_out/profiles_r03
mkdir -p $S
bash tools/collect_profiles.sh r03 > gpurun_out/collect.log 2>&1 || { tail -30 gpurun_out/collect.log; exit 1; }
tail -3 gpurun_out/collect.log
python3 tools/power_probe.py > $S/r03_power_probe.txt 2>&1 || true
tail -6 $S/r03_power_probe.txt
CNT="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
cd /tmp
rocprofv3 --pmc $CNT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq_real -o run -- python3 $GRAFT_REPO_ROOT/tools/ab_pass.py --rounds 2 --reps 3 real=ntt_aie_amd/libntt_hip_exp.so > $GRAFT_REPO_ROOT/gpurun_out/sq_real.log 2>&1
rocprofv3 --pmc $CNT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq_floor -o run -- python3 $GRAFT_REPO_ROOT/tools/ab_pass.py --rounds 2 --reps 3 --dbg 3 floor=ntt_aie_amd/libntt_hip_exp.so > $GRAFT_REPO_ROOT/gpurun_out/sq_floor.log 2>&1
cd $GRAFT_REPO_ROOT
{ echo "# real forward kernels (experiment build, no debug flags), rocprofv3 --pmc $CNT"; python3 tools/sq_table.py $(find gpurun_out/sq_real -name '*counter_collection.csv' | head -1);
  echo; echo "# the same kernels with L2-resident loads and no stores (ntt_plan_set_debug(3)): the VALU floor"; python3 tools/sq_table.py $(find gpurun_out/sq_floor -name '*counter_collection.csv' | head -1); } > $S/r03_sq_real_vs_floor.txt
cat $S/r03_sq_real_vs_floor.txt
rm -rf gpurun_out/sq_real gpurun_out/sq_floor
python3 tools/bench_configs.py > $S/r03_bench_all_configs.jsonl 2> gpurun_out/cfg.err || true
python3 tools/bench_m64.py > $S/r03_bench_m64.jsonl 2>> gpurun_out/cfg.err || true
if [ -f ab/libntt_r02.so ]; then
  python3 tools/regress_sweep.py ab/libntt_r02.so ntt_aie_amd/libntt_hip.so > $S/r03_regress_gl.txt 2>&1 || true
  python3 tools/regress_sweep.py --word-bytes 4 --p 998244353 --g 3 ab/libntt_r02.so ntt_aie_amd/libntt_hip.so > $S/r03_regress_m32_lazy.txt 2>&1 || true
  python3 tools/regress_sweep.py --word-bytes 4 --p 3221225473 --g 5 ab/libntt_r02.so ntt_aie_amd/libntt_hip.so > $S/r03_regress_m32_any.txt 2>&1 || true
  grep -h "^#" $S/r03_regress_*.txt
fi
ls -la $S
