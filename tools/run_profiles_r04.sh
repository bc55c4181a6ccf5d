#!/bin/bash
# One GPU-box call that produces the judged r04 evidence under gpurun_out/profiles_r04/ (copy into profiles/):
#   tools/collect_profiles.sh r04 (headline bench line, rocprofv3 kernel stats, HBM traffic + SQ counters keyed by the full kernel
#   identity; the same three files for configs 2 / 2-saturating / 4; the weighted VALU model's inputs), forward kernels real vs
#   VALU floor with the held clock, power probe, every BASELINE config with its roofline object, the general 64-bit modulus
#   beside Goldilocks, and the one-process / N-device line in rehearsal mode.
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
S=gpurun_out/profiles_r04
mkdir -p $S
bash tools/collect_profiles.sh r04 > gpurun_out/collect.log 2>&1 || { tail -40 gpurun_out/collect.log; exit 1; }
tail -5 gpurun_out/collect.log
python3 tools/power_probe.py > $S/r04_power_probe.txt 2>&1 || true
tail -6 $S/r04_power_probe.txt
CNT="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
cd /tmp
rocprofv3 --pmc $CNT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq_real -o run -- python3 $GRAFT_REPO_ROOT/tools/ab_pass.py --rounds 2 --reps 3 real=ntt_aie_amd/libntt_hip_exp.so > $GRAFT_REPO_ROOT/gpurun_out/sq_real.log 2>&1
rocprofv3 --pmc $CNT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq_floor -o run -- python3 $GRAFT_REPO_ROOT/tools/ab_pass.py --rounds 2 --reps 3 --dbg 3 floor=ntt_aie_amd/libntt_hip_exp.so > $GRAFT_REPO_ROOT/gpurun_out/sq_floor.log 2>&1
cd $GRAFT_REPO_ROOT
{ echo "# real forward kernels (experiment build, no debug flags), rocprofv3 --pmc $CNT"; python3 tools/sq_table.py $(find gpurun_out/sq_real -name '*counter_collection.csv' | head -1);
  echo; echo "# the same kernels with L2-resident loads and no stores (ntt_plan_set_debug(3)): the VALU floor"; python3 tools/sq_table.py $(find gpurun_out/sq_floor -name '*counter_collection.csv' | head -1); } > $S/r04_sq_real_vs_floor.txt
cat $S/r04_sq_real_vs_floor.txt
rm -rf gpurun_out/sq_real gpurun_out/sq_floor
python3 tools/bench_configs.py > $S/r04_bench_all_configs.jsonl 2> gpurun_out/cfg.err || true
python3 tools/bench_m64.py > $S/r04_bench_m64.jsonl 2>> gpurun_out/cfg.err || true
NTT_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 2 --single-process --no-cpu-baseline --no-valu-floor > $S/r04_bench_single_process_rehearsal.json 2>> gpurun_out/cfg.err || true
ls -la $S
