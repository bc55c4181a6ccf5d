set -eo pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 tools/ab_pass.py --rounds 7 base=ab/libntt_base.so new=ntt_aie_amd/libntt_hip.so > gpurun_out/ab1.log 2>&1
cat gpurun_out/ab1.log
python3 tools/ab_pass.py --rounds 3 --dbg 3 floor=ntt_aie_amd/libntt_hip_exp.so >> gpurun_out/ab1.log 2>&1
tail -1 gpurun_out/ab1.log
CNT="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
cd /tmp
rocprofv3 --pmc $CNT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq_real -o run -- python3 $GRAFT_REPO_ROOT/tools/ab_pass.py --rounds 2 --reps 3 real=ntt_aie_amd/libntt_hip_exp.so > $GRAFT_REPO_ROOT/gpurun_out/sq_real.log 2>&1
rocprofv3 --pmc $CNT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq_floor -o run -- python3 $GRAFT_REPO_ROOT/tools/ab_pass.py --rounds 2 --reps 3 --dbg 3 floor=ntt_aie_amd/libntt_hip_exp.so > $GRAFT_REPO_ROOT/gpurun_out/sq_floor.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/sq_table.py $(find gpurun_out/sq_real -name '*counter_collection.csv' | head -1) > gpurun_out/sq_real.txt
python3 tools/sq_table.py $(find gpurun_out/sq_floor -name '*counter_collection.csv' | head -1) > gpurun_out/sq_floor.txt
cat gpurun_out/sq_real.txt gpurun_out/sq_floor.txt
rm -rf gpurun_out/sq_real gpurun_out/sq_floor
