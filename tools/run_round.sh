#!/bin/bash
# The GPU-box calls a round's judged evidence comes from, as ONE documented script (rounds 3-4 kept a wrapper per call; their
# one-off A/B wrappers are in the history: `git show a073ee5:tools/run_ab_col8_r04.sh` etc., results under profiles/r0N_ab_*.txt).
#   gpurun --timeout 1200 -- 'bash tools/run_round.sh profiles r05'   -> gpurun_out/profiles_r05/  (copy into profiles/)
#       tools/collect_profiles.sh (headline bench line, rocprofv3 kernel stats, HBM traffic + SQ counters keyed by the full kernel
#       identity; the same three for configs 2 / 2-saturating / 4; the issue-cost probe), forward kernels real vs VALU floor with the
#       held clock, power probe, every BASELINE config with its roofline object, the general 64-bit modulus beside Goldilocks, the
#       one-process / N-device line in rehearsal mode
#   gpurun --timeout 1200 -- 'bash tools/run_round.sh final r05'      -> gpurun_out/final_r05/
#       the reference-format series (profile/kerneltime, profile/exectime) re-made with this round's kernels, the determinism soak,
#       the PCIe-inclusive rate
set -eo pipefail
WHAT=${1:?profiles|final}
TAG=${2:?round tag, e.g. r05}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
case "$WHAT" in
profiles)
  S=gpurun_out/profiles_$TAG
  mkdir -p $S
  # the vector ALU's throughput per instruction form: the price list of tools/hw.py (tools/valu_peak.hip, built HERE)
  if [ -x tools/valu_peak ]; then timeout -k 10 200 tools/valu_peak > $S/${TAG}_valu_peak.txt 2>&1 || true; fi
  # the butterfly statement against occupancy, 1 .. 8 waves per SIMD (tools/stream_occupancy.hip; built right here, stamped with
  # the hash of the generator + probe it was made from: bench.statement_steady_state quotes it only when that hash is the tree's)
  if [ ! -x tools/stream_occupancy ]; then
    { mkdir -p ab && NTT_GEN_W=1 python3 tools/gen_gl_asm.py ab/gl_asm_w.h && hipcc -O3 --offload-arch=gfx950 -I ab tools/stream_occupancy.hip -o tools/stream_occupancy; } > gpurun_out/build_stream_occupancy.log 2>&1 || true
  fi
  if [ -x tools/stream_occupancy ]; then
    { echo "# stream_src_hash $(python3 -c 'import bench; print(bench.stream_source_hash())')"; timeout -k 10 200 tools/stream_occupancy; } > $S/${TAG}_stream_occupancy.txt 2>&1 || true
  else echo "SKIPPED: stream occupancy table (tools/stream_occupancy failed to build: gpurun_out/build_stream_occupancy.log)"; fi
  cp $S/${TAG}_valu_peak.txt $S/${TAG}_stream_occupancy.txt profiles/ 2>/dev/null || true  # (on the box: so that the bench line below quotes THIS round's tables)
  if [ ! -x tools/valu_peak ]; then echo "SKIPPED: valu_peak (tools/valu_peak absent: __graft_entry__.build() makes it)"; fi
  bash tools/collect_profiles.sh $TAG > gpurun_out/collect.log 2>&1 || { tail -40 gpurun_out/collect.log; exit 1; }
  tail -5 gpurun_out/collect.log
  python3 tools/power_probe.py > $S/${TAG}_power_probe.txt 2>&1 || true
  tail -6 $S/${TAG}_power_probe.txt
  CNT="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
  cd /tmp
  rocprofv3 --pmc $CNT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq_real -o run -- python3 $GRAFT_REPO_ROOT/tools/ab_pass.py --rounds 2 --reps 3 real=ntt_aie_amd/libntt_hip_exp.so > $GRAFT_REPO_ROOT/gpurun_out/sq_real.log 2>&1
  rocprofv3 --pmc $CNT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq_floor -o run -- python3 $GRAFT_REPO_ROOT/tools/ab_pass.py --rounds 2 --reps 3 --dbg 3 floor=ntt_aie_amd/libntt_hip_exp.so > $GRAFT_REPO_ROOT/gpurun_out/sq_floor.log 2>&1
  cd $GRAFT_REPO_ROOT
  { echo "# real forward kernels (experiment build, no debug flags), rocprofv3 --pmc $CNT"; python3 tools/sq_table.py $(find gpurun_out/sq_real -name '*counter_collection.csv' | head -1);
    echo; echo "# the same kernels with L2-resident loads and no stores (ntt_plan_set_debug(3)): the VALU floor"; python3 tools/sq_table.py $(find gpurun_out/sq_floor -name '*counter_collection.csv' | head -1); } > $S/${TAG}_sq_real_vs_floor.txt
  cat $S/${TAG}_sq_real_vs_floor.txt
  rm -rf gpurun_out/sq_real gpurun_out/sq_floor
  # where the cycles of the two headline kernels go (diagnostic side build with s_memtime stamps; git-ignored, so built right here
  # unless it travelled with the snapshot; a failed build is SAID, never silently skipped)
  [ -f ab/libntt_stamps.so ] || bash tools/ab_build.sh stamps -DNTT_PHASE_STAMPS > gpurun_out/build_stamps.log 2>&1 || true
  if [ ! -f ab/libntt_stamps.so ]; then echo "SKIPPED: phase stamps (ab/libntt_stamps.so failed to build: gpurun_out/build_stamps.log)"; fi
  if [ -f ab/libntt_stamps.so ]; then
    mkdir -p $S/trace
    timeout -k 10 300 python3 tools/phase_stamps.py --trace-prefix $S/trace/trace_mi355x_n16 --trace-cycles 150000 > $S/${TAG}_phase_stamps.json 2> gpurun_out/stamps.err || tail -3 gpurun_out/stamps.err
    timeout -k 10 300 python3 tools/phase_stamps.py --logn 12 --p 3221225473 --g 5 --word-bytes 4 --batch 1024 --shape 4,0,8,512 > $S/${TAG}_phase_stamps_cfg2.json 2>> gpurun_out/stamps.err || tail -3 gpurun_out/stamps.err
  fi
  python3 tools/bench_configs.py > $S/${TAG}_bench_all_configs.jsonl 2> gpurun_out/cfg.err || true
  python3 tools/bench_m64.py > $S/${TAG}_bench_m64.jsonl 2>> gpurun_out/cfg.err || true
  NTT_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 2 --single-process --no-cpu-baseline --no-valu-floor > $S/${TAG}_bench_single_process_rehearsal.json 2>> gpurun_out/cfg.err || true
  ls -la $S
  ;;
final)
  S=gpurun_out/final_$TAG
  mkdir -p $S
  python3 tests/profile_series.py gpurun_out/profiles_series > $S/profile_series.log 2>&1 || { tail -5 $S/profile_series.log; exit 1; }
  tail -3 $S/profile_series.log
  bash tools/kerneltime_rocprof.sh > $S/kerneltime_rocprof.log 2>&1 || true
  python3 tools/race_soak.py 100 > $S/${TAG}_race_soak.txt 2>&1 || { tail -5 $S/${TAG}_race_soak.txt; exit 1; }
  tail -8 $S/${TAG}_race_soak.txt
  python3 tools/pcie_inclusive.py > $S/${TAG}_pcie_inclusive.txt 2>&1 || true
  tail -3 $S/${TAG}_pcie_inclusive.txt
  ;;
*) echo "usage: run_round.sh profiles|final TAG"; exit 2 ;;
esac
