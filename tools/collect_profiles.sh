#!/bin/bash
# Collect the judged profile set of one build on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r04
# kernel-trace stats of the bench command, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc runs), SQ counters -- for the
# headline (bench.py) AND, since round 4, for BASELINE configs 2 (batch 1024 and 65536) and 4 (tools/config_profile.py), plus the
# inputs of the weighted VALU model (tools/valu_issue_cost, tools/valu_mix.py).
# Everything lands under gpurun_out/prof_<tag>/ and the reduced summaries under gpurun_out/profiles_<tag>/ (copy the
# latter into profiles/).  rocprofv3 gets `python3 bench.py ...` directly after `--` (no wrapper: the profiler's
# preloaded library initialises the GPU before the program starts).
set -eo pipefail
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-valu-floor --no-configs"
BENCH5="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-valu-floor --no-configs"
SQCNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
# the weighted VALU model's inputs first: measured issue cost per instruction class, exact stream mix (stamped with the source hash)
if [ -x "$ROOT/tools/valu_issue_cost" ]; then
  timeout -k 10 300 "$ROOT/tools/valu_issue_cost" > "$SUM/${TAG}_valu_issue_cost.json" 2> "$OUT/valu_issue_cost.log"
  cp "$SUM/${TAG}_valu_issue_cost.json" "$ROOT/profiles/"
fi
python3 $ROOT/tools/valu_mix.py > "$SUM/${TAG}_valu_mix.json" 2> /dev/null
cp "$SUM/${TAG}_valu_mix.json" "$ROOT/profiles/"
cd /tmp
python3 $ROOT/bench.py --explain > "$SUM/${TAG}_bench.json"   # the full line (explanatory keys kept: tools/design_table.py reads them); the driver runs the slim default
tail -c 600 "$SUM/${TAG}_bench.json"; echo
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $BENCH > "$OUT/stats.log" 2>&1
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$SUM/${TAG}_kernel_stats.csv"
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o run -- $BENCH5 > "$OUT/fetch.log" 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o run -- $BENCH5 > "$OUT/write.log" 2>&1
echo "write done"
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
python3 $ROOT/tools/pmc_summary.py "$F" "$W" > "$SUM/${TAG}_pmc_traffic.json"
rocprofv3 --pmc $SQCNT \
    --output-format csv -d "$OUT/sq" -o run -- $BENCH5 > "$OUT/sq.log" 2>&1
echo "sq done"
S=$(find "$OUT/sq" -name '*counter_collection.csv' | head -1)
python3 $ROOT/tools/sq_summary.py "$S" > "$SUM/${TAG}_sq_counters.json"
cp "$SUM/${TAG}_pmc_traffic.json" "$SUM/${TAG}_sq_counters.json" "$ROOT/profiles/"
head -c 1500 "$SUM/${TAG}_pmc_traffic.json"; echo
# ---- the other BASELINE configurations under the same standard: per-kernel durations, HBM bytes, SQ counters per OPERATION ----
collect_config() {  # $1 = config key of tools/configs.py, $2 = reps under --stats, $3 = reps under --pmc (kernels are serialised there)
  local CFG=$1 OPS=$(( $3 + 3 ))
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${CFG}_stats" -o run -- python3 $ROOT/tools/config_profile.py $CFG $2 > "$OUT/${CFG}_stats.log" 2>&1
  cp "$(find "$OUT/${CFG}_stats" -name '*kernel_stats.csv' | head -1)" "$SUM/${TAG}_${CFG}_kernel_stats.csv"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${CFG}_fetch" -o run -- python3 $ROOT/tools/config_profile.py $CFG $3 > "$OUT/${CFG}_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${CFG}_write" -o run -- python3 $ROOT/tools/config_profile.py $CFG $3 > "$OUT/${CFG}_write.log" 2>&1
  python3 $ROOT/tools/config_summary.py pmc $CFG $OPS "$(find "$OUT/${CFG}_fetch" -name '*counter_collection.csv' | head -1)" \
      "$(find "$OUT/${CFG}_write" -name '*counter_collection.csv' | head -1)" > "$SUM/${TAG}_${CFG}_pmc_traffic.json"
  rocprofv3 --pmc $SQCNT --output-format csv -d "$OUT/${CFG}_sq" -o run -- python3 $ROOT/tools/config_profile.py $CFG $3 > "$OUT/${CFG}_sq.log" 2>&1
  python3 $ROOT/tools/config_summary.py sq $CFG $OPS "$(find "$OUT/${CFG}_sq" -name '*counter_collection.csv' | head -1)" > "$SUM/${TAG}_${CFG}_sq_counters.json"
  cp "$SUM/${TAG}_${CFG}_pmc_traffic.json" "$SUM/${TAG}_${CFG}_sq_counters.json" "$ROOT/profiles/"
  tail -1 "$OUT/${CFG}_stats.log"; echo "$CFG done"
}
collect_config cfg2 200 20
collect_config cfg2_sat 20 5
collect_config cfg4 10 3
# the judged bench line once more, now that the counter summaries of THIS build exist (headline AND configs 2 / 4): bench.py quotes
# roofline.traffic and the VALU counters only from profiles/<tag>_*.json stamped with the tree's kernel-source hash
cd /tmp
python3 $ROOT/bench.py --explain > "$SUM/${TAG}_bench.json"   # the full line (explanatory keys kept: tools/design_table.py reads them); the driver runs the slim default
tail -c 300 "$SUM/${TAG}_bench.json"; echo
# ... and the line exactly as the driver gets it (no flags: numbers only, < 6 KB)
python3 $ROOT/bench.py > "$SUM/${TAG}_bench_driver_line.json"
wc -c "$SUM/${TAG}_bench_driver_line.json"
ls -la "$SUM"
