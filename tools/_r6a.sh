set -eo pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6a
python -m pytest tests -m gpu -x -q > gpurun_out/r6a/gpu_tests.log 2>&1 || { tail -30 gpurun_out/r6a/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r6a/gpu_tests.log
for n in 10 11; do for b in 1 255 256; do
python3 tools/ab_latency.py --logn $n --p 18446744069414584321 --g 7 --word-bytes 8 --batch $b --rounds 7 --k 30 r05=ab/libntt_base.so fixed=ntt_aie_amd/libntt_hip.so
done; done > gpurun_out/r6a/r06_ab_dma_clamp.txt 2>&1
cat gpurun_out/r6a/r06_ab_dma_clamp.txt
python3 bench.py > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err; python3 -c "
import json; d=json.load(open('gpurun_out/r6a/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], [c.get('ms') for c in d.get('configs',[])])"
