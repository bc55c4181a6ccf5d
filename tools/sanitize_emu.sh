#!/bin/bash
# The host index model (tests/emu: the kernels' own pass.h / plan.h / field.h compiled with g++) under UBSan and ASan, on the CPU
# (GPU sanitizers are not available on the pool).  Runs tests/test_emu.py + test_emu_property.py against each instrumented build and
# restores the ordinary one.  usage: tools/sanitize_emu.sh
set -e
cd "$(dirname "$0")/.."
SO=tests/emu/libntt_emu.so
python3 -c "import sys; sys.path.insert(0, 'tests'); import emu_lib; emu_lib.lib()"   # make sure the ordinary build exists
cp $SO /tmp/libntt_emu_plain.so
trap 'cp /tmp/libntt_emu_plain.so '"$SO"'; touch '"$SO" EXIT
echo "== UBSan"
g++ -O1 -g -std=c++17 -shared -fPIC -fsanitize=undefined -fno-sanitize-recover=undefined tests/emu/emu.cpp -o $SO; touch $SO
python3 -m pytest tests/test_emu.py tests/test_emu_property.py -x -q -k "not hazard_tracker"
echo "== ASan"
g++ -O1 -g -std=c++17 -shared -fPIC -fsanitize=address tests/emu/emu.cpp -o $SO; touch $SO
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python3 -m pytest tests/test_emu.py tests/test_emu_property.py -x -q -k "not hazard_tracker"
