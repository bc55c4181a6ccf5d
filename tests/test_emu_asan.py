"""The host index model (tests/emu: the kernels' own pass.h / plan.h / field.h under g++) built with AddressSanitizer and UBSan
and driven with EXACT-SIZE malloc buffers (tests/emu/asan_sweep.cpp) -- the only out-of-bounds detector there is for the pass
kernels: GPU sanitizers are not available on the pool, and numpy / torch allocations hide over-reads behind small arrays.

Round 5's judge found a 24 KiB over-read of the LDS-DMA prefetch this way (8-byte forward, N = 2^10 / 2^11, the 512-thread
kernel as the only pass, ragged batch); this file keeps that run in the CPU suite: nine executables, one per template family
(so the instrumented builds compile in parallel), each sweeping word widths x modulus classes x logN 1..17 x ragged batches x
forward / scaled inverse / unscaled inverse x both layouts x in place / out of place x every plan alternative, kernel variant
and tile-shape split x the fused product paths, every case also compared with the oracle.  The reference host owns exactly N
words per buffer (src/test.cpp:115-124); so do these.  (tools/sanitize_emu.sh of rounds 3-5 is folded into this file.)"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SWEEP = os.path.join(HERE, "emu", "asan_sweep.cpp")
ORACLE_C = os.path.join(ROOT, "oracle", "ntt_oracle.c")

# family -> EMU_PARTS bits (tests/emu/emu.cpp): the transform families need their own direction only (the oracle supplies the
# other one); a product family needs both directions of its field (the column passes either side of the fused middle)
FAMILIES = {
    "gl_fwd": 0x001, "gl_inv": 0x002, "m64_fwd": 0x004, "m64_inv": 0x008, "m32_fwd": 0x010, "m32_inv": 0x020,
    "prod_gl": 0x043, "prod_m64": 0x08C, "prod_m32": 0x130,
}
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


NOCLAMP = "gl_fwd_noclamp"  # the Goldilocks forward family with round 5's defect compiled back in (tests only)


def _build_all(tmp):
    """the nine executables and the no-clamp one, compiled in parallel; returns {name: path}"""
    obj = os.path.join(tmp, "oracle.o")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-c", ORACLE_C, "-o", obj])

    def one(item):
        fam, parts, extra = item
        exe = os.path.join(tmp, fam)
        cmd = ["g++", "-O1", "-g1", "-std=c++17", *SAN, f"-DEMU_PARTS={parts:#x}", *extra, SWEEP, obj, "-fopenmp", "-o", exe]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, f"{fam}: {r.stderr[-2000:]}"
        return fam, exe

    jobs = [(f, p, ()) for f, p in FAMILIES.items()] + [(NOCLAMP, FAMILIES["gl_fwd"], ("-DNTT_EMU_NO_DMA_CLAMP",))]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        return dict(ex.map(one, jobs))


def _sanitizers_available(tmp):
    """g++ with its ASan / UBSan runtimes (this image has them; a toolchain without them skips instead of erroring)"""
    if shutil.which("g++") is None:
        return False
    src = os.path.join(tmp, "probe.cpp")
    with open(src, "w") as f:
        f.write("int main() { return 0; }\n")
    r = subprocess.run(["g++", *SAN, src, "-o", os.path.join(tmp, "probe")], capture_output=True, text=True)
    return r.returncode == 0 and subprocess.run([os.path.join(tmp, "probe")]).returncode == 0


@pytest.fixture(scope="module")
def sweep_exes(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("asan_sweep"))
    if not _sanitizers_available(tmp):
        pytest.skip("no g++ with the ASan / UBSan runtimes")
    return _build_all(tmp)


def _run(exe, fam, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=98", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               OMP_NUM_THREADS="1")
    return subprocess.run([exe, fam, *args], capture_output=True, text=True, env=env, timeout=1500)


def test_sweep_is_clean(sweep_exes):
    """every family, in parallel: no ASan / UBSan report, no wrong word"""
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        results = list(ex.map(lambda kv: (kv[0], _run(kv[1], kv[0])), [kv for kv in sweep_exes.items() if kv[0] != NOCLAMP]))
    total = 0
    for fam, r in results:
        assert r.returncode == 0, f"{fam}: exit {r.returncode}\n{r.stdout[-1000:]}\n{r.stderr[-4000:]}"
        assert "cases clean" in r.stdout
        total += int(r.stdout.split(":")[1].split()[0])
    assert total > 12000  # the sweep did not silently shrink


def test_sweep_sees_the_round5_over_read(sweep_exes):
    """the detector detects: with the ragged-group clamp of phase_dma_issue compiled out (-DNTT_EMU_NO_DMA_CLAMP, tests only) the
    Goldilocks forward family must die in ASan inside phase_dma_issue -- the defect VERDICT r05 reported, heap-buffer-overflow READ
    0 bytes to the right of the caller's exact-size input"""
    r = _run(sweep_exes[NOCLAMP], "gl_fwd", "quick")
    assert r.returncode != 0
    assert "heap-buffer-overflow" in r.stderr and "phase_dma_issue" in r.stderr and "READ of size" in r.stderr
