"""BASELINE.json's configurations as data: what one "operation" of each is, its algorithmic bytes and its
butterfly count.  Shared by tools/config_profile.py (the program rocprofv3 wraps), tools/config_summary.py (the reduction),
tools/bench_configs.py (the timed lines) and tools/design_table.py."""
GOLD = 0xFFFFFFFF00000001

CONFIGS = {
    "cfg2": dict(name="BASELINE config 2: N=2^12 forward, 32-bit prime 3221225473, batch 1024", logn=12, p=3221225473, g=5, wb=4,
                 batch=1024, op="forward", kind=0),
    "cfg2_sat": dict(name="config 2's shape at a saturating batch: N=2^12 forward, 32-bit prime 3221225473, batch 65536", logn=12,
                     p=3221225473, g=5, wb=4, batch=65536, op="forward", kind=0),
    "cfg3": dict(name="BASELINE config 3 (headline): N=2^16 forward, Goldilocks, batch 4096", logn=16, p=GOLD, g=7, wb=8, batch=4096,
                 op="forward", kind=0),
    "cfg4": dict(name="BASELINE config 4: N=2^20 negacyclic product (NTT -> pointwise -> iNTT), Goldilocks, batch 512", logn=20, p=GOLD,
                 g=7, wb=8, batch=512, op="polymul", kind=2),
    # config 5 is an 8-GPU job (65536 rows = 8 x 8192, no data-path collective): what ONE GPU of it does is this shard
    "cfg5_shard": dict(name="BASELINE config 5's per-GPU shard: N=2^16 forward, Goldilocks, 8192 of the job's 65536 rows on one MI355X", logn=16,
                       p=GOLD, g=7, wb=8, batch=8192, op="forward", kind=0),
}


def algorithmic_bytes(c: dict) -> int:
    """per operation: a transform reads and writes every word once (2 N words per polynomial); the product is priced on the
    UNFUSED pipeline SURVEY 8(d) names: fwd(a) 2N + fwd(b) 2N + pointwise 3N + inverse 2N = 9 N words per product."""
    n = 1 << c["logn"]
    return (9 if c["op"] == "polymul" else 2) * n * c["wb"] * c["batch"]


def butterflies(c: dict) -> int:
    """per operation: batch * N/2 * log2 N per transform; three transforms per product."""
    n = 1 << c["logn"]
    return (3 if c["op"] == "polymul" else 1) * c["batch"] * (n // 2) * c["logn"]
