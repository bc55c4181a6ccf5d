"""The generated gfx950 instruction streams of csrc/gl_asm.h, executed one lane at a time by the generator's own
interpreter (tools/gen_gl_asm.py: simulate) and checked against plain Python integers: random words plus the edge
residues / twiddles that drive every carry, borrow and correction path.  No GPU needed; the GPU parity tests then
show that the hardware agrees with this reading of the instructions."""
import importlib.util
import os
import random

import pytest

from conftest import ROOT

P = 0xFFFFFFFF00000001
R = 1 << 64
RINV = pow(R, -1, P)
EDGE = [0, 1, 2, 0xFFFFFFFF, 0x100000000, 0x100000001, P - 1, P - 2, P - 0xFFFFFFFF, 0xFFFFFFFF00000000, 0xFFFFFFFE00000001,
        0x8000000000000000, 0x7FFFFFFFFFFFFFFF, 0x00000001FFFFFFFF, 0xFFFFFFFEFFFFFFFF]
ANY = EDGE + [P, P + 1, (1 << 64) - 1, (1 << 64) - 2, 0xFFFFFFFF00000002]  # non-canonical 64-bit representatives


@pytest.fixture(scope="module")
def gen():
    spec = importlib.util.spec_from_file_location("gen_gl_asm", os.path.join(ROOT, "tools", "gen_gl_asm.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _env64(vbase, vals):
    """vals: per butterfly slot b a dict name -> 64-bit value for x, y, t"""
    env = {"%[pp]": P}
    for b, v in enumerate(vals):
        for nm, val in v.items():
            env["%%[%s0_%d]" % (nm, b)] = val & 0xFFFFFFFF
            env["%%[%s1_%d]" % (nm, b)] = val >> 32
        env["v%d" % (vbase + 12 * b + 5)] = 0  # the pinned zero half
    return env


def _get(env, nm, b):
    return env["%%[%s0_%d]" % (nm, b)] | (env["%%[%s1_%d]" % (nm, b)] << 32)


def _cases(rng, pool_x, pool_y, n_random):
    for x in pool_x:
        for y in pool_y:
            yield x, y
    for _ in range(n_random):
        yield rng.getrandbits(64), rng.getrandbits(64)


@pytest.mark.parametrize("vbase", [104, 72])
def test_goldilocks_streams(gen, vbase):
    rng = random.Random(vbase)
    tw = EDGE + [rng.randrange(P) for _ in range(6)]
    for kind in ("fwd", "inv", "mul", "invs"):
        lines = gen.stream(kind, 2, vbase)
        canon_in = kind == "fwd"  # forward butterflies take canonical words; the others any 64-bit representative of x
        pool_x = EDGE if canon_in else ANY
        n = 0
        for x, y in _cases(rng, pool_x, EDGE, 300):
            if canon_in:
                x, y = x % P, y % P
            else:
                y = y % P  # inverse: y is a canonical transform-domain word; mul: unused
            for t in (tw if n % 7 == 0 else tw[n % len(tw):][:2]):
                vals = [{"x": x, "y": y, "t": t}, {"x": y if canon_in else x ^ 1, "y": x % P, "t": tw[(n + 3) % len(tw)]}]
                if kind == "mul":
                    vals = [{"x": v["x"], "t": v["t"]} for v in vals]
                e0 = _env64(vbase, vals)
                cs = tw[(n + 5) % len(tw)]  # the folded scale constant of the "invs" stream (any canonical word)
                e0["%[c0]"], e0["%[c1]"] = cs & 0xFFFFFFFF, cs >> 32
                env = gen.simulate(lines, e0)
                for b, v in enumerate(vals):
                    gx = _get(env, "x", b)
                    if kind == "invs":  # u = x*c, w = y*(T*c): both outputs canonical, x any 64-bit representative
                        u, w = v["x"] * cs * RINV % P, v["y"] * v["t"] * RINV % P
                        assert gx == (u + w) % P and _get(env, "y", b) == (u - w) % P, (kind, v, cs)
                        continue
                    if kind == "fwd":
                        assert gx == (v["x"] + v["y"]) % P, (kind, v)
                        assert _get(env, "y", b) == (v["x"] - v["y"]) * v["t"] * RINV % P, (kind, v)
                    elif kind == "inv":
                        w = v["y"] * v["t"] * RINV % P
                        assert gx < (1 << 64) and gx % P == (v["x"] + w) % P, (kind, v)
                        assert _get(env, "y", b) % P == (v["x"] - w) % P, (kind, v)
                    else:
                        assert gx == v["x"] * v["t"] * RINV % P, (kind, v)  # canonical for ANY 64-bit multiplicand
            n += 1


@pytest.mark.parametrize("mode,p", [("lazy", 3329), ("lazy", 998244353), ("lazy", (1 << 30) - 35), ("small", (1 << 30) + 3),
                                    ("small", (1 << 31) - 1), ("any", (1 << 31) + 11), ("any", 3221225473), ("any", (1 << 32) - 5),
                                    ("any", 3), ("small", 5), ("lazy", 7)])
def test_four_byte_word_streams(gen, mode, p):
    rng = random.Random(p)
    pinv = pow(p, -1, 1 << 32)
    rinv = pow(1 << 32, -1, p)
    top = 2 * p if mode == "lazy" else p  # lazy streams keep values in [0, 2p)
    edge = sorted({0, 1, p - 1, p // 2, top - 1, top - 2 if top > 2 else 0, p % top, (p + 1) % top})
    tws = sorted({0, 1, p - 1, p // 2 + 1}) + [rng.randrange(p) for _ in range(4)]
    lines = gen.stream("mul32", 4, mode=mode)  # the scaling sweep: any 32-bit input word (lazy values included), canonical output
    for i, x in enumerate(edge + [top - 1, (1 << 32) - 1 if mode != "lazy" else 2 * p - 1] + [rng.randrange(top) for _ in range(200)]):
        env = {"%[p]": p, "%[npinv]": (-pinv) & 0xFFFFFFFF}
        ts = [tws[(i + b) % len(tws)] for b in range(4)]
        for b in range(4):
            env["%%[x_%d]" % b], env["%%[t_%d]" % b] = (x + b) % (1 << 32) if mode != "lazy" else (x + b) % top, ts[b]
        want = [env["%%[x_%d]" % b] * ts[b] * rinv % p for b in range(4)]
        gen.simulate(lines, env)
        assert [env["%%[x_%d]" % b] for b in range(4)] == want, (mode, p, x)
    for kind in ("fwd32", "inv32"):
        lines = gen.stream(kind, 4, mode=mode)
        cases = [(x, y) for x in edge for y in edge] + [(rng.randrange(top), rng.randrange(top)) for _ in range(300)]
        for i, (x, y) in enumerate(cases):
            env = {"%[p]": p, "%[npinv]": (-pinv) & 0xFFFFFFFF, "%[p2]": (2 * p) & 0xFFFFFFFF}
            vals = []
            for b in range(4):
                xv, yv = (x, y) if b % 2 == 0 else (y, x)
                t = tws[(i + b) % len(tws)]
                vals.append((xv, yv, t))
                env["%%[x_%d]" % b], env["%%[y_%d]" % b], env["%%[t_%d]" % b] = xv, yv, t
            gen.simulate(lines, env)
            for b, (xv, yv, t) in enumerate(vals):
                gx, gy = env["%%[x_%d]" % b], env["%%[y_%d]" % b]
                if kind == "fwd32":
                    wx, wy = (xv + yv) % p, (xv - yv) * t * rinv % p
                else:
                    w = yv * t * rinv % p
                    wx, wy = (xv + w) % p, (xv - w) % p
                if mode == "lazy":
                    assert gx < 2 * p and gy < 2 * p and gx % p == wx and gy % p == wy, (kind, mode, p, xv, yv, t)
                else:
                    assert (gx, gy) == (wx, wy), (kind, mode, p, xv, yv, t)


@pytest.mark.parametrize("p", [0x3FFFFFEE00000001, 0xFFFFFFFC00000001, P, 0xFFFFFFFFFFFFFFC5, (1 << 61) - 1, 3 * 5 * 17 * 257 * 65537 * 641,
                               (1 << 63) + 29, 3329, 3])
@pytest.mark.parametrize("vbase", [104, 72])
def test_general_64bit_modulus_streams(gen, p, vbase):
    """The FieldM64 streams (any odd p < 2^64, Montgomery R = 2^64) against Python integers: fwd64 / inv64 take canonical words,
    the multiplied operand of mul64 may be ANY 64-bit word; edge residues around 0, p, 2^32, 2^63 and the carries they drive."""
    rng = random.Random(p ^ vbase)
    pinv = pow(p, -1, 1 << 64)
    rinv = pow(1 << 64, -1, p)
    edge = sorted({0, 1, 2 % p, p - 1, p - 2 if p > 2 else 0, p // 2, p // 2 + 1, ((1 << 32) - 1) % p, (1 << 32) % p, ((1 << 63) - 1) % p, (1 << 63) % p,
                   0xFFFFFFFF00000000 % p})
    tws = edge + [rng.randrange(p) for _ in range(4)]
    for kind in ("fwd64", "inv64", "mul64", "invs64"):
        lines = gen.stream(kind, 2, vbase)
        pool = [(x, y) for x in edge for y in edge] + [(rng.randrange(p), rng.randrange(p)) for _ in range(200)]
        if kind == "mul64":
            pool += [(v, 0) for v in (p, (1 << 64) - 1, (1 << 63) + 5, p + 1 if p + 1 < (1 << 64) else p)]
        for i, (x, y) in enumerate(pool):
            vals = [{"x": x, "y": y, "t": tws[i % len(tws)]}, {"x": y if kind != "mul64" else x ^ 1, "y": x % p, "t": tws[(i + 5) % len(tws)]}]
            if kind == "mul64":
                vals = [{"x": v["x"], "t": v["t"]} for v in vals]
            env = _env64(vbase, vals)
            env.update({"%[p0]": p & 0xFFFFFFFF, "%[p1]": p >> 32, "%[pi0]": pinv & 0xFFFFFFFF, "%[pi1]": pinv >> 32,
                        "%[vp0]": p & 0xFFFFFFFF, "%[vp1]": p >> 32})
            cs = tws[(i + 2) % len(tws)]  # the folded scale constant of the invs64 stream
            env["%[c0]"], env["%[c1]"] = cs & 0xFFFFFFFF, cs >> 32
            env = gen.simulate(lines, env)
            for b, v in enumerate(vals):
                gx = _get(env, "x", b)
                if kind == "invs64":
                    u, w = v["x"] * cs * rinv % p, v["y"] * v["t"] * rinv % p
                    assert gx == (u + w) % p and _get(env, "y", b) == (u - w) % p, (kind, p, v, cs)
                    continue
                if kind == "fwd64":
                    assert gx == (v["x"] + v["y"]) % p and _get(env, "y", b) == (v["x"] - v["y"]) * v["t"] * rinv % p, (kind, p, v)
                elif kind == "inv64":
                    w = v["y"] * v["t"] * rinv % p
                    assert gx == (v["x"] + w) % p and _get(env, "y", b) == (v["x"] - w) % p, (kind, p, v)
                else:
                    assert gx == v["x"] * v["t"] * rinv % p, (kind, p, v)
