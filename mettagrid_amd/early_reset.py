"""EarlyResetHandler draws for a whole batch of envs at once.

The reference ends every env's FIRST episode early at ``np.random.default_rng(seed).integers(1, max_steps + 1)``
(python/src/mettagrid/envs/early_reset_handler.py:6-22), one generator per env.  Constructing 65 536 generators costs
0.7 s per ``reset()``; this module restates what that expression computes — ``SeedSequence`` pool mixing
(numpy/random/bit_generator.pyx), PCG64 seeding and its XSL-RR output (numpy/random/src/pcg64), the buffered 32-bit
Lemire draw (numpy/random/src/distributions/distributions.c) — on arrays of seeds.  tests/test_host_logic.py checks it
against numpy itself.
"""
import numpy as np
M32 = np.uint64(0xFFFFFFFF)
def _u32(x): return x & M32
def seedseq_pool(seeds):
    """SeedSequence(int seed < 2**32).pool for many seeds at once (numpy/random/bit_generator.pyx)."""
    n = len(seeds)
    INIT_A, MULT_A = 0x43b0d7e5, 0x931e8875
    MIX_L, MIX_R = np.uint64(0xca01f9dd), np.uint64(0x4973f715)
    hc = [INIT_A]
    def hashmix(v):
        v = _u32(v ^ np.uint64(hc[0]))
        hc[0] = (hc[0] * MULT_A) & 0xFFFFFFFF
        v = _u32(v * np.uint64(hc[0]))
        return v ^ (v >> np.uint64(16))
    def mix(x, y):
        r = _u32(MIX_L * x - MIX_R * y)
        return r ^ (r >> np.uint64(16))
    ent = [seeds.astype(np.uint64)] + [np.zeros(n, np.uint64)] * 3
    pool = [hashmix(ent[i]) for i in range(4)]
    for s in range(4):
        for d in range(4):
            if s != d:
                pool[d] = mix(pool[d], hashmix(pool[s]))
    return pool
def generate_state8(pool):
    INIT_B, MULT_B = 0x8b51f9dd, 0x58f38ded
    hc = INIT_B
    out = []
    for i in range(8):
        v = _u32(pool[i % 4] ^ np.uint64(hc))
        hc = (hc * MULT_B) & 0xFFFFFFFF
        v = _u32(v * np.uint64(hc))
        out.append(v ^ (v >> np.uint64(16)))
    return out
def mul64(a, b):
    """(hi, lo) of the 128-bit product of two uint64 arrays."""
    a0, a1, b0, b1 = a & M32, a >> np.uint64(32), b & M32, b >> np.uint64(32)
    p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
    mid = (p00 >> np.uint64(32)) + (p01 & M32) + (p10 & M32)
    lo = (p00 & M32) | (mid << np.uint64(32))
    hi = p11 + (p01 >> np.uint64(32)) + (p10 >> np.uint64(32)) + (mid >> np.uint64(32))
    return hi, lo
def add128(ah, al, bh, bl):
    lo = al + bl
    return ah + bh + (lo < al).astype(np.uint64), lo
MH, ML = np.uint64(0x2360ED051FC65DA4), np.uint64(0x4385DF649FCCF645)
def step(sh, sl, ih, il):
    hi, lo = mul64(sl, ML)
    hi = hi + sl * MH + sh * ML
    return add128(hi, lo, ih, il)
def first_integers(seeds, high):
    """[int(np.random.default_rng(int(s)).integers(1, high + 1)) for s in seeds], vectorised (PCG64 + Lemire)."""
    seeds = np.asarray(seeds, dtype=np.uint64)
    with np.errstate(over="ignore"):
        w = generate_state8(seedseq_pool(seeds))
        v = [w[2 * i] | (w[2 * i + 1] << np.uint64(32)) for i in range(4)]
        ih = (v[2] << np.uint64(1)) | (v[3] >> np.uint64(63)); il = (v[3] << np.uint64(1)) | np.uint64(1)
        sh, sl = step(np.zeros_like(v[0]), np.zeros_like(v[0]), ih, il)
        sh, sl = add128(sh, sl, v[0], v[1])
        sh, sl = step(sh, sl, ih, il)
        sh, sl = step(sh, sl, ih, il)      # next64: step, then output XSL-RR
        x = sh ^ sl
        rot = sh >> np.uint64(58)
        out = (x >> rot) | (x << ((np.uint64(64) - rot) & np.uint64(63)))
        r32 = out & M32
        rng = high - 1
        if rng == 0:
            return np.ones(len(seeds), np.uint32)
        excl = np.uint64(rng + 1)
        m = r32 * excl
        res = (m >> np.uint64(32)).astype(np.int64) + 1
        left = m & M32
        thr = np.uint64((0xFFFFFFFF - rng) % (rng + 1))
        redo = np.nonzero((left < excl) & (left < thr))[0]
    for i in redo:   # Lemire rejection (probability < high / 2**32): the scalar generator
        res[i] = int(np.random.default_rng(int(seeds[i])).integers(1, high + 1))
    return res.astype(np.uint32)
