"""The carry-lookahead the eight-lanes-per-pair banded Myers kernel uses for its multi-word addition
(genomeworks_amd/csrc/gwhip_myers.hip: group_advance): every lane adds its 32-bit word alone and reports "generates a carry"
(g) / "would propagate one" (p); with G = ballot(g), P = ballot(p) and TOP = the last lane of every group,

    carry_in = (((G | P) & ~TOP) + (G & ~TOP)) ^ (P & ~TOP)

gives every lane its carry-in at once, and no carry crosses from one group of lanes into the next. Checked here against the
word-by-word ripple on random operands rich in all-ones words (the propagate case), for group sizes 4, 8 and 16."""
import random

MASK32 = 0xFFFFFFFF
MASK64 = (1 << 64) - 1


def ripple(a, b, group):
    """carry into every lane of the multi-word additions a + b of each group of `group` lanes (lane 0 of a group: 0)"""
    carries = []
    c = 0
    for lane, (x, y) in enumerate(zip(a, b)):
        if lane % group == 0:
            c = 0
        carries.append(c)
        c = 1 if x + y + c > MASK32 else 0
    return carries


def lookahead(a, b, group):
    gen = prp = top = 0
    for lane, (x, y) in enumerate(zip(a, b)):
        s0 = (x + y) & MASK32
        if s0 < x:
            gen |= 1 << lane
        if s0 == MASK32:
            prp |= 1 << lane
        if lane % group == group - 1:
            top |= 1 << lane
    A = (gen | prp) & ~top & MASK64
    B = gen & ~top & MASK64
    cin = ((A + B) & MASK64) ^ (prp & ~top & MASK64)
    return [(cin >> lane) & 1 for lane in range(len(a))]


def test_generate_and_propagate_are_disjoint():
    rng = random.Random(1)
    for _ in range(20000):
        x = rng.choice([0, MASK32, rng.getrandbits(32)])
        y = rng.choice([0, MASK32, MASK32 - x, rng.getrandbits(32)])
        s0 = (x + y) & MASK32
        assert not (s0 < x and s0 == MASK32)


def test_carry_lookahead_equals_ripple():
    rng = random.Random(2)
    special = [0, 1, MASK32, MASK32 - 1, 0x80000000, 0x7FFFFFFF]
    for group in (4, 8, 16):
        for _ in range(4000):
            a, b = [], []
            for _lane in range(64):
                x = rng.choice(special) if rng.random() < 0.5 else rng.getrandbits(32)
                r = rng.random()
                if r < 0.35:
                    y = MASK32 - x          # x + y = all ones: propagates
                elif r < 0.5:
                    y = (MASK32 - x + 1) & MASK32  # x + y = 2^32 (or 0): generates with a zero sum
                elif r < 0.7:
                    y = rng.choice(special)
                else:
                    y = rng.getrandbits(32)
                a.append(x)
                b.append(y)
            assert lookahead(a, b, group) == ripple(a, b, group)


def test_myers_xh_uses_the_same_carries():
    """Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq over a multi-word column: the lookahead carries reproduce the big-integer sum."""
    rng = random.Random(3)
    for _ in range(2000):
        words = 8
        eq = [rng.getrandbits(32) if rng.random() < 0.7 else MASK32 for _ in range(words)]
        pv = [rng.getrandbits(32) if rng.random() < 0.6 else MASK32 for _ in range(words)]
        a = [e & p for e, p in zip(eq, pv)]
        cin = lookahead(a + [0] * 56, pv + [0] * 56, 8)[:words]
        big_a = sum(x << (32 * k) for k, x in enumerate(a))
        big_p = sum(x << (32 * k) for k, x in enumerate(pv))
        big_s = big_a + big_p
        for k in range(words):
            assert ((a[k] + pv[k] + cin[k]) & MASK32) == (big_s >> (32 * k)) & MASK32
