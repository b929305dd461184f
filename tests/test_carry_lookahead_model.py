"""The multi-word addition of the wave-wide Myers kernels (gwhip_myers.hip: group_advance, hirschberg_wave_kernel,
hirschberg_levels_kernel). Lane i adds its word alone and reports g_i ("my addition overflowed") and p_i ("my sum is all
ones: a carry-in would pass through"); the carry into every lane then comes from ONE 64-bit integer addition on the scalar
unit: cin = ((G | P) + G) ^ P with the bits of every group's / segment's last lane cleared in G and P, so that no carry
crosses into the next group. This checks that formula against the lane-by-lane ripple it replaces: plain Python."""
import random

MASK = (1 << 64) - 1


def ripple(g, p, cut):
    cin, c = 0, 0
    for lane in range(64):
        cin |= c << lane
        c = ((g >> lane) & 1) | (((p >> lane) & 1) & c)
        if (cut >> lane) & 1:       # last lane of a group: its carry-out goes nowhere
            c = 0
    return cin


def lookahead(g, p, cut):
    gm, pm = g & ~cut & MASK, p & ~cut & MASK
    return (((gm | pm) + gm) & MASK) ^ pm


def test_one_scalar_addition_gives_every_lane_its_carry_in():
    rng = random.Random(11)
    for case in range(20000):
        style = case % 4
        g = rng.getrandbits(64) if style else rng.getrandbits(64) & rng.getrandbits(64)
        p = rng.getrandbits(64) & ~g & MASK                      # a word cannot both overflow and be all ones
        if style == 2:
            p = ~g & MASK                                        # long runs of propagate bits
        if style == 0:
            cut = 0x8080808080808080                             # groups of eight lanes
        elif style == 1:
            cut = sum(1 << (6 * j + 5) for j in range(10))       # groups of six, four idle lanes
        else:
            cut = rng.getrandbits(64) | (1 << 63)                # segments of any length
        assert lookahead(g, p, cut) == ripple(g, p, cut)
