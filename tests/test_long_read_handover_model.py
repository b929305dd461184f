"""Model of the hand-over protocol of the long-read forward pipeline (generic_forward_skew, genomeworks_amd/csrc/poa_device.h:
MwShared::hand). Eight wavefronts work through the rows of a band in order; wavefront w publishes, when it has finished row r,
ONE entry {its last cell of row r, r} at slot r & 7; its right neighbour needs that cell for its own row r and looks for it in
that slot ("ready" = the slot's row >= r); a wavefront stores row r (and with it reuses the ring slot of row r - R and, eight
rows later, the hand-over slot) only once its right neighbour has finished row r - kSkLead - 1. A row the band does not touch in
a wavefront's block is skipped: nothing holds the wavefront back there, and it records the row in a word of its own
(MwShared::skipped), not in the entries -- "finished row r" is max(entry row, skipped) >= r. The model runs random
interleavings of the wavefronts' steps and checks what the kernel relies on: a reader that finds the slot ready reads the cell
of exactly its row; nobody deadlocks; a wavefront is never more than kSkLead + 1 rows ahead of its right neighbour."""
import random

K_WAVES, K_LEAD, SLOTS = 8, 4, 8


def run(seed, rows, first_block_rows, favourite=None, skipping=frozenset(), skips_write_entries=False):
    rng = random.Random(seed)
    # hand[w][slot] = (cell, row); row 0 counts as finished everywhere
    hand = [[(0, 0)] * SLOTS for _ in range(K_WAVES)]
    skipped = [0] * K_WAVES               # last row skipped, per wavefront
    finished = lambda w, r: max(hand[w][r % SLOTS][1], skipped[w]) >= r
    done = [0] * K_WAVES                  # rows finished per wavefront (the model's ground truth)
    got = [dict() for _ in range(K_WAVES)]  # row -> cell the wavefront consumed from its left neighbour
    phase = [0] * K_WAVES                 # 0 = before the carry, 1 = carry consumed, about to store + publish
    cell = lambda w, r: 1000 * w + r      # what wavefront w's last cell of row r "is"
    steps = 0
    while min(done) < rows:
        steps += 1
        assert steps < 200 * rows * K_WAVES, "no progress: deadlock"
        w = favourite if (favourite is not None and rng.random() < 0.8) else rng.randrange(K_WAVES)  # a fast wavefront
        r = done[w] + 1
        if r > rows:
            continue
        left, right = (w - 1) % K_WAVES, (w + 1) % K_WAVES
        if (w, r) in skipping:  # the band does not touch the wavefront's block in this row: note it and go on, unchecked
            if skips_write_entries:  # (what rounds 2-4 did: the skipped rows lap the reader of an earlier row's entry)
                hand[w][r % SLOTS] = (0, r)
            else:
                skipped[w] = r
            done[w] = r
            continue
        needs_carry = (w, r) not in first_block_rows  # the first block of a row has no left neighbour in that row
        if phase[w] == 0:
            # the predecessor rows' boundary cells are the left neighbour's: it must be past row r - 1
            if not finished(left, r - 1):
                continue
            if needs_carry:
                c, row = hand[left][r % SLOTS]
                if row < r:
                    continue  # not there yet: poll again later
                assert row == r, "slot %d of wave %d holds row %d while wave %d is at row %d" % (r % SLOTS, left, row, w, r)
                assert c == cell(left, r)
                got[w][r] = c
            phase[w] = 1
        else:
            # ring space: the right neighbour must have finished row r - kSkLead - 1 (read from ITS entries)
            target = r - K_LEAD - 1
            if target > 0 and not finished(right, target):
                continue
            hand[w][r % SLOTS] = (cell(w, r), r)
            done[w] = r
            phase[w] = 0
            assert done[w] - done[right] <= K_LEAD + 1 or done[right] >= rows
    for w in range(K_WAVES):
        for r in range(1, rows + 1):
            if (w, r) not in first_block_rows and (w, r) not in skipping:
                assert got[w][r] == cell((w - 1) % K_WAVES, r)


def test_every_carry_is_the_row_it_was_asked_for():
    for seed in range(40):
        rng = random.Random(1000 + seed)
        rows = rng.choice([1, 7, 8, 9, 40, 200])
        # the first block of the band moves from wavefront to wavefront as the band moves right
        # (and the wavefront to its left, whose block the band has passed or not reached, skips the row: seven blocks of a row)
        first, first_rows, skipped = rng.randrange(K_WAVES), set(), set()
        for r in range(1, rows + 1):
            if rng.random() < 0.05:
                first = (first + 1) % K_WAVES
            first_rows.add((first, r))
            if rng.random() < 0.9:
                skipped.add(((first - 1) % K_WAVES, r))
        run(seed, rows, first_rows, favourite=rng.choice([None, first, rng.randrange(K_WAVES)]), skipping=frozenset(skipped))


def test_skipped_rows_must_not_write_hand_over_entries():
    # round 4: the model found that a wavefront which works row r and then skips rows r + 1 .. r + 8 (the band has passed its
    # block) overwrote slot r & 7 while its right neighbour might still be rows behind; skipped rows have their own word now
    tripped = False
    for seed in range(200):
        rng = random.Random(seed)
        skipped = frozenset((3, r) for r in range(30, 61))
        try:
            run(seed, 60, {(4, r) for r in range(30, 61)} | {(0, r) for r in range(1, 30)}, favourite=3, skipping=skipped,
                skips_write_entries=True)
        except AssertionError:
            tripped = True
            break
    assert tripped
    for seed in range(50):
        run(seed, 60, {(4, r) for r in range(30, 61)} | {(0, r) for r in range(1, 30)}, favourite=3,
            skipping=frozenset((3, r) for r in range(30, 61)))


def test_a_stale_slot_is_never_mistaken_for_a_ready_one():
    # without the ring-space rule a writer could lap its reader: the model must notice (the assertion inside run() is live)
    global K_LEAD
    keep = K_LEAD
    try:
        K_LEAD = 12  # more than the eight slots can cover
        tripped = False
        for seed in range(60):
            try:
                run(seed, 60, {(0, r) for r in range(1, 61)}, favourite=0, skipping=frozenset((7, r) for r in range(1, 61)))
            except AssertionError:
                tripped = True
                break
        assert tripped
    finally:
        K_LEAD = keep
