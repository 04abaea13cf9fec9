"""CPU model of how the tile decoders follow the reference's chain of table steps (fdeflate_amd/csrc/
inflate_stream.h: tile_step, STEP_START / STEP_SECOND / STEP_UNKNOWN).

The reference decodes a block table step by table step; a step is one symbol or TWO literals whose codes fit the
table index together (src/huffman.rs:110-130), and when the input runs out inside a pair neither literal is
produced (src/decompress.rs:852).  A tile ends at a symbol; to let the exact serial decoder take over there it
has to know whether that symbol starts a step.  Per symbol the state changes as
    pair entry at the symbol : START -> SECOND, SECOND -> START   (a swap)
    anything else            : -> START                           (a reset)
A lane folds its symbols into (reset seen, parity of the swaps behind the last reset), the wavefront folds the
lanes with two ballots: the last lane with a reset, the parity of the swap-parities from that lane on.  This test
checks that fold against the plain sequential walk on random symbol sequences cut into random lanes."""
import random

START, SECOND, UNKNOWN = 1, 2, 0


def sequential(state, pairs):
    for p in pairs:
        if state == UNKNOWN:
            state = UNKNOWN if p else START
        else:
            state = SECOND if (state == START and p) else START
    return state


def lane_fold(pairs):
    reset, flip = False, 0
    for p in pairs:
        if p:
            flip ^= 1
        else:
            reset, flip = True, 0
    return reset, flip


def wave_fold(state, lanes):
    folds = [lane_fold(l) for l in lanes]
    R = [i for i, (r, f) in enumerate(folds) if r]
    if R:
        lr = R[-1]
        odd = sum(f for (r, f) in folds[lr:]) & 1
        return SECOND if odd else START
    odd = sum(f for (r, f) in folds) & 1
    if state == UNKNOWN:
        return UNKNOWN
    return (SECOND if state == START else START) if odd else state


def test_lane_and_wavefront_fold_equal_the_sequential_walk():
    rnd = random.Random(12)
    for trial in range(4000):
        n = rnd.randrange(0, 200)
        density = rnd.choice((0.0, 0.3, 0.7, 0.95, 1.0))
        pairs = [rnd.random() < density for _ in range(n)]
        cuts = sorted(rnd.randrange(0, n + 1) for _ in range(rnd.randrange(0, 64)))
        lanes, a = [], 0
        for c in cuts + [n]:
            lanes.append(pairs[a:c])
            a = c
        for state in (START, SECOND, UNKNOWN):
            assert wave_fold(state, lanes) == sequential(state, pairs), (trial, state)


def test_a_symbol_that_pairs_with_nothing_puts_any_walk_in_step():
    # what resync_to_step_start relies on: behind a symbol whose entry is no pair a step starts, whatever was before
    rnd = random.Random(3)
    for trial in range(500):
        pairs = [rnd.random() < 0.8 for _ in range(rnd.randrange(1, 60))] + [False]
        assert {sequential(s, pairs) for s in (START, SECOND, UNKNOWN)} == {START}
