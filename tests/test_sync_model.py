"""The ordering argument of render_kernel_coop2's cooperative calls (csrc/rf_coop2.h, "SYNCHRONISATION"), mechanised.

A workgroup's waves execute the same sequence of barriers (every branch that contains one is block-uniform), so two
accesses by different waves are ordered exactly when the waves have passed a different number of barriers before
them.  This file writes down, per wave and sample, which LDS arrays every phase of the kernel reads, writes or
updates atomically -- the protocol of rf_coop2.h, array by array -- and checks for every pair of phases that can meet
in the same barrier epoch that they do not conflict.  It is a model next to the code, not the code: what it buys is
that the argument in the header's comment is checked over every combination of call outcomes (no stragglers / one
round / two rounds) instead of being read, and that it demonstrably finds the race of the round-3 form of the kernel
(no barrier after the collect, one counter), which tests/test_gpu_parity.py::test_delayed_waves_change_nothing shows
on the hardware."""

import itertools

import pytest

R, W, A = "read", "write", "atomic"


class Wave:
    """Collects (epoch, phase, array, kind) of one wave; barrier() starts the next epoch."""

    def __init__(self, index):
        self.index = index
        self.epoch = 0
        self.events = []

    def touch(self, phase, array, kind):
        self.events.append((self.epoch, phase, array, kind))

    def barrier(self):
        self.epoch += 1


# accesses of one phase that cannot collide with each other although several waves make them in one epoch: the slots
# are dealt by an atomic (park, round-1 survivors), indexed by the worker's own thread (rounds), or the owner's own
# (collect); atomics on one word are ordered by the LDS unit
SELF_COMPATIBLE = {"park", "round1", "round2", "finish", "collect", "count", "stage", "store"}


def overlap(array_a, array_b):
    """"state0/quarter2" is a wave's own quarter of state0: it overlaps state0 as a whole, not another quarter."""
    whole_a, _, part_a = array_a.partition("/")
    whole_b, _, part_b = array_b.partition("/")
    return whole_a == whole_b and (not part_a or not part_b or part_a == part_b)


def conflicts(a, b):
    (_, phase_a, array_a, kind_a), (_, phase_b, array_b, kind_b) = a, b
    if not overlap(array_a, array_b) or (kind_a == R and kind_b == R) or (kind_a == A and kind_b == A):
        return False
    if phase_a == phase_b and phase_a in SELF_COMPATIBLE:
        return False
    return True


def races(waves):
    found = []
    for x, y in itertools.combinations(waves, 2):
        for a in x.events:
            for b in y.events:
                if a[0] == b[0] and conflicts(a, b):
                    found.append((x.index, a, y.index, b))
    return found


def cooperative_call(wave, dim, parity, counter, outcome, fenced, sample):
    """coop_finish2m<DIM, *, FENCED>: `outcome` is block-uniform: "none" (no stragglers), "one" (the packed entries are
    finished in one round), "two" (sphere calls with more than 64 entries: packing round + round 2)."""
    state, other = f"state{parity}", f"state{parity ^ 1}"
    wave.touch("park", counter, A)
    wave.touch("park", state, W)
    wave.barrier()  # B1
    wave.touch("count", counter, R)
    if outcome == "none":
        return
    if outcome == "two":
        assert dim == 3
        wave.touch("round1", state, R)
        wave.touch("round1", state, W)
        wave.touch("round1", "words4", W)
        wave.touch("round1", "words2", W)
        wave.touch("round1", "cnt2", A)
        wave.touch("round1", other, W)
        wave.touch("round1", "owner", W)
        wave.barrier()  # B2
        wave.touch("count", "cnt2", R)
        wave.touch("round2", other, R)
        wave.touch("round2", "owner", R)
        wave.touch("round2", state, W)
        wave.touch("round2", "words4", W)
        wave.touch("round2", "words2", W)
        wave.barrier()  # B3
        if wave.index == 0:
            wave.touch("reset", "cnt2", W)
    else:
        wave.touch("finish", state, R)
        wave.touch("finish", state, W)
        wave.touch("finish", "words4", W)
        if dim == 3:
            wave.touch("finish", "words2", W)
        wave.barrier()  # B3
    if wave.index == 0:
        wave.touch("reset", counter, W)
    wave.touch("collect", state, R)
    wave.touch("collect", "words4", R)
    if dim == 3:
        wave.touch("collect", "words2", R)
    if fenced:
        wave.barrier()  # B4


def sample_loop(outcomes, in_wave_disc, fenced, alternate, n_waves=3, passes=1):
    """The cooperative part of the sample loop for every combination in `outcomes` (one entry per sample: the sphere
    call's outcome, and for the block-wide disc call the disc call's).  passes = 2: the two-pass instance of the fused
    environment step (the same samples again after the first pass's frame has gone through the stage)."""
    waves = [Wave(i) for i in range(n_waves)]
    for wave in waves:
      for p in range(passes):
        if wave.index == 0:  # threads 0 .. 2 clear the counters
            for counter in ("cnt0", "cnt1", "cnt2"):
                wave.touch("clear", counter, W)
        wave.barrier()  # the pass's prologue: counters cleared, then __syncthreads
        for k, (disc, sphere) in enumerate(outcomes):
            if in_wave_disc:
                # disc_tails_wave: the wave's own quarter of state[0], nothing else
                # (a part of state[0] as round 1 of the sphere call uses it: `other`)
                wave.touch("disc-in-wave", f"state0/quarter{wave.index}", W)
                wave.touch("disc-in-wave", f"state0/quarter{wave.index}", R)
            else:
                cooperative_call(wave, 2, 0, "cnt0", disc, False, k)
            counter = f"cnt{k & 1}" if (in_wave_disc and alternate) else "cnt1"
            cooperative_call(wave, 3, 1, counter, sphere, in_wave_disc and fenced, k)
        wave.barrier()  # end of the loop: the cooperative arrays become the frame stage
        wave.touch("stage", "words4", W)  # every thread its own bytes
        wave.barrier()
        wave.touch("store", "words4", R)  # rows of the tile, read by other threads than wrote them
    return waves


SPHERE = ("none", "one", "two")
DISC = ("none", "one")


def all_outcomes(samples, with_disc):
    per_sample = list(itertools.product(DISC if with_disc else ("none",), SPHERE))
    return itertools.product(per_sample, repeat=samples)


def test_block_wide_disc_call_instances_are_ordered_by_the_alternation():
    """Frames that are not powers of two: disc call (parity 0) and sphere call (parity 1) alternate, no B4."""
    for outcomes in all_outcomes(3, with_disc=True):
        found = races(sample_loop(outcomes, in_wave_disc=False, fenced=False, alternate=False))
        assert not found, (outcomes, found[:3])


def test_in_wave_disc_instances_are_ordered_by_the_alternating_counter_and_b4():
    """Power-of-two frames (the benchmarked instance): no barrier in the disc phase; cnt[k & 1] and B4."""
    for outcomes in all_outcomes(4, with_disc=False):
        found = races(sample_loop(outcomes, in_wave_disc=True, fenced=True, alternate=True))
        assert not found, (outcomes, found[:3])


@pytest.mark.parametrize("fenced,alternate", [(False, False), (True, False), (False, True)])
def test_the_checker_finds_what_each_measure_is_there_for(fenced, alternate):
    """The round-3 form (neither measure), and each measure alone: the model must report the unordered pairs --
    the collect reads and thread 0's reset against the next park without B4; a late wave's read of the counter
    against the next park's atomics without the alternation (in a block that left its call after B1)."""
    found = set()
    for outcomes in all_outcomes(3, with_disc=False):
        for _, a, _, b in races(sample_loop(outcomes, in_wave_disc=True, fenced=fenced, alternate=alternate)):
            found.add(tuple(sorted([(a[1], a[2]), (b[1], b[2])])))
    if not fenced:
        assert (("collect", "state1"), ("park", "state1")) in found  # a slow wave's collect, a fast wave's next park
    if not fenced and not alternate:
        assert (("park", "cnt1"), ("reset", "cnt1")) in found  # thread 0's late reset, the next park's atomics
    if not alternate:
        assert (("count", "cnt1"), ("park", "cnt1")) in found  # the empty-list exit: no B4 there by design
    assert found


@pytest.mark.parametrize("in_wave_disc", [True, False])
def test_the_two_passes_of_the_fused_step_need_no_barrier_of_their_own(in_wave_disc):
    """render_kernel_coop2<.., TWO = true>: a block's second pass re-uses every cooperative array and the stage.  The
    barrier at the top of every pass (behind the counter clears) is enough: the first pass's row stores (reads of the
    stage) and the clears share an epoch without touching the same words, and everything else of the second pass comes
    after it."""
    for outcomes in all_outcomes(2, with_disc=not in_wave_disc):
        found = races(sample_loop(outcomes, in_wave_disc, fenced=True, alternate=True, passes=2))
        assert not found, (outcomes, found[:3])


# --- the model against the sources: which synchronisation-relevant steps the kernels take, in which order ---------------

import os
import re

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reinfocus_amd", "csrc")
STEPS = [("clear", r"lds\.cnt\[tid\] = 0"), ("barrier", r"__syncthreads\(\)"), ("disc-in-wave", r"disc_tails_wave\(lds\.state\[0\]"),
         ("disc-call", r"coop_finish2m<2, kWaveSlots, false>\(lds, 0, &lds\.cnt\[0\]"),
         ("sphere-counter", r"&lds\.cnt\[kDiscInWave \? \(k & 1\) : 1\]"),
         ("sphere-call", r"coop_finish2m<3, kWaveSlots, kDiscInWave>\(lds, 1, sphere_cnt"),
         ("stage-write", r"sb\[slot \* 3 \+ 0\] ="), ("stage-read", r"= stage\[r \* kRowDw \+ d\w*\]"),
         ("own-colour", r"lds_colour\[j\]\[0\]\[tid\] =")]


def protocol_steps(header, start, end):
    """The synchronisation-relevant steps of the code between two markers of a header, in source order."""
    text = open(os.path.join(CSRC, header)).read()
    body = text[text.index(start):text.index(end, text.index(start))]
    body = re.sub(r"//[^\n]*", "", body)  # (comments talk about barriers too)
    found = []
    for name, pattern in STEPS:
        found += [(m.start(), name) for m in re.finditer(pattern, body)]
    return [name for _, name in sorted(found)]


# what sample_loop() above models, as the source has to spell it: counters cleared, barrier, [per sample: the disc phase
# in one of its two forms, the sphere call on the alternating counter], barrier, stage written, barrier, stage read.
# (own-colour: the colour sums of two pixel sets live in LDS at the thread's own index: no other thread touches them)
EXPECTED = ["clear", "barrier", "own-colour", "disc-in-wave", "disc-call", "sphere-counter", "sphere-call", "own-colour", "barrier",
            "stage-write", "barrier", "stage-read"]


def test_the_fast_path_kernel_spells_the_modelled_protocol():
    got = protocol_steps("rf_coop2.h", "__device__ __forceinline__ void render_tile_coop2(", "template <bool POW2, int LENS, int WX = kWavesX")
    assert got == EXPECTED, got


def test_the_one_shape_general_kernel_spells_the_same_protocol():
    """render_general_one_kernel (rf_general_one.h) uses the cooperative machinery of rf_coop2.h -- disc_tails_wave or
    the block-wide disc call, the fenced sphere call on the alternating counter, the frame stage in words4 -- in the
    same order as render_tile_coop2, so the model checks above cover it; what it adds touches no shared LDS word: the
    abstention flags are lane masks in scalar registers and a NaN in the thread's own colour sum, the fix-up list is
    global memory (one atomic per abstaining pixel, read by the next kernel on the stream)."""
    got = protocol_steps("rf_general_one.h", "render_general_one_kernel(GeneralOneArgs ra)", "// The listed pixels, literally")
    assert got == EXPECTED, got
    text = open(os.path.join(CSRC, "rf_general_one.h")).read()
    body = text[text.index("render_general_one_kernel(GeneralOneArgs ra)"):text.index("// The listed pixels, literally")]
    assert "__shared__ CoopLds2 lds;" in body and body.count("__shared__") == 2  # (the cooperative arrays + the own colour sums)
    assert "redo_append(ra," in body and "atomicAdd(&lds" not in body
    for in_wave_disc in (True, False):  # the two instances (power-of-two frames / others), single pass
        for outcomes in all_outcomes(3, with_disc=not in_wave_disc):
            assert not races(sample_loop(outcomes, in_wave_disc, fenced=True, alternate=True))
