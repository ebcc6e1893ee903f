// rf_coop2.h -- render_kernel_coop2<POW2, LENS>: the default render kernel.
//
// Same arithmetic, same pixel <-> RNG-state mapping and the same idea as render_kernel_coop
// (rf_kernels.h) -- lanes whose rejection loop is not done after their in-wave attempts hand their
// RNG state to a packed list in LDS that full waves finish -- with three refinements, each
// measured at the headline config (G samples/s; render_kernel_coop: 122):
//
//  * kSets = 3 pixels per thread (tile 128 x 6: a thread owns (x, y + 2 j), j < 3).  57 % of
//    the one-pixel kernel's wave-cycles are barrier waits for the tails; now every wave does the
//    in-wave work of three pixel sets between two barriers and one cooperative call serves all
//    three.  2 / 3 / 4 sets: 133 / 138 / 114.  Spilling is not an option even where it is cheap in
//    time (every spilled dword of a wave is 256 B of scratch traffic: the first 3-set build moved
//    2.5x the algorithmic HBM bytes), so the kernel (a) re-derives the per-thread geometry inside
//    the sample loop from an index the compiler cannot see through -- the loop invariants were
//    what got spilled --, (b) moves block-uniform values computed with vector instructions into
//    scalar registers, (c) keeps the colour accumulators of two sets in LDS and (d) stages the
//    frame through a cooperative array: 72 VGPRs (7 waves per SIMD), no scratch, 20.5 KB LDS.
//  * a packing round (RF_TWO_ROUNDS): when the list needs more than one wave, its entries first
//    make a bounded number of attempts on as many waves as they fill and the survivors are packed
//    again, so that a single wave runs the sparse end of the tail: +1.5 %.
//  * with that round in place, the second in-wave sphere attempt (only 48 % of a hit wave's lanes
//    take part) moves into it: one in-wave attempt, two in the packing round: 146.6.  A block
//    whose tile lies inside the target then has ~366 stragglers for the 256 entries of the list
//    (the rest finish in place), so each block watches its own count and returns to two in-wave
//    attempts while its list would overflow: 146.3, and 121.6 at 300 px / 100 spp (always one:
//    147.0 / 107.7; always two: 142.5 / 119.5).
//
// Round 3 (k env-steps/s on the driver's metric): lane predicates as scalar-register masks
// (RF_MASKS: 145.8 -> 150.7), worker waves at s_setprio 1 (RF_TAIL_PRIO: -> 155.5), list slots by one LDS atomic per
// wave for power-of-two frames (RF_PARK_WAVE: -> 157.8), disc tails inside the wave for power-of-two frames
// (RF_DISC_WAVE, disc_tails_wave: 3 barriers per sample instead of 5, -> 159.9).
#pragma once

#include "rf_kernels.h"

namespace rf {

#ifndef RF_SETS
#define RF_SETS 3
#endif
#ifndef RF_SETS_OCC
#define RF_SETS_OCC 7 // waves per SIMD the register allocator is held to
#endif
constexpr int kSets = RF_SETS;
#ifndef RF_TWO_ROUNDS
#define RF_TWO_ROUNDS 1
#endif
#ifndef RF_TWO_ROUNDS_MIN
#define RF_TWO_ROUNDS_MIN 64 // entries above which the packing round is used (128: 140.0 instead of 142.9)
#endif
#ifndef RF_TWO_ROUNDS_MIN_DISC
// The disc tails never take the packing round: their ~165 stragglers per call finish on the three
// waves they fill (a rejected disc attempt is accepted next time with probability 0.785, so the
// per-wave tails are short) with two barriers instead of three.  64 / 128 / 256 (= never):
// 122.3 / 122.2 / 124.0 k env-steps/s.
#define RF_TWO_ROUNDS_MIN_DISC 256
#endif
#ifndef RF_COOP2_DISC_TRIPS
#define RF_COOP2_DISC_TRIPS 1 // in-wave disc attempts before the cooperative call (2: see DESIGN.md)
#endif
#ifndef RF_R1_SPHERE
#define RF_R1_SPHERE 2 // attempts per entry in the packing round (1 / 3: 142.9 / 144.0 with one in-wave attempt)
#endif
#ifndef RF_R1_DISC
#define RF_R1_DISC 1
#endif
#ifndef RF_COOP2_TRIPS
#define RF_COOP2_TRIPS 1
#endif
constexpr int kCoopTrips2 = RF_COOP2_TRIPS; // in-wave sphere attempts before the cooperative call
#ifndef RF_ADAPT_ON // hysteresis of the per-block switch between one and two in-wave sphere attempts
#define RF_ADAPT_ON 32
#endif
#ifndef RF_ADAPT_OFF
#define RF_ADAPT_OFF -32
#endif
#ifndef RF_NW
#define RF_NW 4 // waves per block of render_kernel_coop2 (2: measured; the host then takes the two-wave-wide layouts)
#endif
constexpr int kBlock2 = 64 * RF_NW; // threads per block of render_kernel_coop2
static_assert(RF_NW == 2 || RF_NW == 4, "tile layouts exist for two and four waves");
using CoopLds2 = CoopLdsT<kBlock2>;
#ifndef RF_COOP_CAP
#define RF_COOP_CAP kBlock2 // entries of the packed list; tests build a 32-entry one to stress the overflow path
#endif
constexpr int kCoopCap = RF_COOP_CAP;
static_assert(kCoopCap >= 1 && kCoopCap <= kBlock2, "the packed list lives in CoopLds2");
#ifndef RF_COLOUR_LDS
#define RF_COLOUR_LDS 2
#endif
#ifndef RF_MAYBE
#define RF_MAYBE 0 // 1: rejection loops leave on "not certainly rejected" (one compare), exact test after the conversion
#endif
#if RF_MAYBE
#define RF_DISC_TRY disc_attempt_maybe
#define RF_SPHERE_TRY sphere_attempt_maybe
#else
#define RF_DISC_TRY disc_attempt
#define RF_SPHERE_TRY sphere_attempt
#endif
// TIMING EXPERIMENTS ONLY (wrong frames; never set in a shipped build): cap the trips of the sparse tail loops --
// RF_TAILCAP_SPHERE: the second round of the sphere tails (one wave, ~6.5 trips), RF_TAILCAP_DISC: the disc
// workers (three waves, ~3.5 trips).  What the kernel gains with a cap of 0 / 1 is the most any organisation
// that removes / densifies those trips could gain.
#if defined(RF_TAILCAP_SPHERE) || defined(RF_TAILCAP_DISC)
#define RF_TAIL_LOOP(cap, attempt)                                                                 \
    for (int trip_ = 0; trip_ < (cap); ++trip_)                                                    \
        if (attempt)                                                                               \
            break;
#endif
#ifndef RF_PARK_ATOMIC
#define RF_PARK_ATOMIC 1
#endif
#ifndef RF_GEOM_OPAQUE
#define RF_GEOM_OPAQUE 1 // 0: let the compiler keep the per-thread geometry across the sample loop (it spills)
#endif
constexpr int kTileH2 = kTileH * kSets;

// A 32-bit value nobody has to compute: the raw-draw words of a sample are written by the first
// attempt of every lane that will ever read them (disc: every live lane; sphere: every lane that
// hit), so their initial value is irrelevant -- but it has to be *some* value for the compiler.
// An empty asm with an output gives it one without an instruction (zeroing 18 words per
// iteration was 2 % of the kernel's VALU instructions).
__device__ __forceinline__ uint32_t any_u32();
#ifndef RF_WORD_INIT
#define RF_WORD_INIT any_u32()
#endif
// Order of a draw's two raw words in an LDS entry: {low, high} is the order of the register pair the 64-bit sum
// was written to, so that a 16-byte read can land where the words are used (no copies).
#ifndef RF_WORDS_LOHI
#define RF_WORDS_LOHI 1
#endif
#if RF_WORDS_LOHI
#define RF_WORDS4(ww) make_uint4(ww[1], ww[0], ww[3], ww[2])
#define RF_WORDS2(ww) make_uint2(ww[5], ww[4])
#else
#define RF_WORDS4(ww) make_uint4(ww[0], ww[1], ww[2], ww[3])
#define RF_WORDS2(ww) make_uint2(ww[4], ww[5])
#endif
#ifndef RF_ANY_VOLATILE
#define RF_ANY_VOLATILE 1
#endif
__device__ __forceinline__ uint32_t any_u32()
{
    uint32_t v;
#if RF_ANY_VOLATILE
    asm volatile("" : "=v"(v)); // volatile: two calls are two values (merged, they cost a copy per use)
#else
    asm("" : "=v"(v));
#endif
    return v;
}

// the 16-byte entry at byte offset `offset` of an LDS array of uint4
__device__ __forceinline__ uint4 *entry16(uint4 *array, int offset)
{
    return reinterpret_cast<uint4 *>(reinterpret_cast<char *>(array) + offset);
}

// The workers' part of a cooperative call: `total` parked entries (state[parity][0 .. total)) are finished by the
// first lanes of the block; results in state / words4 / words2 at the entry's index.  Ends with a barrier.
#ifndef RF_TAIL_PRIO
#define RF_TAIL_PRIO 1 // s_setprio of the waves inside coop_workers: the tails are the block's critical path (0 / 1 / 2 / 3: 151.5 / 155.5 / 154.8 / 153.7 k env-steps/s)
#endif
#ifndef RF_TAIL_PRIO_R2_ONLY
#define RF_TAIL_PRIO_R2_ONLY 0
#endif
template <int DIM>
__device__ __forceinline__ void coop_workers(CoopLds2 &lds, int parity, int total, int tid)
{
    uint4 *const state = lds.state[parity];
#if RF_TAIL_PRIO && !RF_TAIL_PRIO_R2_ONLY
    __builtin_amdgcn_s_setprio(RF_TAIL_PRIO);
#endif
#if RF_TWO_ROUNDS
    if (total > (DIM == 2 ? RF_TWO_ROUNDS_MIN_DISC : RF_TWO_ROUNDS_MIN)) { // block-uniform
        // Round 1: the packed entries make a bounded number of attempts on as many waves as they
        // fill; the survivors are packed again -- into the other parity's state buffer, idle
        // during this call -- and finished in round 2 by (usually) a single wave, instead of every
        // worker wave dragging its own sparse tail.
        uint4 *const other = lds.state[parity ^ 1];
        if (tid < ((total + 63) & ~63)) { // whole waves
            bool pend = tid < total;
            Rng wg{0, 0, 0, 0};
            uint32_t ww[6] = {0, 0, 0, 0, 0, 0};
            if (pend) {
                const uint4 ps = state[tid];
                wg = Rng{ps.x, ps.y, ps.z, ps.w};
                for (int trip = 0; trip < (DIM == 2 ? RF_R1_DISC : RF_R1_SPHERE); ++trip) {
                    if (DIM == 2 ? RF_DISC_TRY(wg, ww) : RF_SPHERE_TRY(wg, ww)) {
                        pend = false;
                        break;
                    }
                }
                if (!pend) {
                    state[tid] = make_uint4(wg.a_lo, wg.a_hi, wg.b_lo, wg.b_hi);
                    lds.words4[tid] = RF_WORDS4(ww);
                    if (DIM == 3)
                        lds.words2[tid] = RF_WORDS2(ww);
                }
            }
#if RF_PARK_ATOMIC
            if (pend) { // as in the park step: one LDS atomic per surviving lane, in units of one entry's 16 bytes
                const int off2 = atomicAdd(&lds.cnt2, 16);
                *entry16(other, off2) = make_uint4(wg.a_lo, wg.a_hi, wg.b_lo, wg.b_hi);
                *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(lds.owner) + (off2 >> 3)) = (uint16_t)tid;
            }
#else
            const unsigned long long b2 = __ballot(pend);
            if (b2 != 0) {
                int base2 = 0;
                if ((tid & 63) == 0)
                    base2 = atomicAdd(&lds.cnt2, (int)__popcll(b2));
                base2 = __builtin_amdgcn_readfirstlane(base2);
                const int slot2 = base2 + __builtin_amdgcn_mbcnt_hi((unsigned)(b2 >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((unsigned)b2, 0));
                if (pend) {
                    other[slot2] = make_uint4(wg.a_lo, wg.a_hi, wg.b_lo, wg.b_hi);
                    lds.owner[slot2] = (uint16_t)tid;
                }
            }
#endif
        }
        __syncthreads();
#if RF_PARK_ATOMIC
        const int total2 = __builtin_amdgcn_readfirstlane(lds.cnt2) >> 4;
#else
        const int total2 = lds.cnt2;
#endif
#if RF_TAIL_PRIO && RF_TAIL_PRIO_R2_ONLY
        __builtin_amdgcn_s_setprio(RF_TAIL_PRIO);
#endif
        if (tid < total2) {
            const uint4 ps = other[tid];
            const int own = lds.owner[tid];
            Rng wg{ps.x, ps.y, ps.z, ps.w};
            uint32_t ww[6] = {0, 0, 0, 0, 0, 0};
            if (DIM == 2) {
                while (!RF_DISC_TRY(wg, ww)) {
                }
            } else {
#ifdef RF_TAILCAP_SPHERE
                RF_TAIL_LOOP(RF_TAILCAP_SPHERE, RF_SPHERE_TRY(wg, ww))
#else
                while (!RF_SPHERE_TRY(wg, ww)) {
                }
#endif
            }
            state[own] = make_uint4(wg.a_lo, wg.a_hi, wg.b_lo, wg.b_hi);
            lds.words4[own] = RF_WORDS4(ww);
            if (DIM == 3)
                lds.words2[own] = RF_WORDS2(ww);
        }
        __syncthreads();
        if (tid == 0)
            lds.cnt2 = 0;
    } else
#endif
    {
        if (tid < total) {
            const uint4 ps = state[tid];
            Rng wg{ps.x, ps.y, ps.z, ps.w};
            uint32_t ww[6] = {0, 0, 0, 0, 0, 0};
            if (DIM == 2) {
#ifdef RF_TAILCAP_DISC
                RF_TAIL_LOOP(RF_TAILCAP_DISC, RF_DISC_TRY(wg, ww))
#else
                while (!RF_DISC_TRY(wg, ww)) {
                }
#endif
            } else {
                while (!RF_SPHERE_TRY(wg, ww)) {
                }
            }
            state[tid] = make_uint4(wg.a_lo, wg.a_hi, wg.b_lo, wg.b_hi);
            lds.words4[tid] = RF_WORDS4(ww);
            if (DIM == 3)
                lds.words2[tid] = RF_WORDS2(ww);
        }
        __syncthreads();
    }
#if RF_TAIL_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if (tid == 0)
        lds.cnt[parity] = 0;
}

// coop_finish for kSets pixel sets at once.  The packed list holds kCoopCap entries (the LDS
// arrays of CoopLds); stragglers that do not fit finish their loop in their own wave.  Returns
// the number of stragglers the block had (block-uniform).
template <int DIM>
__device__ __forceinline__ int coop_finish2(CoopLds2 &lds, int parity, bool (&need)[kSets], Rng (&g)[kSets],
                                            uint32_t (&w)[kSets][6], int tid)
{
    asm volatile("" : "+v"(tid)); // keeps the LDS addresses derived from it out of long-lived registers
    uint4 *const state = lds.state[parity];
#if RF_PARK_ATOMIC
    // Every straggler takes its slot with its own LDS atomic (ds_add_rtn_u32 under the lanes' mask: the LDS
    // unit hands the lanes of one instruction consecutive values) instead of a ballot, two v_mbcnt and one
    // atomic per wave and set: the serialisation happens in the LDS pipe, which has room, the saved
    // instructions were slow-path VALU ones.  +0.4 % at the headline configuration, +0.9 % at 512 px, +2.8 %
    // at 300 px / 100 spp.  The Makefile passes -mllvm -amdgpu-atomic-optimizer-strategy=None: the compiler's
    // atomic optimizer would otherwise turn this back into exactly the ballot form.
    int slot[kSets];
    bool parked[kSets];
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        // (the counter counts in units of 16 bytes -- an entry of the state array -- so that what the atomic
        // returns is the entry's byte offset, without a shift)
        slot[j] = kCoopCap * 16;
        if (need[j])
            slot[j] = atomicAdd(&lds.cnt[parity], 16);
        parked[j] = need[j] && slot[j] < kCoopCap * 16;
#else
    unsigned long long ballot[kSets];
    int pop = 0;
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        ballot[j] = __ballot(need[j]);
        pop += (int)__popcll(ballot[j]);
    }
    int base = 0;
    if (pop != 0) { // wave-uniform
        if ((tid & 63) == 0)
            base = atomicAdd(&lds.cnt[parity], pop);
        base = __builtin_amdgcn_readfirstlane(base);
    }
    int slot[kSets];
    bool parked[kSets];
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        slot[j] = base + __builtin_amdgcn_mbcnt_hi((unsigned)(ballot[j] >> 32),
                                                   __builtin_amdgcn_mbcnt_lo((unsigned)ballot[j], 0));
        base += (int)__popcll(ballot[j]);
        parked[j] = need[j] && slot[j] < kCoopCap;
        slot[j] *= 16; // byte offset of the entry, as in the atomic form
#endif
        if (parked[j])
            *entry16(state, slot[j]) = make_uint4(g[j].a_lo, g[j].a_hi, g[j].b_lo, g[j].b_hi);
        if (need[j] && !parked[j]) { // overflow of the packed list: finish in place
            if (DIM == 2) {
                while (!RF_DISC_TRY(g[j], w[j])) {
                }
            } else {
                while (!RF_SPHERE_TRY(g[j], w[j])) {
                }
            }
        }
    }
    __syncthreads();
#if RF_PARK_ATOMIC
    const int stragglers = lds.cnt[parity] >> 4;
#else
    const int stragglers = lds.cnt[parity];
#endif
    const int total = min(stragglers, kCoopCap);
    if (total == 0) // block-uniform
        return 0;
    coop_workers<DIM>(lds, parity, total, tid);
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        if (parked[j]) {
            const uint4 ps = *entry16(state, slot[j]);
            g[j] = Rng{ps.x, ps.y, ps.z, ps.w};
            const uint4 w4 = *entry16(lds.words4, slot[j]);
#if RF_WORDS_LOHI
            w[j][1] = w4.x; w[j][0] = w4.y; w[j][3] = w4.z; w[j][2] = w4.w;
#else
            w[j][0] = w4.x; w[j][1] = w4.y; w[j][2] = w4.z; w[j][3] = w4.w;
#endif
            if (DIM == 3) {
                const uint2 w2 = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(lds.words2) + (slot[j] >> 1));
#if RF_WORDS_LOHI
                w[j][5] = w2.x; w[j][4] = w2.y;
#else
                w[j][4] = w2.x; w[j][5] = w2.y;
#endif
            }
        }
    }
    return stragglers;
}

// Lane predicates that live across the sample loop or a cooperative call are kept as 64-bit lane masks in scalar
// registers (RF_MASKS): a `bool` that crosses control flow ends up as a 0 / 1 byte in a vector register -- one
// v_cndmask to make it, a v_mov to clear it, a v_and + v_cmp to use it, all of them per set and phase, the compares on
// the slow VALU path.  __builtin_amdgcn_inverse_ballot_w64 turns a mask back into the lanes' predicate without an
// instruction (it is the s_and_saveexec operand).
#ifndef RF_MASKS
#define RF_MASKS 1
#endif
#ifndef RF_PARK_WAVE
#define RF_PARK_WAVE -1 // list slots per wave (1) / per straggler (0) / per kernel instance (-1): see coop_finish2m
#endif
#ifndef RF_COLOUR_ATOMIC
// 1: the colour sums in LDS by ds_add_f32 instead of read + add + write.  Bit-identical (tests/gpucheck
// gc_check_lds_add) and 2.5x slower end to end (61.8 k against 155.4 k env-steps/s): the LDS unit's float atomics
// are nowhere near one wave instruction per few cycles.
#define RF_COLOUR_ATOMIC 0
#endif
typedef unsigned long long lanemask;
// a block-uniform integer condition, compared where it is used (s_cmp + s_cbranch_scc): hoisted out of the sample loop
// as a boolean it becomes a lane mask that vector instructions test
__device__ __forceinline__ int scalar_now(int v)
{
    v = __builtin_amdgcn_readfirstlane(v);
    asm volatile("" : "+s"(v));
    return v;
}
__device__ __forceinline__ bool lane_in(lanemask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ lanemask lanes_where(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// coop_finish2 with the stragglers given as lane masks
// WAVE_SLOTS: the stragglers' list slots by ONE LDS atomic per wave and call -- the masks are the ballots, the ranks
// come from v_mbcnt (3 vector instructions per set) -- instead of one LDS atomic per straggler (no vector instruction;
// the LDS unit serialises the lanes of a same-address atomic, and it is 40 % busy in this kernel).  Measured per kernel
// instance (profiles/r03_ab.txt): power-of-two frames +1.0 ... +1.7 % (128 / 256 / 512 px), the others, whose float64
// pixel coordinates leave them more issue-bound, -1.3 ... -2.3 % (300 / 384 / 600 px): the kernel takes it for POW2.
template <int DIM, bool WAVE_SLOTS>
__device__ __forceinline__ int coop_finish2m(CoopLds2 &lds, int parity, const lanemask (&need)[kSets], Rng (&g)[kSets],
                                             uint32_t (&w)[kSets][6], int tid)
{
    asm volatile("" : "+v"(tid)); // keeps the LDS addresses derived from it out of long-lived registers
    uint4 *const state = lds.state[parity];
    int slot[kSets];
    lanemask parked[kSets];
    if (RF_PARK_WAVE < 0 ? WAVE_SLOTS : RF_PARK_WAVE != 0) {
        int pop[kSets], all = 0;
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            pop[j] = (int)__builtin_popcountll(need[j]);
            all += pop[j];
        }
        int base = 0;
        if (all != 0) { // wave-uniform
            if ((tid & 63) == 0)
                base = atomicAdd(&lds.cnt[parity], all * 16);
            base = __builtin_amdgcn_readfirstlane(base);
        }
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(need[j] >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((unsigned)need[j], 0));
            slot[j] = base + rank * 16;
            base += pop[j] * 16;
        }
    } else {
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            // one LDS atomic per straggler, counting in units of an entry's 16 bytes (see coop_finish2); all sets'
            // atomics are issued before the first result is waited for
            slot[j] = (int)any_u32();
            if (lane_in(need[j]))
                slot[j] = atomicAdd(&lds.cnt[parity], 16);
        }
    }
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        parked[j] = need[j] & lanes_where(slot[j] < kCoopCap * 16); // (a ballot of one compare is that compare)
        if (lane_in(parked[j]))
            *entry16(state, slot[j]) = make_uint4(g[j].a_lo, g[j].a_hi, g[j].b_lo, g[j].b_hi);
        if (lane_in(need[j] & ~parked[j])) { // overflow of the packed list: finish in place
            if (DIM == 2) {
                while (!RF_DISC_TRY(g[j], w[j])) {
                }
            } else {
                while (!RF_SPHERE_TRY(g[j], w[j])) {
                }
            }
        }
    }
    __syncthreads();
    const int stragglers = __builtin_amdgcn_readfirstlane(lds.cnt[parity]) >> 4; // (a scalar: the branches below are s_cmp)
    const int total = min(stragglers, kCoopCap);
    if (total == 0) // block-uniform
        return 0;
    coop_workers<DIM>(lds, parity, total, tid);
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        if (lane_in(parked[j])) {
            uint4 ps = *entry16(state, slot[j]);
            // (opaque: or the first draw's 64-bit sum is fed by a second, 8-byte read of the same entry)
            asm volatile("" : "+v"(ps.x), "+v"(ps.y), "+v"(ps.z), "+v"(ps.w));
            g[j] = Rng{ps.x, ps.y, ps.z, ps.w};
            const uint4 w4 = *entry16(lds.words4, slot[j]);
            static_assert(RF_WORDS_LOHI, "entry layout");
            w[j][1] = w4.x; w[j][0] = w4.y; w[j][3] = w4.z; w[j][2] = w4.w;
            if (DIM == 3) {
                const uint2 w2 = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(lds.words2) + (slot[j] >> 1));
                w[j][5] = w2.x; w[j][4] = w2.y;
            }
        }
    }
    return stragglers;
}

// Disc tails without the block: every wave packs the stragglers of its own three pixel sets (3 x 64 x 0.215 = 41 on
// average) onto its first lanes through its quarter of state[0], finishes them there and hands the results back --
// all of it inside the wave, in LDS order, with no barrier (the block-wide form costs two per sample and makes every
// wave wait for the slowest worker).  An entry's 16 bytes carry the state to the worker, the advanced state back, and
// then the accepted draws' four words back (two round trips through the same slot: the sphere phase of the previous
// sample may still be read from every other array by slower waves).  Stragglers beyond the 64 slots finish in place.
// Measured per kernel instance (profiles/r03_ab.txt): +1.5 % at 256 px, +2.5 % at 512 px; the instances for other
// frame sizes (float64 pixel coordinates: more registers, more issue-bound) spill with it and lose 1.2 %, so they keep
// the block-wide call.  RF_DISC_WAVE: 1 / 0 = everywhere / nowhere, -1 = power-of-two frames only.
#ifndef RF_DISC_WAVE
#define RF_DISC_WAVE -1
#endif
#ifndef RF_DISC_WAVE_SLOTS
#define RF_DISC_WAVE_SLOTS 64 // entries per wave; tests build an 8-entry form to exercise the in-place path
#endif
constexpr int kDiscWaveSlots = RF_DISC_WAVE_SLOTS;
static_assert(kDiscWaveSlots >= 1 && kDiscWaveSlots <= 64, "a wave's quarter of state[0]");
__device__ __forceinline__ void wave_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void disc_tails_wave(uint4 *region, const lanemask (&need)[kSets], Rng (&g)[kSets],
                                                uint32_t (&w)[kSets][6], int tid)
{
    // region = the block's state[0]; this wave's entries are [wbase, wbase + 64) (a scalar: folded into the slots)
    const int wbase = __builtin_amdgcn_readfirstlane(tid) & ~63;
    int total = 0, first[kSets];
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        first[j] = total;
        total += (int)__builtin_popcountll(need[j]);
    }
    if (total == 0) // wave-uniform
        return;
    int slot[kSets];
    lanemask packed[kSets];
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(need[j] >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((unsigned)need[j], 0));
        slot[j] = (wbase + first[j] + rank) * 16; // byte offset of the entry
        packed[j] = need[j] & lanes_where(slot[j] < (wbase + kDiscWaveSlots) * 16);
        if (lane_in(packed[j]))
            *entry16(region, slot[j]) = make_uint4(g[j].a_lo, g[j].a_hi, g[j].b_lo, g[j].b_hi);
        if (lane_in(need[j] & ~packed[j])) { // more stragglers in one wave than slots (p ~ 1e-5 with 64): in place
            while (!disc_attempt(g[j], w[j])) {
            }
        }
    }
    wave_lds_order();
    const bool worker = tid < wbase + min(total, kDiscWaveSlots);
    Rng wg{0, 0, 0, 0};
    if (worker) {
        const uint4 ps = region[tid];
        wg = Rng{ps.x, ps.y, ps.z, ps.w};
        uint32_t ww[6] = {0, 0, 0, 0, 0, 0};
        while (!disc_attempt(wg, ww)) {
        }
        region[tid] = RF_WORDS4(ww); // the accepted draws first: the worker keeps the four state words meanwhile
    }
    wave_lds_order();
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        if (lane_in(packed[j])) {
            const uint4 w4 = *entry16(region, slot[j]);
            static_assert(RF_WORDS_LOHI, "entry layout");
            w[j][1] = w4.x; w[j][0] = w4.y; w[j][3] = w4.z; w[j][2] = w4.w;
        }
    }
    wave_lds_order();
    if (worker)
        region[tid] = make_uint4(wg.a_lo, wg.a_hi, wg.b_lo, wg.b_hi);
    wave_lds_order();
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        if (lane_in(packed[j])) {
            uint4 ps = *entry16(region, slot[j]);
            asm volatile("" : "+v"(ps.x), "+v"(ps.y), "+v"(ps.z), "+v"(ps.w));
            g[j] = Rng{ps.x, ps.y, ps.z, ps.w};
        }
    }
    wave_lds_order(); // (the next sample's stragglers overwrite the slots)
}

template <bool POW2, int LENS, int WX = kWavesX, int WW = kWaveW>
__global__ __launch_bounds__(kBlock2, RF_SETS_OCC) void render_kernel_coop2(RenderArgs a)
{
    // tile of a block: WX waves (of WW x 64 / WW pixels) side by side, 4 / WX down, kSets sets
    constexpr int tWaveW = WW, tWaveH = 64 / WW;
    static_assert(WX <= RF_NW, "waves side by side");
    constexpr int tWavesX = WX, tTileW = WX * tWaveW, tTileH = (RF_NW / WX) * tWaveH, tTileH2 = tTileH * kSets;
    __shared__ CoopLds2 lds;
    // the frame staging buffer (kSets * 768 B) reuses the words4 array once the sample loop is over
    static_assert(sizeof(lds.words4) >= (size_t)kSets * kBlock2 * 3, "stage does not fit");
    uint32_t *const stage = reinterpret_cast<uint32_t *>(lds.words4);
#if RF_COLOUR_LDS > 0
    // colour accumulators of the first RF_COLOUR_LDS pixel sets live in LDS (one read-modify-write
    // per sample and channel, off the vector ALU) to keep the kernel inside its VGPR budget
    __shared__ float lds_colour[RF_COLOUR_LDS][3][kBlock2];
#endif

    const int e = blockIdx.y;
    const int tid = threadIdx.x;
#ifdef RF_LDS_PAD // TIMING EXPERIMENTS ONLY: extra LDS per block, to take resident blocks away from a CU
    __shared__ uint32_t lds_pad[RF_LDS_PAD / 4];
    if (a.spp < 0)
        lds_pad[tid] = (uint32_t)e;
    asm volatile("" ::"v"(lds_pad[tid & 3]));
#endif
    if (skip_env(a.rect, e)) // block-uniform, before any barrier
        return;
    if (tid < 2)
        lds.cnt[tid] = 0;
    if (tid == 2)
        lds.cnt2 = 0;
    __syncthreads();
    const int tiles_x = (a.w + tTileW - 1) / tTileW;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
    const bool mirror = (2 * tile_x + 1) * tTileW > a.w; // see render_kernel_coop

    // Pixel geometry of a thread.  Set j covers the rows tTileH * j further down.  All of it is
    // cheap to derive from the thread index, and the sample loop derives it afresh every
    // iteration from an index the compiler cannot see through (Geometry::opaque): kept alive
    // across the loop these loop invariants -- x, y, their float forms as packed-math pairs,
    // liveness masks, LDS addresses -- are what the register allocator spills.
    struct Geometry {
        int col, row0, x, y0;
        bool live_x;
        int h;
        static __device__ __forceinline__ int opaque(int v)
        {
            asm volatile("" : "+v"(v));
            return v;
        }
        __device__ __forceinline__ int y_of(int j) const { return y0 + j * tTileH; }
        __device__ __forceinline__ bool live_of(int j) const { return live_x && y_of(j) < h; }
    };
    static_assert((tWavesX & (tWavesX - 1)) == 0 && (tWaveW & (tWaveW - 1)) == 0, "masks and shifts below");
    const unsigned mirror_mask = mirror ? (unsigned)(tWavesX - 1) : 0u;
    auto geometry = [&](int t) {
        // (unsigned masks and shifts: the signed / and % of the same powers of two cost sign fix-ups every iteration)
        __builtin_assume(t >= 0 && t < kBlock2); // (the per-iteration index is opaque: without this, bits 8.. are computed with)
        const unsigned ut = (unsigned)t, wv = ut >> 6, lane = ut & 63u;
        const unsigned wq = wv & (unsigned)(tWavesX - 1);
        // mirrored blocks count their waves from the right: (tWavesX - 1) - wq == wq ^ (tWavesX - 1), a block-uniform mask
        const int wx = (int)(wq ^ mirror_mask);
        Geometry r;
        r.col = wx * tWaveW + (int)(lane & (unsigned)(tWaveW - 1));
        r.row0 = (int)(wv / (unsigned)tWavesX) * tWaveH + (int)(lane / (unsigned)tWaveW);
        r.x = tile_x * tTileW + r.col;
        r.y0 = tile_y * tTileH2 + r.row0;
        r.live_x = r.x < a.w;
        r.h = a.h;
        return r;
    };
    auto pix_of = [&](const Geometry &q, int j) {
        return (size_t)e * a.hw + (q.live_of(j) ? (size_t)q.y_of(j) * a.w + q.x : 0);
    };
    Rng g[kSets];
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        const Geometry g0 = geometry(tid);
        g[j] = rng_load(0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull); // dead lanes: any state
        if (g0.live_of(j)) {
            const ulonglong2 st = a.states[pix_of(g0, j)];
            g[j] = rng_load(st.x, st.y);
        }
    }
    PixelEnv env0 = make_pixel_env(a.cam_dyn + (size_t)e * 9, a.rect + (size_t)e * 2);
    // block-uniform values computed with vector instructions: keep them in scalar registers
    auto uniform = [](float v) { // (the builtin alone is folded away for values known to be uniform)
        int bits = __builtin_bit_cast(int, v);
        asm volatile("" : "+v"(bits));
        return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(bits));
    };
    env0.tt = uniform(env0.tt);
    env0.den = uniform(env0.den);
    env0.rden = uniform(env0.rden);

    float cr[kSets], cg[kSets], cb[kSets];
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        cr[j] = cg[j] = cb[j] = 0.0f;
#if RF_COLOUR_LDS > 0
        if (j < RF_COLOUR_LDS)
            lds_colour[j][0][tid] = lds_colour[j][1][tid] = lds_colour[j][2][tid] = 0.0f;
#endif
    }

    int sphere_trips = kCoopTrips2; // in-wave sphere attempts of the current sample (block-uniform)
#if RF_MASKS
    static_assert(RF_COOP2_DISC_TRIPS == 1 && !RF_MAYBE, "the mask form of the sample loop");
    lanemask live_m[kSets]; // lanes whose pixel of set j is inside the frame
#pragma unroll
    for (int j = 0; j < kSets; ++j)
        live_m[j] = lanes_where(geometry(tid).x < a.w) & lanes_where(geometry(tid).y_of(j) < a.h);
    // per-environment conditions as scalars (a uniform `bool` is a lane mask that vector instructions test)
    const int tmiss_s = __builtin_amdgcn_readfirstlane((int)env0.tmiss);
    for (int k = 0; k < a.spp; ++k) {
        const PixelEnv &env = env0;
        uint32_t w[kSets][6];
        float s[kSets], t[kSets];
        lanemask need_m[kSets];
        const Geometry gk = geometry(RF_GEOM_OPAQUE ? Geometry::opaque(tid) : tid);
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            // ((float)y of the further sets by an exact float addition: conversions issue on the slow path)
            sample_coords<POW2>(g[j], gk.x, gk.y_of(j), (float)gk.x, (float)gk.y0 + (float)(j * tTileH), a.h64, a.w64,
                                a.inv_w, a.inv_h, a.rw64, a.rh64, s[j], t[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w[j][i] = RF_WORD_INIT;
            // (every lane makes the attempt -- a wave's instructions cost the same with any lanes off, and a lane
            // outside the frame never stores its state)
            const float sq = disc_attempt_sq(g[j], w[j]);
            need_m[j] = live_m[j] & ~lanes_where(sq < 1.0f);
        }
        if (RF_DISC_WAVE < 0 ? POW2 : RF_DISC_WAVE != 0)
            disc_tails_wave(lds.state[0], need_m, g, w, tid);
        else
            coop_finish2m<2, POW2>(lds, 0, need_m, g, w, tid);

        float rdx[kSets], rdy[kSets], rdz[kSets];
        lanemask hit_m[kSets];
        bool red[kSets];
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            float p0, p1;
            disc_finish(w[j], p0, p1);
            const AxisRay ray = sample_axis_point<LENS>(p0, p1, env, a.cs, s[j], t[j]);
            rdx[j] = ray.dx;
            rdy[j] = ray.dy;
            rdz[j] = ray.dz;
            // (the ballot of ONE compare is that compare; the lanes' predicate comes back from the mask)
            hit_m[j] = live_m[j] & lanes_where(!(ray.reach > env.rect.half)); // rectangle.py:135
            if (tmiss_s)                                          // rectangle.py:130
                hit_m[j] = 0;
            red[j] = false;
            if (lane_in(hit_m[j]))
                red[j] = sample_axis_red(ray.px, ray.py, env, a.tab);
            w[j][4] = RF_WORD_INIT;
            w[j][5] = RF_WORD_INIT;
            need_m[j] = hit_m[j];
#pragma unroll
            for (int trip = 0; trip < kCoopTrips2 + 1; ++trip) {
                if (trip >= kCoopTrips2 && scalar_now(sphere_trips) <= kCoopTrips2) // block-uniform, a scalar
                    break;
                if (need_m[j] != 0) { // wave-uniform
                    float sq = 2.0f;
                    if (lane_in(need_m[j]))
                        sq = sphere_attempt_sq(g[j], w[j]);
                    asm volatile("" : "+v"(sq)); // (or the compare moves into the branch and its ballot costs two more)
                    need_m[j] &= ~lanes_where(sq < 1.0f);
                }
            }
        }
        // (the per-block switch between one and two in-wave attempts: see the bool form below)
        const int stragglers = __builtin_amdgcn_readfirstlane(coop_finish2m<3, POW2>(lds, 1, need_m, g, w, tid));
        if (sphere_trips == kCoopTrips2 && stragglers > kCoopCap + RF_ADAPT_ON)
            sphere_trips = kCoopTrips2 + 1;
        else if (sphere_trips != kCoopTrips2 && 2 * stragglers < kCoopCap + RF_ADAPT_OFF)
            sphere_trips = kCoopTrips2;

#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
            const bool hit = lane_in(hit_m[j]);
            if (hit)
                sphere_finish(w[j], q0, q1, q2);
            const Colour c = sample_axis_shade(hit, red[j], rdx[j], rdy[j], rdz[j], q0, q1, q2);
#if RF_COLOUR_LDS > 0
            if (j < RF_COLOUR_LDS) {
#if RF_COLOUR_ATOMIC
                // ds_add_f32: the LDS unit's float32 addition, nothing returned, nothing to wait for (measured: see above)
                atomicAdd(&lds_colour[j][0][tid], c.r);
                atomicAdd(&lds_colour[j][1][tid], c.g);
                atomicAdd(&lds_colour[j][2][tid], c.b);
#else
                lds_colour[j][0][tid] = add2_not_negzero(lds_colour[j][0][tid], c.r);
                lds_colour[j][1][tid] = add2_not_negzero(lds_colour[j][1][tid], c.g);
                lds_colour[j][2][tid] = add2_not_negzero(lds_colour[j][2][tid], c.b);
#endif
                continue;
            }
#endif
            cr[j] = add2_not_negzero(cr[j], c.r);
            cg[j] = add2_not_negzero(cg[j], c.g);
            cb[j] = add2_not_negzero(cb[j], c.b);
        }
    }
#else
    for (int k = 0; k < a.spp; ++k) {
        const Geometry gk = geometry(RF_GEOM_OPAQUE ? Geometry::opaque(tid) : tid);
        const PixelEnv &env = env0;
        uint32_t w[kSets][6];
        float s[kSets], t[kSets];
        bool need[kSets];
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            sample_coords<POW2>(g[j], gk.x, gk.y_of(j), (float)gk.x, (float)gk.y_of(j), a.h64, a.w64, a.inv_w, a.inv_h,
                                a.rw64, a.rh64, s[j], t[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w[j][i] = RF_WORD_INIT;
            need[j] = gk.live_of(j);
#pragma unroll
            for (int trip = 0; trip < RF_COOP2_DISC_TRIPS; ++trip) {
                if (trip == 0 || __any(need[j])) { // wave-uniform
                    if (need[j] && RF_DISC_TRY(g[j], w[j]))
                        need[j] = false;
                }
            }
        }
        coop_finish2<2>(lds, 0, need, g, w, tid);

        AxisPre pre[kSets];
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            float p0, p1;
            disc_finish(w[j], p0, p1);
#if RF_MAYBE
            {
                const bool redo = gk.live_of(j) && !disc_exact_ok(p0, p1); // inside the band, on the wrong side
                if (__builtin_expect(__any(redo), 0)) {
                    if (redo) {
                        while (!disc_attempt(g[j], w[j])) {
                        }
                        disc_finish(w[j], p0, p1);
                    }
                }
            }
#endif
            pre[j] = sample_axis_ray<LENS>(p0, p1, env, a.cs, s[j], t[j], a.tab);
            w[j][4] = RF_WORD_INIT;
            w[j][5] = RF_WORD_INIT;
            need[j] = gk.live_of(j) && pre[j].hit;
#pragma unroll
            for (int trip = 0; trip < kCoopTrips2 + 1; ++trip) {
                if ((trip < kCoopTrips2 || sphere_trips > kCoopTrips2) && __any(need[j])) { // wave-uniform
                    if (need[j] && RF_SPHERE_TRY(g[j], w[j]))
                        need[j] = false;
                }
            }
        }
        // One in-wave attempt leaves ~48 % of a hit wave's lanes for the packed list, which is where
        // a second attempt is best made (full waves) -- unless the list then overflows: a block
        // whose tile lies inside the target has ~366 such lanes for 256 entries, and the rest would
        // finish in place.  So a block switches to two in-wave attempts when its list overflowed
        // on the previous sample, and back when it would fit again with one.
        const int stragglers = coop_finish2<3>(lds, 1, need, g, w, tid);
        if (sphere_trips == kCoopTrips2 && stragglers > kCoopCap + RF_ADAPT_ON)
            sphere_trips = kCoopTrips2 + 1;
        else if (sphere_trips != kCoopTrips2 && 2 * stragglers < kCoopCap + RF_ADAPT_OFF)
            sphere_trips = kCoopTrips2;

#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
            if (pre[j].hit)
                sphere_finish(w[j], q0, q1, q2);
#if RF_MAYBE
            {
                const bool redo = gk.live_of(j) && pre[j].hit && !sphere_exact_ok(q0, q1, q2);
                if (__builtin_expect(__any(redo), 0)) {
                    if (redo) {
                        while (!sphere_attempt(g[j], w[j])) {
                        }
                        sphere_finish(w[j], q0, q1, q2);
                    }
                }
            }
#endif
            const Colour c = sample_axis_shade(pre[j], q0, q1, q2);
#if RF_COLOUR_LDS > 0
            if (j < RF_COLOUR_LDS) {
                // (the sums and every sample colour are >= +0, never -0: see add2_not_negzero)
                lds_colour[j][0][tid] = add2_not_negzero(lds_colour[j][0][tid], c.r);
                lds_colour[j][1][tid] = add2_not_negzero(lds_colour[j][1][tid], c.g);
                lds_colour[j][2][tid] = add2_not_negzero(lds_colour[j][2][tid], c.b);
                continue;
            }
#endif
            cr[j] = add2_not_negzero(cr[j], c.r);
            cg[j] = add2_not_negzero(cg[j], c.g);
            cb[j] = add2_not_negzero(cb[j], c.b);
        }
    }
#endif // RF_MASKS
#if RF_COLOUR_LDS > 0
#pragma unroll
    for (int j = 0; j < RF_COLOUR_LDS && j < kSets; ++j) {
        cr[j] = lds_colour[j][0][tid];
        cg[j] = lds_colour[j][1][tid];
        cb[j] = lds_colour[j][2][tid];
    }
#endif
    __syncthreads(); // the cooperative arrays are dead from here on: words4 becomes the stage

    uint8_t *sb = reinterpret_cast<uint8_t *>(stage);
    const Geometry ge = geometry(Geometry::opaque(tid));
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        if (ge.live_of(j))
            a.states[pix_of(ge, j)] = make_ulonglong2(rng_s0(g[j]), rng_s1(g[j]));
        const uint8_t r8 = (uint8_t)(cr[j] * a.scale);
        const uint8_t g8 = (uint8_t)(cg[j] * a.scale);
        const uint8_t b8 = (uint8_t)(cb[j] * a.scale);
        if ((a.w & 3) == 0) {
            const int slot = (j * tTileH + ge.row0) * tTileW + ge.col;
            sb[slot * 3 + 0] = r8;
            sb[slot * 3 + 1] = g8;
            sb[slot * 3 + 2] = b8;
        } else if (ge.live_of(j)) {
            uint8_t *dst = a.frames + pix_of(ge, j) * 3;
            dst[0] = r8;
            dst[1] = g8;
            dst[2] = b8;
        }
    }
    if ((a.w & 3) == 0) {
        // the tile's rows (tTileW * 3 B each) -> LDS -> coalesced dword stores per row
        __syncthreads();
        constexpr int kRowDw = tTileW * 3 / 4;
        for (int i = tid; i < tTileH2 * kRowDw; i += kBlock2) {
            const int r = i / kRowDw, d = i - r * kRowDw;
            const int yy = tile_y * tTileH2 + r;
            const int valid_dw = min(tTileW, a.w - tile_x * tTileW) * 3 / 4; // w % 4 == 0
            if (yy < a.h && d < valid_dw) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(
                    a.frames + (((size_t)e * a.h + yy) * a.w + (size_t)tile_x * tTileW) * 3);
                dst[d] = stage[r * kRowDw + d];
            }
        }
    }
}

} // namespace rf
