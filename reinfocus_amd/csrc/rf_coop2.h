// rf_coop2.h -- render_kernel_coop2<POW2, LENS, WX, WW>: the default render kernel.
//
// Same arithmetic and the same pixel <-> RNG-state mapping as render_kernel (rf_render.h), with the rejection
// loops made block-cooperative.  In a wave whose 64 pixels all hit the target the sphere loop of physics.py:31-44
// runs for max-over-lanes = ~6.5 trips although a lane needs 1.9 on average (disc: 3.5 against 1.27).  Here every
// lane makes its first attempts in its own wave; the lanes still looking hand their RNG state to a packed list in
// LDS that full waves finish, and take the advanced state and the accepted draws back.  Which lane executes an
// attempt does not matter -- the stream of a pixel is advanced by exactly the same draws -- so results are
// bit-identical to render_kernel (and to the oracle).  All 256 threads of a block run the sample loop in lockstep
// (dead threads of a partial block included) so that every barrier is reached by every thread.
// Organised for gfx950 (every choice below was measured; the rejected alternatives are in git history,
// profiles/HISTORY.md and profiles/r0*_ab.txt, not in this file):
//
//  * kSets = 3 pixels per thread (tile 128 x 6: a thread owns (x, y + 2 j), j < 3): every wave does
//    the in-wave work of three pixel sets between two barriers and one cooperative call serves all
//    three.  Spilling is not an option (every spilled dword of a wave is 256 B of scratch traffic),
//    so the kernel (a) re-derives the per-thread geometry inside the sample loop from an index the
//    compiler cannot see through, (b) moves block-uniform values into scalar registers, (c) keeps
//    the colour accumulators of two sets in LDS and (d) stages the frame through a cooperative
//    array: 72 VGPRs (7 waves per SIMD), 20.5 KB LDS.
//  * lane predicates that live across the sample loop or a cooperative call are 64-bit lane masks
//    in scalar registers (a `bool` that crosses control flow is a 0 / 1 byte in a vector register:
//    a v_cndmask to make it, a v_mov to clear it, a v_and + v_cmp to use it, the compares on the
//    slow VALU path); __builtin_amdgcn_inverse_ballot_w64 turns a mask back into the lanes'
//    predicate without an instruction (it is the s_and_saveexec operand).
//  * sphere tails: one in-wave attempt, then the block-wide list.  When the list needs more than
//    one wave its entries first make two attempts on as many waves as they fill and the survivors
//    are packed again, so that a single wave runs the sparse end of the tail.  A block whose list
//    overflows (a tile inside the target has ~366 stragglers for 256 entries) goes back to two
//    in-wave attempts until it would fit again.  The worker waves run at s_setprio 1.
//  * disc tails: power-of-two frames finish them inside the wave (disc_tails_wave: no block
//    barrier in the disc phase, 3 barriers per sample instead of 5); the other instances, whose
//    float64 pixel coordinates leave no registers for that, use the block-wide list.
//  * list slots: one LDS atomic per wave (ranks from v_mbcnt on the masks) for power-of-two
//    frames, one per straggler for the others (measured per instance, profiles/r03_ab.txt).
//
// SYNCHRONISATION (which barrier orders what; `k` is the sample index):
//   sphere call of sample k = park P(k) | B1 | [round 1 | B2 | round 2 | B3]  or  [finish | B3] | collect C(k) [| B4]
//   * P(k): atomics on cnt[c], writes state[1][slot].  B1.  Everyone reads cnt[c]: 0 -> return.
//   * rounds: workers read / write state[1], words4, words2, `other` = state[0], owner, cnt2; thread 0
//     clears cnt2 and cnt[c] after B3.
//   * C(k): owners read state[1][slot], words4[slot], words2[slot].
//   Instances with the block-wide disc call (not POW2): the disc call of sample k + 1 (parity 0, its own
//   B1 .. B3) lies between C(k) / the reset of cnt[1] and P(k + 1), and the sphere call's B1 between the
//   disc call's collect / reset and the next disc park: every reuse of a counter or an array is separated
//   from its last reader by at least one barrier that all four waves execute (c = parity).
//   Instances with in-wave disc tails (POW2): the disc phase has no barrier, so without more a fast
//   wave's P(k + 1) -- atomics on the counter, writes into state[1] -- would be unordered against a slow
//   wave's C(k) reads of state[1], against thread 0's reset, and (in a block with no stragglers, which
//   leaves after B1) against a slow wave's read of the counter.  Two measures close that:
//     - the counter alternates with the sample (c = k & 1): cnt[c] is read after B1(k) and next touched
//       by P(k + 2), after B1(k + 1), which no wave passes before it has read cnt[c] for sample k;
//     - B4, a barrier after C(k) and after the resets: P(k + 1) and disc_tails_wave(k + 1) follow it.
//       All four waves leave B3 together and C(k) is nine LDS reads, so they reach B4 together.
//   disc_tails_wave works on the wave's own quarter of state[0], which after B3 nobody else reads (`other`
//   is dead once round 2 is over) and which round 1 of the next call writes only after B1.
//   Checked two ways: tests/test_sync_model.py writes this protocol down array by array and verifies, for every
//   combination of call outcomes, that no two conflicting accesses of different waves share a barrier epoch (and that
//   the round-3 form fails that check); tests/test_gpu_parity.py::test_delayed_waves_change_nothing runs a build
//   whose waves are delayed at exactly these points (RF_TEST_SKEW below).
#pragma once

#include <type_traits>

#include "rf_render.h"

namespace rf {

// The LDS arrays of a cooperative call.  `state` is doubled because a fast wave parks the stragglers of the next
// call while a slow one is still collecting; the draws are only written after the next call's first barrier.
template <int N>
struct CoopLdsT {
    uint4 state[2][N];
    uint4 words4[N];
    uint2 words2[N];
    uint16_t owner[N]; // original slot of a re-packed entry
    int cnt[2];
    int cnt2;          // entries in the second round
};

// RF_TEST_SKEW (tests/gpucheck/libreinfocus_skew.so, tests/test_gpu_parity.py; used by the cooperative calls
// below and by rf_general_one.h): one wave of every block -- a different
// one from call to call -- sleeps ~8 000 cycles at each point where a cooperative call is ordered against the next
// one by a barrier alone: before it reads the counter after B1, before thread 0's resets, before the collect reads.
// With the ordering right the sleeps change nothing (frames and RNG states stay bit-identical to the oracle); the
// round-3 form of the call -- no B4, one counter -- produces wrong frames under them (profiles/r04_ab.txt section 7).
#ifndef RF_TEST_SKEW
#define RF_TEST_SKEW 0
#endif
__device__ __forceinline__ void test_skew(int tid, int turn)
{
#if RF_TEST_SKEW
    if (((tid >> 6) & 3) == (turn & 3)) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_s_sleep(127); // 8 x 127 x 64 cycles
    }
#else
    (void)tid;
    (void)turn;
#endif
}

// Pixel <-> lane mapping of the cooperative kernel: a wave owns kWaveW x kWaveH pixels and a
// block kWavesX x (4 / kWavesX) waves.  The mapping only changes which thread owns a pixel,
// not the pixel's RNG stream, so it is a pure scheduling knob (a wave always reads / writes
// whole 128-B lines of RNG state as long as kWaveW >= 8).  Measured at the headline config
// (tools sweep, round 1): 32 x 2 pixel waves side by side (128 x 2 blocks) are 8 % faster
// than 64 x 1 rows and than squarer tiles.
constexpr int kWaveW = 32, kWaveH = 64 / kWaveW;
constexpr int kWavesX = 4, kWavesY = (kBlock / 64) / kWavesX;
constexpr int kTileW = kWavesX * kWaveW, kTileH = kWavesY * kWaveH;


constexpr int kSets = 3;      // pixels per thread (2 / 3 / 4: 133 / 138 / 114 G samples/s)
constexpr int kSetsOcc = 7;   // waves per SIMD the register allocator is held to
constexpr int kBlock2 = 256;  // threads per block (four waves; two-wave blocks: -7 ... -17 %)
using CoopLds2 = CoopLdsT<kBlock2>;
#ifndef RF_COOP_CAP
#define RF_COOP_CAP kBlock2 // entries of the packed list; tests/gpucheck builds a 32-entry one to stress the overflow path
#endif
constexpr int kCoopCap = RF_COOP_CAP;
static_assert(kCoopCap >= 1 && kCoopCap <= kBlock2, "the packed list lives in CoopLds2");
#ifndef RF_DISC_WAVE_SLOTS
#define RF_DISC_WAVE_SLOTS 64 // entries per wave of disc_tails_wave; tests/gpucheck builds an 8-entry form (in-place path)
#endif
constexpr int kDiscWaveSlots = RF_DISC_WAVE_SLOTS;
static_assert(kDiscWaveSlots >= 1 && kDiscWaveSlots <= 64, "a wave's quarter of state[0]");

constexpr int kCoopTrips2 = 1;      // in-wave sphere attempts before the cooperative call
#ifndef RF_TWO_ROUNDS_MIN
#define RF_TWO_ROUNDS_MIN 64
#endif
#ifndef RF_ROUND1_SPHERE
#define RF_ROUND1_SPHERE 2
#endif
constexpr int kTwoRoundsMin = RF_TWO_ROUNDS_MIN;   // sphere entries above which the packing round is used
constexpr int kRound1Sphere = RF_ROUND1_SPHERE;    // attempts per entry in the packing round
constexpr int kAdaptOn = 32, kAdaptOff = -32; // hysteresis of the per-block switch between one and two in-wave attempts
constexpr int kColourLds = 2;       // pixel sets whose colour sums live in LDS
#ifndef RF_TAIL_PRIO
#define RF_TAIL_PRIO 1 // (tests/test_gpu_perf_guard.py was tried on a build with 0: profiles/r05_ab.txt section 5)
#endif
constexpr int kTailPrio = RF_TAIL_PRIO;        // s_setprio of the waves inside coop_workers (0 / 1 / 2 / 3: 151.5 / 155.5 / 154.8 / 153.7 k)
constexpr int kTileH2 = kTileH * kSets;

// A 32-bit value nobody has to compute: the raw-draw words of a sample are written by the first
// attempt of every lane that will ever read them (disc: every live lane; sphere: every lane that
// hit), so their initial value is irrelevant -- but it has to be *some* value for the compiler.
// An empty asm with an output gives it one without an instruction (zeroing 18 words per
// iteration was 2 % of the kernel's VALU instructions).
__device__ __forceinline__ uint32_t any_u32()
{
    uint32_t v;
    asm volatile("" : "=v"(v)); // volatile: two calls are two values (merged, they cost a copy per use)
    return v;
}
// Order of a draw's two raw words in an LDS entry: {low, high} is the order of the register pair the 64-bit sum
// was written to, so that a 16-byte read lands where the words are used (no copies).
#define RF_WORDS4(ww) make_uint4(ww[1], ww[0], ww[3], ww[2])
#define RF_WORDS2(ww) make_uint2(ww[5], ww[4])

// the 16-byte entry at byte offset `offset` of an LDS array of uint4
__device__ __forceinline__ uint4 *entry16(uint4 *array, int offset)
{
    return reinterpret_cast<uint4 *>(reinterpret_cast<char *>(array) + offset);
}

// the workers' loops: RF_COOP_UNROLL attempts per trip (a worker wave is what the block's other waves wait for at the next
// barrier; a taken branch costs it more than the instructions around it: two per trip +0.5 ... 0.9 % env-steps/s at the
// headline, three and four no more -- profiles/r06_ab.txt section 11)
#ifndef RF_COOP_UNROLL
#define RF_COOP_UNROLL 2
#endif
__device__ __forceinline__ void sphere_until_accepted(Rng &wg, uint32_t (&ww)[6])
{
    for (;;) {
        if (sphere_attempt(wg, ww))
            break;
#if RF_COOP_UNROLL >= 2
        if (sphere_attempt(wg, ww))
            break;
#endif
#if RF_COOP_UNROLL >= 3
        if (sphere_attempt(wg, ww))
            break;
#endif
#if RF_COOP_UNROLL >= 4
        if (sphere_attempt(wg, ww))
            break;
#endif
    }
}
__device__ __forceinline__ void disc_until_accepted(Rng &wg, uint32_t (&ww)[6])
{
    for (;;) {
        if (disc_attempt(wg, ww))
            break;
#if RF_COOP_UNROLL >= 2
        if (disc_attempt(wg, ww))
            break;
#endif
#if RF_COOP_UNROLL >= 3
        if (disc_attempt(wg, ww))
            break;
#endif
    }
}

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

// The workers' part of a cooperative call: `total` parked entries (state[parity][0 .. total)) are finished by the
// first lanes of the block; results in state / words4 / words2 at the entry's index.  Ends with a barrier (B3), after
// which thread 0 clears the counters (`cnt` is the one this call parked with).
template <int DIM>
__device__ __forceinline__ void coop_workers(CoopLds2 &lds, int parity, int *cnt, int total, int tid)
{
    uint4 *const state = lds.state[parity];
    __builtin_amdgcn_s_setprio(kTailPrio); // the tails are serial work three other waves of the block wait for
    if (DIM == 3 && total > kTwoRoundsMin) { // block-uniform
        // Round 1: the packed entries make a bounded number of attempts on as many waves as they
        // fill; the survivors are packed again -- into the other parity's state buffer, idle
        // during this call -- and finished in round 2 by (usually) a single wave, instead of every
        // worker wave dragging its own sparse tail.  (The disc tails never take this round: their
        // ~165 stragglers per call finish on the three waves they fill -- a rejected disc attempt
        // is accepted next time with probability 0.785 -- with two barriers instead of three.)
        uint4 *const other = lds.state[parity ^ 1];
        if (tid < ((total + 63) & ~63)) { // whole waves
            bool pend = tid < total;
            Rng wg{0, 0, 0, 0};
            uint32_t ww[6] = {0, 0, 0, 0, 0, 0};
            if (pend) {
                const uint4 ps = state[tid];
                wg = Rng{ps.x, ps.y, ps.z, ps.w};
                for (int trip = 0; trip < kRound1Sphere; ++trip) {
                    if (sphere_attempt(wg, ww)) {
                        pend = false;
                        break;
                    }
                }
                if (!pend) {
                    state[tid] = make_uint4(wg.a_lo, wg.a_hi, wg.b_lo, wg.b_hi);
                    lds.words4[tid] = RF_WORDS4(ww);
                    lds.words2[tid] = RF_WORDS2(ww);
                }
            }
            if (pend) { // as in the park step: one LDS atomic per surviving lane, in units of one entry's 16 bytes
                const int off2 = atomicAdd(&lds.cnt2, 16);
                *entry16(other, off2) = make_uint4(wg.a_lo, wg.a_hi, wg.b_lo, wg.b_hi);
                *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(lds.owner) + (off2 >> 3)) = (uint16_t)tid;
            }
        }
        __syncthreads(); // B2
        const int total2 = __builtin_amdgcn_readfirstlane(lds.cnt2) >> 4;
        if (tid < total2) {
            const uint4 ps = other[tid];
            const int own = lds.owner[tid];
            Rng wg{ps.x, ps.y, ps.z, ps.w};
            uint32_t ww[6] = {0, 0, 0, 0, 0, 0};
            sphere_until_accepted(wg, ww);
            state[own] = make_uint4(wg.a_lo, wg.a_hi, wg.b_lo, wg.b_hi);
            lds.words4[own] = RF_WORDS4(ww);
            lds.words2[own] = RF_WORDS2(ww);
        }
        __syncthreads(); // B3
        if (tid == 0)
            lds.cnt2 = 0;
    } else {
        if (tid < total) {
            const uint4 ps = state[tid];
            Rng wg{ps.x, ps.y, ps.z, ps.w};
            uint32_t ww[6] = {0, 0, 0, 0, 0, 0};
            if (DIM == 2)
                disc_until_accepted(wg, ww);
            else
                sphere_until_accepted(wg, ww);
            state[tid] = make_uint4(wg.a_lo, wg.a_hi, wg.b_lo, wg.b_hi);
            lds.words4[tid] = RF_WORDS4(ww);
            if (DIM == 3)
                lds.words2[tid] = RF_WORDS2(ww);
        }
        __syncthreads(); // B3
    }
    __builtin_amdgcn_s_setprio(0);
    test_skew(tid, 0); // (wave 0, late with its resets)
    if (tid == 0)
        *cnt = 0;
}

// One cooperative call for the stragglers (lane masks `need`) of the kSets pixel sets: park, finish on packed waves,
// collect.  The packed list holds kCoopCap entries; stragglers that do not fit finish their loop in their own wave.
// Returns the number of stragglers the block had (block-uniform).
// WAVE_SLOTS: the stragglers' list slots by ONE LDS atomic per wave and call -- the masks are the ballots, the ranks
// come from v_mbcnt (3 vector instructions per set) -- instead of one LDS atomic per straggler (ds_add_rtn_u32 under the
// lanes' mask hands the lanes of one instruction consecutive values; no vector instruction, but the LDS unit serialises
// the lanes of a same-address atomic).  The counter counts in units of an entry's 16 bytes, so what the atomic returns
// is the entry's byte offset.  (The Makefile passes -mllvm -amdgpu-atomic-optimizer-strategy=None: the compiler's
// atomic optimizer would turn the per-straggler form into exactly the ballot form.)
// FENCED: the caller's disc phase has no block barrier (see SYNCHRONISATION above): `cnt` alternates with the sample
// and the call ends with B4.
template <int DIM, bool WAVE_SLOTS, bool FENCED>
__device__ __forceinline__ int coop_finish2m(CoopLds2 &lds, int parity, int *cnt, const lanemask (&need)[kSets],
                                             Rng (&g)[kSets], uint32_t (&w)[kSets][6], int tid)
{
    asm volatile("" : "+v"(tid)); // keeps the LDS addresses derived from it out of long-lived registers
    uint4 *const state = lds.state[parity];
    int slot[kSets];
    lanemask parked[kSets];
    if (WAVE_SLOTS) {
        int pop[kSets], all = 0;
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            pop[j] = (int)__builtin_popcountll(need[j]);
            all += pop[j];
        }
        int base = 0;
        if (all != 0) { // wave-uniform
            if ((tid & 63) == 0)
                base = atomicAdd(cnt, all * 16);
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
            // (all sets' atomics are issued before the first result is waited for)
            slot[j] = (int)any_u32();
            if (lane_in(need[j]))
                slot[j] = atomicAdd(cnt, 16);
        }
    }
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        parked[j] = need[j] & lanes_where(slot[j] < kCoopCap * 16); // (a ballot of one compare is that compare)
        if (lane_in(parked[j]))
            *entry16(state, slot[j]) = make_uint4(g[j].a_lo, g[j].a_hi, g[j].b_lo, g[j].b_hi);
        if (lane_in(need[j] & ~parked[j])) { // overflow of the packed list: finish in place
            if (DIM == 2) {
                while (!disc_attempt(g[j], w[j])) {
                }
            } else {
                while (!sphere_attempt(g[j], w[j])) {
                }
            }
        }
    }
    __syncthreads(); // B1
    test_skew(tid, (int)(cnt - lds.cnt) + 1); // (a wave late with its read of the counter)
    const int stragglers = __builtin_amdgcn_readfirstlane(*cnt) >> 4; // (a scalar: the branches below are s_cmp)
    const int total = min(stragglers, kCoopCap);
    if (total == 0) // block-uniform
        return 0;
    coop_workers<DIM>(lds, parity, cnt, total, tid);
    test_skew(tid, total + 2); // (a wave late with its collect reads)
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        if (lane_in(parked[j])) {
            uint4 ps = *entry16(state, slot[j]);
            // (opaque: or the first draw's 64-bit sum is fed by a second, 8-byte read of the same entry)
            asm volatile("" : "+v"(ps.x), "+v"(ps.y), "+v"(ps.z), "+v"(ps.w));
            g[j] = Rng{ps.x, ps.y, ps.z, ps.w};
            const uint4 w4 = *entry16(lds.words4, slot[j]);
            w[j][1] = w4.x; w[j][0] = w4.y; w[j][3] = w4.z; w[j][2] = w4.w;
            if (DIM == 3) {
                const uint2 w2 = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(lds.words2) + (slot[j] >> 1));
                w[j][5] = w2.x; w[j][4] = w2.y;
            }
        }
    }
    if (FENCED)
        __syncthreads(); // B4: the reads above and thread 0's resets, before anybody parks for the next sample
    return stragglers;
}

// Disc tails without the block: every wave packs the stragglers of its own three pixel sets (3 x 64 x 0.215 = 41 on
// average) onto its first lanes through its quarter of state[0], finishes them there and hands the results back --
// all of it inside the wave, in LDS order, with no barrier (the block-wide form costs two per sample and makes every
// wave wait for the slowest worker).  An entry's 16 bytes carry the state to the worker, then the accepted draws' four
// words back, then the advanced state back (two round trips through the same slot: nothing else in LDS is this wave's
// alone).  Stragglers beyond the kDiscWaveSlots slots finish in place.
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
        disc_until_accepted(wg, ww);
        region[tid] = RF_WORDS4(ww); // the accepted draws first: the worker keeps the four state words meanwhile
    }
    wave_lds_order();
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        if (lane_in(packed[j])) {
            const uint4 w4 = *entry16(region, slot[j]);
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

// TWO: the environment step's render as one launch (RenderArgs::count2): the blocks of the environments below *count2
// make two passes over their tile -- the step's frame, then the frame of the environment that takes the slot's row in the
// auto-reset's compacted set -- instead of a second launch behind the first one's last blocks (and a host round trip for
// its size).  Everything else is the same code: a pass ends with the pixels' states in memory, the next one reads them.
// A pass must also COMPILE like the single-pass kernel: anything the compiler carries from one pass to the next (the
// arguments, tile and thread geometry: all loop invariants) lives in registers through the sample loop, which has none
// to spare (34 scalar and 19 vector registers spilled that way).  So every pass reads the arguments afresh through a
// kernel-argument pointer and takes block and thread indices through registers the compiler cannot see through.
//
// ACROSS / REGION: the strip of a frame whose width is not a multiple of the tile's (render_kernel_coop2_strip below).
// ACROSS puts a thread's kSets pixels side by side (tTileW apart) instead of below each other; REGION says which columns
// of the frame the launch's tiles of this shape cover: 0 all, 1 [0, strip_x0), 2 [strip_x0, w) (blocks from main_tiles on).
template <bool POW2, int LENS, int WX, int WW, bool ACROSS, int REGION, bool TWO>
__device__ __forceinline__ void render_tile_coop2(const RenderArgs &a_in, CoopLds2 &lds,
                                                  float (&lds_colour)[kColourLds][3][kBlock2])
{
    // tile of a block: WX waves (of WW x 64 / WW pixels) side by side, 4 / WX down, kSets sets
    constexpr int tWaveW = WW, tWaveH = 64 / WW;
    static_assert(WX <= 4, "waves side by side");
    // per-instance choices (measured per frame class, profiles/r03_ab.txt): power-of-two frames finish their disc
    // tails inside the wave and take list slots per wave; the others (float64 pixel coordinates: more registers,
    // more issue-bound) keep the block-wide disc call and per-straggler slots
    constexpr bool kDiscInWave = POW2, kWaveSlots = POW2;
    constexpr int tWavesX = WX, tTileW = WX * tWaveW, tTileH = (4 / WX) * tWaveH;
    // set j lies j * (tDx, tDy) from set 0; the block's whole tile is tAllW x tAllH pixels
    constexpr int tDx = ACROSS ? tTileW : 0, tDy = ACROSS ? 0 : tTileH;
    constexpr int tAllW = ACROSS ? tTileW * kSets : tTileW, tAllH = ACROSS ? tTileH : tTileH * kSets;
    // the frame staging buffer (kSets * 768 B) reuses the words4 array once the sample loop is over
    static_assert(sizeof(lds.words4) >= (size_t)kSets * kBlock2 * 3, "stage does not fit");
    uint32_t *const stage = reinterpret_cast<uint32_t *>(lds.words4);

    if (!TWO && skip_env(a_in.rect, (int)blockIdx.y)) // block-uniform, before any barrier
        return;
    const int passes = (TWO && a_in.env0 + (int)blockIdx.y < *a_in.count2) ? 2 : 1; // block-uniform
    int e = (int)blockIdx.y, block_x = (int)blockIdx.x, tid = threadIdx.x;
  for (int pass = 0; pass < passes; ++pass) {
    RenderArgs a_pass;
    if (TWO) {
        unsigned long long kernarg = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kernarg), "+s"(e), "+s"(block_x));
        asm volatile("" : "+v"(tid));
        __builtin_assume(tid >= 0 && tid < kBlock2);
        // (the first argument: offset 0 of the segment; constant address space: scalar loads.  CONTRACT: RenderArgs is
        // the kernel's first and only explicit argument, passed by value -- see the static_asserts at the __global__ entry
        // points below; the delayed-waves test build also compares what it read with the argument itself)
        a_pass = *(const RenderArgs *)(const __attribute__((address_space(4))) RenderArgs *)kernarg;
#if RF_TEST_SKEW
        if (a_pass.hw != a_in.hw || a_pass.frames != a_in.frames || a_pass.states != a_in.states || a_pass.spp != a_in.spp)
            __builtin_trap();
#endif
    }
    const RenderArgs &a = TWO ? a_pass : a_in;
    const float *const scene_cam = (TWO && pass == 1) ? a.cam_dyn2 : a.cam_dyn;
    const float *const scene_rect = (TWO && pass == 1) ? a.rect2 : a.rect;
    uint8_t *const out_frames = (TWO && pass + 1 < passes) ? a.frames2 : a.frames;
    // (second pass: the first one's row stores still read the stage -- words4 -- while the counters are cleared here;
    // nothing of this pass writes a cooperative array before the barrier below: tests/test_sync_model.py)
    if (tid < 2)
        lds.cnt[tid] = 0;
    if (tid == 2)
        lds.cnt2 = 0;
    __syncthreads();
    const int x_origin = REGION == 2 ? a.strip_x0 : 0, x_end = REGION == 1 ? a.strip_x0 : a.w;
    const int tile_index = block_x - (REGION == 2 ? a.main_tiles : 0);
    const int tiles_x = (x_end - x_origin + tAllW - 1) / tAllW;
    const int tile_y = tile_index / tiles_x, tile_x = tile_index - tile_y * tiles_x;
    // Tiles in the right half of the frame place their waves right to left, so that wave 0 -- which finishes the
    // cooperative tails -- is the outermost wave on both sides of the (centred) target: the one with the fewest hit
    // lanes of its own (+2.6 % measured).
    const bool mirror = REGION == 0 && (2 * tile_x + 1) * tTileW > a.w;

    // Pixel geometry of a thread.  Set j covers the rows tTileH * j further down.  All of it is
    // cheap to derive from the thread index, and the sample loop derives it afresh every
    // iteration from an index the compiler cannot see through (Geometry::opaque): kept alive
    // across the loop these loop invariants -- x, y, their float forms as packed-math pairs,
    // liveness masks, LDS addresses -- are what the register allocator spills.
    struct Geometry {
        int col, row0, x, y0;
        bool live_x;
        int h, w;
        static __device__ __forceinline__ int opaque(int v)
        {
            asm volatile("" : "+v"(v));
            return v;
        }
        __device__ __forceinline__ int x_of(int j) const { return x + j * tDx; }
        __device__ __forceinline__ int y_of(int j) const { return y0 + j * tDy; }
        __device__ __forceinline__ bool live_of(int j) const
        {
            return ACROSS ? (x_of(j) < w && y0 < h) : (live_x && y_of(j) < h);
        }
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
        r.x = x_origin + tile_x * tAllW + r.col;
        r.y0 = tile_y * tAllH + r.row0;
        r.live_x = r.x < a.w;
        r.h = a.h;
        r.w = a.w;
        return r;
    };
    auto pix_of = [&](const Geometry &q, int j) {
        return (size_t)e * a.hw + (q.live_of(j) ? (size_t)q.y_of(j) * a.w + q.x_of(j) : 0);
    };
    Rng g[kSets];
#pragma unroll
    for (int j = 0; j < kSets; ++j) {
        const Geometry g0 = geometry(tid);
        g[j] = rng_load(0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull); // dead lanes: any state
        if (TWO) // (... that is not a loop invariant of the passes: the compiler keeps those in scratch)
            g[j] = Rng{0x7F4A7C15u ^ (uint32_t)tid, 0x9E3779B9u ^ (uint32_t)tid, 0xD192ED03u ^ (uint32_t)tid,
                       0xD1B54A32u ^ (uint32_t)tid};
        if (g0.live_of(j)) {
            const ulonglong2 st = a.states[pix_of(g0, j)];
            g[j] = rng_load(st.x, st.y);
        }
    }
    PixelEnv env0;
    if (TWO) { // (a pointer picked per pass is not one the compiler reads with scalar loads by itself)
        float cam9[9], rect2[2];
#pragma unroll
        for (int i = 0; i < 9; ++i)
            cam9[i] = as_const(scene_cam + (size_t)e * 9)[i];
        rect2[0] = as_const(scene_rect + (size_t)e * 2)[0];
        rect2[1] = as_const(scene_rect + (size_t)e * 2)[1];
        env0 = make_pixel_env(cam9, rect2);
    } else {
        env0 = make_pixel_env(scene_cam + (size_t)e * 9, scene_rect + (size_t)e * 2);
    }
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
        if (j < kColourLds)
            lds_colour[j][0][tid] = lds_colour[j][1][tid] = lds_colour[j][2][tid] = 0.0f;
    }

    int sphere_trips = kCoopTrips2; // in-wave sphere attempts of the current sample (block-uniform)
    lanemask live_m[kSets]; // lanes whose pixel of set j is inside the frame
#pragma unroll
    for (int j = 0; j < kSets; ++j)
        live_m[j] = lanes_where(geometry(tid).x_of(j) < a.w) & lanes_where(geometry(tid).y_of(j) < a.h);
    // per-environment conditions as scalars (a uniform `bool` is a lane mask that vector instructions test)
    const int tmiss_s = __builtin_amdgcn_readfirstlane((int)env0.tmiss);
    for (int k = 0; k < a.spp; ++k) {
        const PixelEnv &env = env0;
        uint32_t w[kSets][6];
        float s[kSets], t[kSets];
        lanemask need_m[kSets];
        const Geometry gk = geometry(Geometry::opaque(tid));
#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            // ((float)y of the further sets by an exact float addition: conversions issue on the slow path)
            sample_coords<POW2>(g[j], gk.x_of(j), gk.y_of(j), ACROSS ? (float)gk.x + (float)(j * tDx) : (float)gk.x,
                                ACROSS ? (float)gk.y0 : (float)gk.y0 + (float)(j * tDy), a.fc, s[j], t[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w[j][i] = any_u32();
            // (every lane makes the attempt -- a wave's instructions cost the same with any lanes off, and a lane
            // outside the frame never stores its state)
            const float sq = disc_attempt_sq(g[j], w[j]);
            need_m[j] = live_m[j] & ~lanes_where(sq < 1.0f);
        }
        if (kDiscInWave)
            disc_tails_wave(lds.state[0], need_m, g, w, tid);
        else
            coop_finish2m<2, kWaveSlots, false>(lds, 0, &lds.cnt[0], need_m, g, w, tid);

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
            w[j][4] = any_u32();
            w[j][5] = any_u32();
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
        // (with in-wave disc tails the sphere call's counter alternates with the sample: SYNCHRONISATION, rf_coop2.h top)
        int *const sphere_cnt = &lds.cnt[kDiscInWave ? (k & 1) : 1];
        const int stragglers = __builtin_amdgcn_readfirstlane(
            coop_finish2m<3, kWaveSlots, kDiscInWave>(lds, 1, sphere_cnt, need_m, g, w, tid));
        if (sphere_trips == kCoopTrips2 && stragglers > kCoopCap + kAdaptOn)
            sphere_trips = kCoopTrips2 + 1;
        else if (sphere_trips != kCoopTrips2 && 2 * stragglers < kCoopCap + kAdaptOff)
            sphere_trips = kCoopTrips2;

#pragma unroll
        for (int j = 0; j < kSets; ++j) {
            float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f;
            const bool hit = lane_in(hit_m[j]);
            if (hit)
                sphere_finish(w[j], q0, q1, q2);
            const Colour c = sample_axis_shade(hit, red[j], rdx[j], rdy[j], rdz[j], q0, q1, q2);
            if (j < kColourLds) {
                // (read + add + write: the LDS unit's own float add, ds_add_f32, is bit-identical and 2.5x slower end to end)
                lds_colour[j][0][tid] = add2_not_negzero(lds_colour[j][0][tid], c.r);
                lds_colour[j][1][tid] = add2_not_negzero(lds_colour[j][1][tid], c.g);
                lds_colour[j][2][tid] = add2_not_negzero(lds_colour[j][2][tid], c.b);
                continue;
            }
            cr[j] = add2_not_negzero(cr[j], c.r);
            cg[j] = add2_not_negzero(cg[j], c.g);
            cb[j] = add2_not_negzero(cb[j], c.b);
        }
    }
#pragma unroll
    for (int j = 0; j < kColourLds && j < kSets; ++j) {
        cr[j] = lds_colour[j][0][tid];
        cg[j] = lds_colour[j][1][tid];
        cb[j] = lds_colour[j][2][tid];
    }
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
            const int slot = (j * tDy + ge.row0) * tAllW + j * tDx + ge.col;
            sb[slot * 3 + 0] = r8;
            sb[slot * 3 + 1] = g8;
            sb[slot * 3 + 2] = b8;
        } else if (ge.live_of(j)) {
            uint8_t *dst = out_frames + pix_of(ge, j) * 3;
            dst[0] = r8;
            dst[1] = g8;
            dst[2] = b8;
        }
    }
    if ((a.w & 3) == 0) {
        // the tile's rows (tAllW * 3 B each) -> LDS -> coalesced dword stores per row
        __syncthreads();
        constexpr int kRowDw = tAllW * 3 / 4;
        const int x_tile = x_origin + tile_x * tAllW; // (a multiple of 4, like w: every row segment starts on a dword)
        for (int i = tid; i < tAllH * kRowDw; i += kBlock2) {
            const int r = i / kRowDw, d = i - r * kRowDw;
            const int yy = tile_y * tAllH + r;
            const int valid_dw = min(tAllW, a.w - x_tile) * 3 / 4; // w % 4 == 0
            if (yy < a.h && d < valid_dw) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(out_frames + (((size_t)e * a.h + yy) * a.w + (size_t)x_tile) * 3);
                dst[d] = stage[r * kRowDw + d];
            }
        }
    }
  } // pass
}

// (two-pass tile code re-reads the kernel arguments from offset 0 of the kernarg segment: a kernel that uses it takes
// exactly one argument, a RenderArgs by value)
template <class F>
struct takes_render_args_only : std::false_type {};
template <>
struct takes_render_args_only<void (*)(RenderArgs)> : std::true_type {};

template <bool POW2, int LENS, int WX = kWavesX, int WW = kWaveW, bool TWO = false>
__global__ __launch_bounds__(kBlock2, kSetsOcc) void render_kernel_coop2(RenderArgs a_in)
{
    static_assert(takes_render_args_only<decltype(&render_kernel_coop2<POW2, LENS, WX, WW, TWO>)>::value,
                  "render_tile_coop2<.., TWO> reads RenderArgs from offset 0 of the kernarg segment");
    __shared__ CoopLds2 lds;
    // colour accumulators of the first kColourLds pixel sets live in LDS (one read-modify-write
    // per sample and channel, off the vector ALU) to keep the kernel inside its VGPR budget
    __shared__ float lds_colour[kColourLds][3][kBlock2];
    render_tile_coop2<POW2, LENS, WX, WW, false, 0, TWO>(a_in, lds, lds_colour);
}

// Frames whose width is not a multiple of 64 (the reference's default 300 x 300 among them).  With tiles of one shape the
// last tile column covers the remainder plus dead lanes that cost what live ones cost -- 20 of 320 columns at 300 px --;
// this kernel renders columns [0, strip_x0 = w - w % 64) with the tiles of render_kernel_coop2<false, LENS, MAIN_WX, 32>
// (128 x 6 where strip_x0 is a multiple of 128, else 64 x 12) and the remaining w % 64 <= 48 columns with tiles of 48 x 16: four waves of 16 x 4 pixels below each other, a
// thread's three pixels 16 columns apart (44 of 48 columns live at 300 px, 304 of 300 rows: 1.6 % of the lanes dead
// instead of 6.7 %).  One launch: blocks [0, main_tiles) of a row of the grid are main tiles, the rest strip tiles.
// (Always the two-pass form of the tile code: a single pass is *count2 == 0.)
template <int LENS, int MAIN_WX = 2>
__global__ __launch_bounds__(kBlock2, kSetsOcc) void render_kernel_coop2_strip(RenderArgs a_in)
{
    static_assert(takes_render_args_only<decltype(&render_kernel_coop2_strip<LENS, MAIN_WX>)>::value,
                  "render_tile_coop2<.., TWO> reads RenderArgs from offset 0 of the kernarg segment");
    __shared__ CoopLds2 lds;
    __shared__ float lds_colour[kColourLds][3][kBlock2];
    if (skip_env(a_in.rect, (int)blockIdx.y)) // (slots a launch for all n environments has to leave alone: never a real rectangle)
        return;
    if ((int)blockIdx.x < a_in.main_tiles) // block-uniform
        render_tile_coop2<false, LENS, MAIN_WX, 32, false, 1, true>(a_in, lds, lds_colour);
    else
        render_tile_coop2<false, LENS, 1, 16, true, 2, true>(a_in, lds, lds_colour);
}

} // namespace rf
