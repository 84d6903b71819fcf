// smm_viterbi.hip -- factored semi-Markov Viterbi for gfx950 (wave64, fp64 VALU, no MFMA).
//
// Replaces SemiMarkovModule.log_hsmm (reference semimarkov_modules.py:416-523) + torch_struct
// SemiMarkovCRF.argmax / from_parts (modules:677-679) + class un-mapping (modules:683-691) +
// spans_to_labels / trim (semimarkov_utils.py:51-63, modules:532-543).  The recurrence, its association
// order and the arg-max order are specified in oracle/smm_oracle.c (the CPU twin); they are repeated here
// only where the mapping to the hardware needs them.
//
// Mapping.  One workgroup per video with two kinds of waves:
//
//   PUSHER waves (waves 1..NW-1), each owning SPW states ("one wavefront per (video, state) row").  The DP runs in
//   PUSH form so that nothing K-proportional is ever reduced across lanes or leaves the register file:
//     * ring slot p = (n-1) mod RING (RING = 64*R >= kp) holds the accumulator A[n][c] = max_k (h[n-k][c] + len[k][c])
//       of a future position n, in lane p/R, register p%R of the state's wave;
//     * when h[s][c] is final it is a wave-uniform value; every slot does A = max(A, h[s] + len[k]), k = n - s:
//       2 fp64 VALU ops per lattice cell (v_add_f64, v_max_f64), nothing else;
//     * k shrinks by one per step for every slot, so the length table lives in registers too and is ROTATED one slot
//       per step: R-1 registers are renamed (the loop is unrolled, so statically) and one crosses to the next lane
//       with a single DPP wave_ror:1.
//
//   one CHAIN wave (wave 0) that owns the serial part for ALL states, one lane per state: cumE += elp, gamma = cumE + A,
//   the C x C transition, h = beta - cumE, and the history.  The transition is lane = target state: gamma[.] is
//   broadcast through LDS and every lane folds its own row of the transition table (registers) over it; the two
//   halves of the wave take half of the source states each and are merged with one v_permlane32_swap.
//
// Hand-over in BLOCKS of B positions, one barrier per block.  The chain wave evaluates the K0 = 2B+D-1 shortest segment
// lengths itself (h[n-1..n-K0] and len[1..K0] of its state sit in its registers; only the k = 1 term is on the serial
// path); the pushers own k > K0 (their length rings hold -inf for k <= K0).  That slack is what decouples the two:
// during block j the chain turns positions jB+1..(j+1)B into h values from the A' the pushers delivered a block
// earlier, WHILE the pushers push the B sources of block j-1 through their rings and deliver A' of block j+1 (complete
// for sources <= n-K0-1).  max is exact, every candidate is the same expression h[s] + len[k] wherever it is evaluated,
// so the split changes nothing in the result.
//
// The forward pass keeps VALUES only.  The arg-max is recovered afterwards along the optimal path only (one row
// scan per SEGMENT instead of arg-max tracking per cell) from the history, re-evaluating exactly the expressions of
// the forward pass, so it is bit-identical to tracking back-pointers.
//
// HBM traffic per frame (c states): read elp 8c, write history 24c (cumE and gamma frame-major, h state-major), label 8 B.
//
// K > 512: BAND mode below -- 128-slot rings shared by nine length bands, eight of them skipped by an exact bound test,
// blocks of 8 positions, up to 32 states on one CU.  (Rounds 1-3 also had 1024-slot rings and "gangs" of two or three
// workgroups per video that exchanged rows through agent-scope atomics; BAND mode replaced both and round 4 removed them.)
#include <type_traits>
#include "smm_device.h"

// Diagnostic build only (-DSMM_PROFILE, never shipped or timed): per wave of workgroup 0, cycles between leaving a
// block barrier and arriving at the next one ("busy") and cycles spent in the barrier, summed over the blocks and
// written behind the workspace's error word (uint64 slots 8+w and 24+w; slot 7 = number of blocks).
#ifdef SMM_PROFILE
#define SMM_PROF_DECL unsigned long long p_busy = 0, p_wait = 0, p_t0 = __builtin_readcyclecounter(), p_ph[4] = {0, 0, 0, 0}, p_mx = 0
#define SMM_BLOCK_BARRIER()                                                        \
    do {                                                                          \
        const unsigned long long t1 = __builtin_readcyclecounter();               \
        __syncthreads();                                                          \
        const unsigned long long t2 = __builtin_readcyclecounter();               \
        p_busy += t1 - p_t0; p_wait += t2 - t1; p_t0 = t2;                        \
    } while (0)
#define SMM_LDS_BARRIER()                                                          \
    do {                                                                          \
        const unsigned long long t1 = __builtin_readcyclecounter();               \
        smm_lds_barrier();                                                        \
        const unsigned long long t2 = __builtin_readcyclecounter();               \
        p_busy += t1 - p_t0; p_wait += t2 - t1;                                   \
        if (SMM_PROFILE == 2) { if (t2 - t1 < 150) { p_mx += 1; p_ph[jj & 3] += 1; p_ph[0] += (t1 - p_t0) << 20; } } \
        else { p_ph[jj & 3] += t1 - p_t0; p_mx = (t1 - p_t0 > p_mx) ? t1 - p_t0 : p_mx; } \
        p_t0 = t2;                                                                \
    } while (0)
// (BAND mode, 8 waves: slots 16+w = the longest block, 32+4w+ph = busy cycles of the blocks with j mod 4 = ph)
#define SMM_PROF_OUT()                                                            \
    if (blockIdx.x == 0 && lane == 0) {                                           \
        unsigned long long *pp = reinterpret_cast<unsigned long long *>(a.err);   \
        pp[8 + w] = p_busy; pp[24 + w] = p_wait; pp[7] = (unsigned long long)J;   \
        if (BAND) { pp[16 + w] = p_mx; for (int ph = 0; ph < 4; ++ph) pp[32 + 4 * w + ph] = p_ph[ph]; } \
    }
#else
#define SMM_PROF_DECL do { } while (0)
#define SMM_BLOCK_BARRIER() __syncthreads()
#define SMM_LDS_BARRIER() smm_lds_barrier()
#define SMM_PROF_OUT() do { } while (0)
#endif

// Block barrier of the BAND mode: what the waves hand each other per block lives in LDS only, so the barrier waits for
// the wave's LDS operations (lgkmcnt) and NOT for its vector-memory operations.  __syncthreads() is a workgroup-scope
// release + barrier + acquire, i.e. s_waitcnt vmcnt(0) in front of every s_barrier: the mover wave then stalls once per
// block until the elp rows it has just asked for (two blocks ahead, on purpose) have arrived from HBM, and seven waves
// wait for it.  Global data that does change hands inside the workgroup is ordered otherwise: the history rows the
// delayed bands read (agent-scope loads, two blocks ahead of their pushes) were stored by the mover wave at least
// (SMM_BAND_DELAY - 2B) / B - 1 = 11 block steps earlier (blocks of B = 8 positions, delay 112: see the static_assert in
// the kernel), and the mover has since waited, in EVERY one of those steps, for the elp loads it issued a step before --
// loads YOUNGER than the stores, and vmcnt retires a wave's vector-memory operations in issue order: the stores of step s
// are through before the mover leaves step s + 2.  The back-trace starts behind real __syncthreads().
__device__ __forceinline__ void smm_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifndef SMM_B
#define SMM_B 4   // positions per hand-over block (development builds override it)
#endif
#ifndef SMM_D
#define SMM_D 1
#endif
#ifndef SMM_MOVER_STATES
#define SMM_MOVER_STATES 0   // 1 (A/B aid): the mover wave of the 8-wave kernels always owns states, as in rounds 1-2
#endif
#ifndef SMM_CHAIN_DUAL
#define SMM_CHAIN_DUAL 1     // 0 (A/B aid): the chain wave folds the launch's HF sources per lane group for every video
#endif
#ifndef SMM_ABLATE
#define SMM_ABLATE 0   // development builds only (results are WRONG, timing experiments): bit 0 chain wave without the
                       // candidates k = 2..K0, bit 1 without the cumE add / store, bit 2 pushers push nothing, bit 3 mover
                       // keeps no books, bit 4 transition over a third of the sources
#endif
#ifndef SMM_SPEC
#define SMM_SPEC 1       // the chain wave's speculative transition (see SPEC in the kernel); 0: compiled out (A/B aid)
#endif
#ifndef SMM_PIPE0_ALWAYS
#define SMM_PIPE0_ALWAYS 0
#endif
#ifndef SMM_SPEC8
#define SMM_SPEC8 0      // the chain wave's BLOCK-level speculation (see SPEC8 in the kernel).  Built, bit-exact (every GPU test passes
#endif                   // with it) and NOT shipped: same-box A/B on three boxes, cfg3 DP kernel +2.2 / +2.3 / +3.8 % with it (round 5,
                         // profiles/round5_chain_spec8_anchor.txt) -- the chain wave's busy time falls (2716 -> 2380 cycles per block)
                         // and the block does not get shorter: its dense instruction stream takes issue slots from the mover wave on
                         // the same SIMD (1650 -> 2130 busy), which becomes the wave the barrier waits for
#ifndef SMM_DOM_SPARSE
#define SMM_DOM_SPARSE 3   // BAND pushers (DOM): fewer unbeaten sources than this (besides the last) are pushed one by one from the LDS table
#endif
#ifndef SMM_ANCHOR
#define SMM_ANCHOR 0       // BAND pushers: anchor dominance in band 0 (see ANCHOR in the pusher waves).  Built, bit-exact, and NOT
#endif                     // shipped: it leaves out what it was meant to (sources pushed per (state, block) on cfg3 1.75 -> 1.31, the
                           // leading state's 7.9 -> 1.8) and the kernel is 4 % SLOWER with it (same box, cfg3 DP 2.52 -> 2.62 ms;
                           // 64 x 4096 lattices, 23 states: 229 -> 253 ns per frame): the pusher waves' ~2000 busy cycles per block
                           // are not those pushes (profiles/round5_chain_spec8_anchor.txt)
#ifndef SMM_BAND_ALLWIT
#define SMM_BAND_ALLWIT 1  // 0 (A/B aid): the band skip test with the one witness of round 3 (group G - 2) only
#endif
#ifndef SMM_HQ_ALWAYS
#define SMM_HQ_ALWAYS 0    // 1 (A/B aid): the delayed sources are fetched in every block, as in round 3, whether a band is on or not
#endif
#ifndef SMM_B8_R
#define SMM_B8_R 4       // blocks of 8 positions for the 256-slot rings (round 3, same box: cfg2 DP 0.481 -> 0.456 ms; the 64-slot
#endif                   // rings of cfg4 measured 10 % SLOWER with them, 0.406 -> 0.445 ms, and keep blocks of 4)

// wave-level lexicographic arg-max: larger val first, then smaller k, then smaller c
__device__ __forceinline__ void smm_best3(double &v, int &k, int &c, double v2, int k2, int c2)
{
    bool take = (v2 > v) || (v2 == v && (k2 < k || (k2 == k && c2 < c)));
    if (take) { v = v2; k = k2; c = c2; }
}

__device__ __forceinline__ void smm_wave_best3(double &v, int &k, int &c)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double v2 = __shfl_xor(v, off);
        int k2 = __shfl_xor(k, off);
        int c2 = __shfl_xor(c, off);
        smm_best3(v, k, c, v2, k2, c2);
    }
}

// One source step of one state's ring.  Push step t = s + B - 1 (s = source position), u = t mod R (static after
// unrolling): logical register r of the length ring lives in physical register (r - u) mod R, so a step renames R-1
// registers and moves one across lanes.
template <int R>
__device__ __forceinline__ void smm_push(double (&A)[R], double (&L)[R], double hs, int u)
{
    // adds and maxes in groups of four independent registers
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += 4) {
        double tq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (r0 + q < R) tq[q] = hs + L[(r0 + q - u + R) % R];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (r0 + q < R) A[r0 + q] = smm_fmax(A[r0 + q], tq[q]);
    }
    L[(2 * R - 1 - u) % R] = smm_wave_ror1(L[(2 * R - 1 - u) % R]);
}

// Block j of one state's ring (jj = j mod UB, static): push B sources, then hand A' of block j+1 to LDS and clear
// those slots (everything they still receive before they wrap is -inf).  D = 1: the first source is the LAST row of the
// previous block, kept in a register (hd), so the pushes that follow a barrier do not wait for LDS; the last row of
// this block is kept for the next one.
template <int R, int B, int D, bool TRIB = false>
__device__ __forceinline__ void smm_ring_block(double (&A)[R], double (&L)[R], double &hd, const double *h_blk,
                                               double *a_blk, int j, int jj, int lane, const double &ninf)
{
    // ninf: -inf in a register pair the caller keeps alive (one v_mov_b64 per cleared slot; the literal would cost two
    // v_mov_b32 each, 12 VALU slots per block of a 3-state pusher)
    constexpr int RING = 64 * R;
    double hv[B];
#pragma unroll
    for (int i = 0; i < B; ++i) hv[i] = h_blk[i * 2 * SMM_MAX_STATES_DEV];      // (h_blk: the h half of sh_gh's (gamma, h) pairs)
    if constexpr (D == 1) {
        smm_push<R>(A, L, hd, (jj * B) % R);
#pragma unroll
        for (int i = 1; i < B; ++i) smm_push<R>(A, L, hv[i - 1], (jj * B + i) % R);
        hd = hv[B - 1];
    } else {
#pragma unroll
        for (int i = 0; i < B; ++i) smm_push<R>(A, L, hv[i], (jj * B + i) % R);
    }
    if constexpr (R % B == 0) {
        // the B slots share a lane
        if (lane == (((j + 1) * B) & (RING - 1)) / R) {
#pragma unroll
            for (int i = 0; i < B; ++i) {
                const int r = ((jj + 1) * B + i) % R;
                a_blk[i * SMM_MAX_STATES_DEV] = A[r];
                A[r] = ninf;
            }
        }
    } else if constexpr (B % R == 0) {
        // the B slots are all R registers of B/R consecutive lanes
        const int d = lane - (((j + 1) * B) & (RING - 1)) / R;
        if (d >= 0 && d < B / R) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                a_blk[(d * R + r) * SMM_MAX_STATES_DEV] = A[r];
                A[r] = ninf;
            }
        }
        if constexpr (TRIB) {
            // triangular split (see TRI in the kernel): the slots handed over a block ago have since received lengths that
            // were meant for their old targets; nothing real reaches them before this second clearing
            if (((d + 64) & 63) >= 64 - B / R) {
#pragma unroll
                for (int r = 0; r < R; ++r) A[r] = ninf;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const int r = ((jj + 1) * B + i) % R;
            if (lane == (((j + 1) * B + i) & (RING - 1)) / R) {
                a_blk[i * SMM_MAX_STATES_DEV] = A[r];
                A[r] = ninf;
            }
        }
    }
}

// Length ring of one state at push step 0: slot p waits for k = (p + off) mod RING and takes part for kmin <= k <= kmax.
template <int R>
__device__ __forceinline__ void smm_ring_init_range(double (&A)[R], double (&L)[R], const double *len_col, int cm, int off,
                                                    int kmin, int kmax, bool on, int lane)
{
    constexpr int RING = 64 * R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = (lane * R + r + off) & (RING - 1);
        A[r] = SMM_NEG_INF;
        L[r] = (on && k >= kmin && k <= kmax) ? len_col[(size_t)k * cm] : SMM_NEG_INF;
    }
}

// Block protocol (source position -(B-1+D) at push step 0): slot p waits for k = (p + B + D) mod RING; lengths up to
// K0 = 2B+D-1 belong to the chain wave.
template <int R, int B, int D, bool TRIB = false>
__device__ __forceinline__ void smm_ring_init(double (&A)[R], double (&L)[R], const double *len_col, int cm, int kp,
                                              bool on, int lane)
{
    smm_ring_init_range<R>(A, L, len_col, cm, B + D, TRIB ? B + D + 1 : 2 * B + D, kp - 1, on, lane);
}

__device__ __forceinline__ double smm_ld_agent(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------
// BAND mode (BAND = 1): one workgroup per video at K > 512, the lattice cut along the segment length into BANDS that share
// ONE ring of 128 upcoming targets per state, and whole bands skipped -- exactly -- while they cannot matter.
//   band 0     lengths up to 127 that the chain wave does not evaluate itself (see TRI in the kernel), sources
//              undelayed: always evaluated (what a gang leader's short rings do);
//   band m>=1  lengths 16+112m .. 127+112m (m = 1..8 covers 128..1023), fed with the source of 112m positions ago:
//              slot p of the shared ring, waiting for target n at ring distance kr = n - s, receives
//              h[s - 112m] + len[kr + 112m]; ring distances below 16 carry -inf in these bands (a slot that was handed
//              over changes targets; with blocks of 8 the slots within 15 of the current source have been, and a band's
//              own lengths never come that close).
// The accumulators A are shared by the bands of a state (all of them aim at the same 128 targets): 2 registers per
// state for A and none for the length rings -- band 0's comes from an LDS table at the phase of every push that happens
// (round 4: a source that its successor beats at every target is not pushed at all, see DOM in the pusher waves), the
// delayed bands' from the state-major table in L2.
// Blocks are 8 positions and the pushers push the block's own 8 sources (B = 8, D = 0): half as many block barriers as
// with blocks of 4, and the barrier is where a block's slowest wave makes the other seven wait.
//
// Skipping.  Per state and group of 16 sources (2 hand-over blocks) the mover wave keeps hm[g] = max h over the group
// (LDS ring of 64 groups, lane = state).  Two blocks before the sources of group G are pushed, every pusher wave decides
// for its own states (lane = (state, band)): band m is switched off for the group when
//     hm[G - 7m] + max_{k in band m} len[k]   <=   hm[G - 2] + min_{33 <= k <= 174} len[k].
// Left: an upper bound of every candidate the band would push (its 16 sources are group G - 7m).  Right: for every
// target n the band can reach, n in [16G + 16, 16G + 142], the best source s* of group G - 2 (the newest group that is
// complete when the decision is due) is a real candidate of A[n] with 33 <= n - s* <= 174, so the right side is a lower
// bound of the final A[n] (all of it needs kp - 1 >= 174; otherwise the bound is -inf and nothing is skipped).  max is
// exact and rounding is monotone; a skipped candidate may tie with a maximum, but then its witness attains it too and
// is either evaluated (lengths <= 127 always are) or has a newer witness of its own: not a bit of the result changes,
// whatever the inputs are.  What changes is the work: on CrossTask-shaped data ~99 % of the delayed band-groups are
// skipped (14 % of the lattice cells are evaluated, profiles/round3_prune_survival.txt), so the frame time no longer
// grows with K and hardly with the state count.
// A band that is switched on reads its length ring from the state-major length table (len_t) at the phase of every
// block -- no registers are kept for the delayed bands; the first two rings of a wave are fetched a block ahead --; the
// h rows of the delayed sources come from the history the mover wave writes anyway (lane = (state, band), 8 positions
// each, fetched TWO blocks ahead with agent-scope loads and handed to the pushes with v_readlane).
// (SMM_BAND_DELAY = 112, SMM_BAND_LO = 16, SMM_BAND_N = 8 bands, SMM_BAND_TAB: smm_device.h)

// Lane N of every row of 16 lanes, broadcast to its row (one v_mov_b64_dpp row_newbcast: the only DPP control the 64-bit
// ALU takes).
__device__ __forceinline__ double smm_row_bcast(double x, int n)   // n: a constant after unrolling (the switch folds away)
{
    // (`old` is never read -- every lane is written -- but an operand tied to x would cost a copy of x per broadcast:
    // a fresh undefined register instead)
    double u;
    asm volatile("" : "=v"(u));
#define SMM_RB(k) case k: return __builtin_amdgcn_update_dpp(u, x, 0x150 + k, 0xf, 0xf, false);
    switch (n & 15) {
        SMM_RB(0) SMM_RB(1) SMM_RB(2) SMM_RB(3) SMM_RB(4) SMM_RB(5) SMM_RB(6) SMM_RB(7)
        SMM_RB(8) SMM_RB(9) SMM_RB(10) SMM_RB(11) SMM_RB(12) SMM_RB(13) SMM_RB(14)
        default: return __builtin_amdgcn_update_dpp(u, x, 0x150 + 15, 0xf, 0xf, false);
    }
#undef SMM_RB
}

// Length ring of band m at the phase of a block: slot p = 2 lane + r is at ring distance kr = (p + off) & 127, off = B + D
// - (push step).  The state-major table is stored shifted by one (row[k + 1] = len[k]): one 16-byte load per lane.  A
// delayed band keeps NO ring between blocks -- it is read again from the table (8 KB per state, L2-resident) at the phase
// of every block it is switched on for.
__device__ __forceinline__ void smm_band_ring_load(double (&L)[2], const double *lent_row, int off, int m, int kp, int lane)
{
    const int kr0 = (lane * 2 + off) & 127;                       // (kr0 = 127: the second slot wraps to 0, masked)
    const int k0 = kr0 + SMM_BAND_DELAY * m;
    const double2 v = *reinterpret_cast<const double2 *>(lent_row + k0 + 1);
    L[0] = (kr0 >= SMM_BAND_LO && k0 <= kp - 1) ? v.x : SMM_NEG_INF;
    L[1] = (kr0 != 127 && kr0 + 1 >= SMM_BAND_LO && k0 + 1 <= kp - 1) ? v.y : SMM_NEG_INF;
}

// R   ring registers per lane (RING = 64 R >= kp)      SPW  states per pusher wave
// NW  waves per workgroup (1 chain + NW-1 pushers)       HF   source states per lane group of the chain wave: 8, 12 or 16
//     (two groups of 32 lanes) or 4 (launches of at most 16 states: FOUR groups of 16 lanes, merged by a
//     v_permlane16_swap on top of the v_permlane32_swap: 8 instructions fewer per position, same-box A/B: cfg2 DP -1 %,
//     cfg4 DP -2.8 %)
// One workgroup per CU is all that fits (and all that is wanted): tell the register allocator it may use the whole
// architected VGPR budget of NW/4 waves per SIMD instead of spilling for an occupancy nobody asked for.
// B   positions per hand-over block; D = 1: pushers lag one more source (see smm_ring_block); the chain wave evaluates
//     lengths 1..2B+D-1 itself
// BAND  BAND mode (above): one workgroup per video at K > 512, 128-slot rings, SPW = states per state-owning pusher wave (3..6)
// TAG  1: the same kernel under a second name -- the repair launch of a time-split decode (smm_chunk.hip), kept apart from the
//      launch it repairs in per-kernel statistics (it nearly always returns at once)
template <int R, int SPW, int NW, int HF, int B, int D = SMM_D, bool BAND = false, int TAG = 0>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu((BAND && NW == 4) ? 2 : 1, (BAND && NW == 4) ? 2 : (NW + 3) / 4)))
smm_viterbi_kernel(SmmDpArgs a)
{
    static_assert(!BAND || (R == 16 && (NW == 8 || NW == 4) && B == 8 && D == 0),
                  "band mode: K <= 1024, 8 waves, blocks of 8 positions, the pushers push the block's own sources");
    static_assert(BAND || R < 16, "rings of 1024 slots exist in BAND mode only");
#ifndef SMM_TRI
#define SMM_TRI 1
#endif
    // TRI (blocks of 8: BAND mode, D = 0, and the 256-slot rings, D = 1): a TRIANGULAR split of the short lengths.  When
    // target n = (j+1)B + 1 + i is handed over the pushers have pushed every source up to jB - D into its slot, i.e. every
    // length k >= B + D + 1 + i -- if the rings carry the lengths from B + D + 1 on.  The chain wave then evaluates
    // k <= B + D + i only (B(B-1)/2 candidates per block less than with the uniform split at K0 = 2B + D - 1).  A slot keeps
    // receiving those lengths for up to B - 1 pushes after its hand-over, addressed to a target that is gone: the slots
    // are cleared a second time one block later, before anything real reaches them (a ring's own lengths come back at
    // its far end, the delayed bands of BAND mode stay off ring indices < 16).
    constexpr bool TRI = SMM_TRI && ((BAND && D == 0) || (!BAND && B == 8 && B % R == 0 && NW == 8));
    constexpr int K0 = 2 * B + D - 1;                      // segment lengths the chain wave evaluates itself
    constexpr int NP = NW - 1;
    constexpr int UB = (R / B) > 2 ? (R / B) : 2;          // blocks per unrolled pusher iteration (UB*B % R == 0, UB even)
    constexpr int M = (D ? 4 : 2) * B;                     // chain wave: h[n] of the last M > K0 positions, slot n mod M
    constexpr int MW = (NW >= 8) ? 4 : 1;                  // the wave that moves HBM traffic (shares the chain wave's SIMD)
    constexpr int SN = (HF == 4) ? 16 : 2 * HF;            // states the kernel's launches hold at most
#ifdef SMM_DIAG_V255
    asm volatile("v_mov_b32 v255, 0" ::: "v255");      // (diagnostic: the kernel allocates all 256 VGPRs)
#endif
#ifdef SMM_DIAG_SCRATCH
    {   // (diagnostic: the kernel needs a private segment)
        volatile int junk_[8];
        junk_[threadIdx.x & 7] = (int)threadIdx.x;
        if (junk_[(threadIdx.x + 1) & 7] == 123456789) a.err[5] = 1;
    }
#endif
#ifdef SMM_PROFILE
    const unsigned long long p_kern0 = __builtin_readcyclecounter();   // (diagnostic: phases of workgroup 0 -> slots 44..46)
#endif
    // (the repair launch of a time-split decode, smm_chunk_stitch_kernel below: one workgroup per split video, at work only
    // where the stitch asked for the video to be decoded again in one piece)
    if (a.redo && a.redo[blockIdx.x] == 0) return;
    const int vid = a.order[blockIdx.x];
    SmmVideo mv = a.videos[vid];
    if (a.flags & 8) mv.T -= 1;               // no EOS: the DP covers the frames before the last one (smmdp.h)
    // CHUNK units (round 5; see smm_chunk_stitch_kernel): a long video cut along the TIME axis.  A unit is the forward pass
    // over positions a0 .. a0 + T of its parent video, with its own history block; nothing else -- closing step, back-trace
    // and every per-video output belong to the stitch kernel.  The first unit of a video starts like the video (a0 = 0);
    // the others start from a GUESS, "a segment boundary at a0, every state equally good" (h[a0][c] = 0), which the
    // recursion forgets within a segment or two (profiles/round5_rank_convergence.txt), and carry the SERIAL prefix sums on:
    // cumE[a0][c] comes from smm_cum_anchor_kernel (the same additions in the same order as an unsplit decode makes), so the
    // unit's cumE rows are the unsplit decode's, bit for bit.
    const bool chunk = (mv.pad & 1) != 0, chunk_flat = (mv.pad & 2) != 0;
    const double *anchor = chunk_flat ? a.chunk_anchor + (size_t)(vid - a.b_videos) * a.c_max : nullptr;
    const int T = mv.T;
    const int g = mv.group;
    const int C = a.n_states[g];
    const int cm = a.c_max;
    const int kp = mv.kp;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform on purpose: scalar branches
    const int lane = threadIdx.x & 63;

    const double *trans = a.trans + (size_t)g * cm * cm;
    const double *init = a.init + (size_t)g * cm;
    const double *len = a.len + (size_t)g * a.k_rows * cm;
    const double *elp = a.elp + (size_t)mv.frame_off * cm;
    const double *endpen = a.endpen ? a.endpen + (size_t)vid * cm : nullptr;
    const int64_t *cmap = a.class_map ? a.class_map + (size_t)g * (cm + 1) : nullptr;
    // history of this video: rows are C wide (the video's own state count, not the launch's c_max: a launch that mixes
    // 11- and 23-state tasks would otherwise write twice the bytes for the former)
    double *hcum = a.hist + mv.hist_off;                  // [T+1][C]  cumE[n][c]
    double *hh = hcum + (size_t)C * (T + 1);              // [C][T+1]  h[n][c], STATE-major: the back-trace scans one
                                                          // state's column over up to kp-1 positions per segment
    double *hgam = hh + (size_t)C * (T + 1);              // [T+1][C]  gamma[n][c]
    int64_t *spans = (a.spans && !chunk) ? a.spans + (size_t)vid * (a.t_max + 1) : nullptr;
    int64_t *labels = (a.labels && !chunk) ? a.labels + mv.frame_off : nullptr;

    // block q = positions qB+1 .. (q+1)B, buffer q & 1
    __shared__ __attribute__((aligned(16))) double sh_apart[2][B][SMM_MAX_STATES_DEV];   // A'[n][c]   pushers -> chain
    // (gamma[n][c], h[n][c]) pairs: ONE 16-byte store per position by the chain wave (LDS instructions are what its fast
    // positions are made of: round 4 measured ~40 cycles of the wave's time per ds_write_b64 on a CU whose other seven waves
    // use the LDS too); h -> pushers and HBM, gamma -> HBM.  cumE never goes through LDS: the mover wave adds up the elp
    // rows it staged itself -- the same additions in the same order, hence the same bits.
    __shared__ __attribute__((aligned(16))) double sh_gh[2][B][SMM_MAX_STATES_DEV][2];
    __shared__ __attribute__((aligned(16))) double sh_e[2][B][SMM_MAX_STATES_DEV];       // elp[n-1][c] HBM -> chain
    __shared__ double sh_hm[BAND ? SN : 1][BAND ? 64 : 1];   // band mode: max h per group of 16 sources, ring of 64 groups
    // band mode: band 0's length ring as a TABLE (see DOM in the pusher waves): sh_l0[c][kr] = len[kr][c] for the ring
    // distances kr band 0 owns, -inf elsewhere (entry 128 = entry 0: the pair of a lane whose first slot is at distance 127),
    // and per state how far h must RISE from one source to the next for the older one to be beaten at every target
    __shared__ __attribute__((aligned(16))) double sh_l0[BAND ? SN : 1][BAND ? SMM_L0_ROW : 1];
    __shared__ double sh_xd[BAND ? SN : 1];
    __shared__ double sh_dlow[(BAND && SMM_ANCHOR) ? SN : 1][(BAND && SMM_ANCHOR) ? 64 : 1];   // ANCHOR (pusher waves): min of D_c over buckets of 16 distances
    __shared__ __attribute__((aligned(16))) double sh_gam[SMM_MAX_STATES_DEV];           // gamma[n][.] chain-private broadcast
    __shared__ double sh_gfin[SMM_MAX_STATES_DEV];                                        // gamma[T][.] for the closing step
    __shared__ __attribute__((aligned(16))) double sh_junk[2][B][SMM_MAX_STATES_DEV][2]; // where the chain wave's other lane groups store
    // speculative transition (see SPEC in the chain wave): the video's transition table and, per (leader cs, source c),
    // how far gamma[c] must lie below gamma[cs] for source c to lose against cs at EVERY target
    // (as small as the kernel's class sets allow)
    __shared__ double sh_tr[SMM_SPEC ? SN * SN : 1];                                     // [to][from]
    __shared__ double sh_dl[SMM_SPEC ? SN * SN : 1];                                     // [cs][c]
    __shared__ unsigned sh_kmin[3];
    __shared__ int sh_c;
    __shared__ int sh_guess[SMM_MAX_STATES_DEV + 1];   // back-trace: the predecessor state last seen / expected for each state

    if (T <= 0) return;
    if (spans)
        for (int i = threadIdx.x; i <= a.t_max; i += blockDim.x) spans[i] = -1;
    if (threadIdx.x < SMM_MAX_STATES_DEV) {
        const int c = threadIdx.x;
#pragma unroll
        for (int i = 0; i < B; ++i) {
            sh_apart[0][i][c] = SMM_NEG_INF;              // block 0 needs no pusher source
            sh_apart[1][i][c] = SMM_NEG_INF;
            sh_gh[0][i][c][0] = SMM_NEG_INF; sh_gh[0][i][c][1] = SMM_NEG_INF;
            sh_gh[1][i][c][0] = SMM_NEG_INF;
            sh_gh[1][i][c][1] = (i == B - 1 && c < C) ? (chunk_flat ? 0.0 : init[c]) : SMM_NEG_INF;   // "block -1": only position 0 exists
            sh_e[0][i][c] = (c < C && i < T) ? elp[(size_t)i * cm + c] : 0.0;    // block 0
            sh_e[1][i][c] = 0.0;   // (columns >= c_max are never written again; dead lanes of the chain wave read them)
        }
        sh_gam[c] = SMM_NEG_INF;
        sh_gfin[c] = SMM_NEG_INF;
        if (c < C) {                                                               // history of n = 0
            hcum[c] = chunk_flat ? anchor[c] : 0.0;
            hh[(size_t)c * (T + 1)] = chunk_flat ? 0.0 : init[c];
        }
    }
    constexpr bool SPEC = SMM_SPEC != 0;
    if constexpr (SPEC) {
        for (int e = threadIdx.x; e < SN * SN; e += blockDim.x) {
            const int to = e / SN, f = e % SN;
            sh_tr[e] = (to < C && f < C) ? trans[(size_t)to * cm + f] : SMM_NEG_INF;
        }
    }
    if constexpr (BAND) {
        constexpr int KMIN0 = TRI ? B + D + 1 : 2 * B + D;                 // band 0's shortest length (the chain wave owns the rest)
        const int khi = (kp - 1 < 127) ? kp - 1 : 127;
        for (int e = threadIdx.x; e < SN * SMM_L0_ROW; e += blockDim.x) {
            const int c = e / SMM_L0_ROW, kr = e % SMM_L0_ROW;
            sh_l0[c][kr] = (c < C && kr >= KMIN0 && kr <= khi) ? len[(size_t)kr * cm + c] : SMM_NEG_INF;
        }
        if constexpr (SMM_ANCHOR != 0) {
            const double *dl = a.dmin_t + (size_t)g * cm * 64;           // (smm_band_tables_kernel fills it in -DSMM_ANCHOR=1 builds only)
            for (int e = threadIdx.x; e < SN * 64; e += blockDim.x)
                sh_dlow[e / 64][e % 64] = (e / 64 < C) ? dl[(size_t)(e / 64) * 64 + e % 64] : SMM_NEG_INF;
        }
        // DOM (see the pusher waves): X_c = max over band 0's lengths of len[k][c] - len[k-1][c], clipped at 0 and pushed
        // a part in 2^49 up.  h[s+1][c] - h[s][c] > X_c (both differences rounded once: relative error 2^-53 each) then
        // implies h[s][c] + len[k][c] <= h[s+1][c] + len[k-1][c] in real arithmetic for every length of the band, hence
        // -- rounding is monotone -- for the rounded sums the maxima are made of.  A length the table does not reach
        // (-inf) asks for nothing; one whose predecessor is -inf can never be granted (+inf).
        for (int c = w; c < SN; c += NW) {
            double x = SMM_NEG_INF;
            if (c < C)
                for (int k = KMIN0 + lane; k <= khi; k += 64) {
                    const double la = len[(size_t)k * cm + c], lb = len[(size_t)(k - 1) * cm + c];
                    if (la != SMM_NEG_INF) x = fmax(x, la - lb);
                }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) x = fmax(x, __shfl_xor(x, off));
            if (lane == 0) sh_xd[c] = fmax(x, 0.0) * (1.0 + 0x1p-49);
        }
    }
    __syncthreads();
    if constexpr (SPEC) {
        // dl[cs][c] = min over targets of (trans[to][cs] - trans[to][c]), clipped at 0 and pushed one part in 2^50 further
        // down: gamma[c] - gamma[cs] <= dl[cs][c] (both sides rounded) then implies gamma[c] + trans[to][c] <=
        // gamma[cs] + trans[to][cs] in real arithmetic for every target, hence -- rounding is monotone -- for the rounded
        // sums the fold compares.  A target that source c cannot reach (-inf) asks for nothing; one that only cs cannot
        // reach can never be granted (-inf).
        for (int e = threadIdx.x; e < SN * SN; e += blockDim.x) {
            const int cs = e / SN, c = e % SN;
            double d = 0.0;
            if (cs < C && c < C) {
                for (int to = 0; to < C; ++to) {
                    const double x = sh_tr[to * SN + cs], y = sh_tr[to * SN + c];
                    if (y == SMM_NEG_INF) continue;
                    d = fmin(d, x - y);
                }
                d = d * (1.0 + 0x1p-50);
            }
            sh_dl[e] = d;
        }
    }
    __syncthreads();

#ifdef SMM_PROFILE
    const unsigned long long p_kern1 = __builtin_readcyclecounter();
#endif
    // The chain wave touches LDS only (a wave that waits for a load waits for its older stores as well: vmcnt is one
    // in-order counter).  HBM traffic is moved block-wise by pusher wave MW: it fetches the elp rows two blocks ahead
    // (registers for one block, then LDS) and stores cumE, h and gamma a block after the chain wave produced them.
    // A block is B*cm contiguous doubles in HBM and B rows of SMM_MAX_STATES_DEV in LDS.
    constexpr int NE = (B * SMM_MAX_STATES_DEV + 63) / 64;   // elements per lane
    double ninf = SMM_NEG_INF;                             // kept in a register pair (smm_ring_block)
    asm volatile("" : "+v"(ninf));
    const int J = (T + B - 1) / B;                         // blocks; one barrier each, in every wave
    // ---------------------------------------------------------------------------------------------- the mover's block step
    // Wave MW (the chain wave's SIMD partner) moves the HBM traffic, block-wise.  In block j it
    //   * reads the staged elp rows and the (gamma, h) rows of block j-1 from LDS: lane = (half, state), each half of the
    //     wave takes half of the block's positions (B/2 16-byte reads + B 8-byte reads per lane, conflict-free);
    //   * writes the elp rows of block j+1 (fetched a block ago) to LDS and fetches those of block j+2 -- unconditional
    //     loads from clamped addresses (rows >= T are never used): a predicated load waits for the previous one into the
    //     same register;
    //   * adds the elp rows up (cumE[n] = cumE[n-1] + elp[n-1]: the chain wave's own additions, repeated) and stores
    //     cumE, gamma (rows of C in HBM) and h (state-major in HBM: B/2 consecutive positions per lane);
    //   * BAND mode: returns max h over the block's sources per state (lane = state, both halves), for the skip test.
    constexpr int PB = B / 2;                              // positions per half of the mover wave
    int mv_lo[NE];                                         // elp block element e = lane + 64 q  ->  LDS offset (row e / cm, column e % cm), -1: none
    double mv_pre[NE], mv_cum = (chunk_flat && (lane & (SMM_MAX_STATES_DEV - 1)) < C) ? anchor[lane & (SMM_MAX_STATES_DEV - 1)] : 0.0;
    const int64_t e_last = (int64_t)T * cm - 1;
    if (w == MW) {
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int e = lane + 64 * q;
            mv_lo[q] = (e < B * cm) ? (e / cm) * SMM_MAX_STATES_DEV + e % cm : -1;
            const int64_t eg = (int64_t)B * cm + e;                                // block 1
            mv_pre[q] = elp[eg < e_last ? eg : e_last];
        }
    }
    auto mover_step = [&](int j, int jj, bool prefetch) -> double {
        const int mc = lane & (SMM_MAX_STATES_DEV - 1), mh = lane >> 5;
        double ev[B];
        double2 gh[PB];
#pragma unroll
        for (int i = 0; i < B; ++i) ev[i] = sh_e[(jj + 1) & 1][i][mc];
#pragma unroll
        for (int i = 0; i < PB; ++i) gh[i] = *reinterpret_cast<const double2 *>(&sh_gh[(jj + 1) & 1][mh * PB + i][mc][0]);
        __builtin_amdgcn_sched_barrier(0);
        if (prefetch) {
            double *dst = &sh_e[(jj + 1) & 1][0][0];
#pragma unroll
            for (int q = 0; q < NE; ++q)
                if (mv_lo[q] >= 0) dst[mv_lo[q]] = mv_pre[q];
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                const int64_t e = (int64_t)(j + 2) * B * cm + lane + 64 * q;
                mv_pre[q] = elp[e < e_last ? e : e_last];
            }
        }
        double cv[PB];
#pragma unroll
        for (int i = 0; i < B; ++i) {
            mv_cum = mv_cum + ev[i];
            if (i / PB == 0) { if (mh == 0) cv[i % PB] = mv_cum; }
            else { if (mh == 1) cv[i % PB] = mv_cum; }
        }
        double hmax = gh[0].y;
#pragma unroll
        for (int i = 1; i < PB; ++i) hmax = smm_fmax(hmax, gh[i].y);
        if (j >= 1 && mc < C) {
            const int n0 = (j - 1) * B + 1 + mh * PB;                               // this lane's positions n0 .. n0 + PB - 1
            double *hrow = hh + (size_t)mc * (T + 1) + n0;
            if (n0 + PB - 1 <= T) {
#pragma unroll
                for (int i = 0; i < PB; ++i) {
                    hcum[(size_t)(n0 + i) * C + mc] = cv[i];
                    hgam[(size_t)(n0 + i) * C + mc] = gh[i].x;
                    hrow[i] = gh[i].y;
                }
            } else {
#pragma unroll
                for (int i = 0; i < PB; ++i) {
                    if (n0 + i <= T) {
                        hcum[(size_t)(n0 + i) * C + mc] = cv[i];
                        hgam[(size_t)(n0 + i) * C + mc] = gh[i].x;
                        hrow[i] = gh[i].y;
                    }
                }
            }
        }
        return smm_max_halves(hmax);
    };
    if (w == 0) {
        // ============================================================================ chain wave
        // The serial chain is the critical path of a block; its SIMD partner is a pusher wave with an endless
        // supply of independent fp64 work, so the chain wave takes priority in the issue arbitration.
        __builtin_amdgcn_s_setprio(3);
        // The transition's fold is compiled for the launch's largest class set (HF source states per lane group) AND,
        // in BAND launches of more than 16 states, for <= 16 states (four lane groups of 16, four sources each): a
        // corpus mixes tasks of 11..23 states in one launch, and a video of 11 states need not fold 24 sources.
        auto chain = [&](auto hf_c) {
            constexpr int HFC = decltype(hf_c)::value;
            constexpr int LG = (HFC == 4) ? 16 : 32;           // lanes per group: two groups of 32, or (HFC = 4, <= 16 states) four of 16
            const int to = lane & (LG - 1), half = lane / LG;  // (`half`: the lane's group)
            const bool live = to < C;
            double tr[HFC];                                    // trans[to][half*HFC + i]
    #pragma unroll
            for (int i = 0; i < HFC; ++i) {
                const int f = half * HFC + i;
                tr[i] = (live && f < C) ? trans[(size_t)to * cm + f] : SMM_NEG_INF;
            }
            double lk[K0 + 1];                                // len[k][to], k = 1..K0
    #pragma unroll
            for (int k = 1; k <= K0; ++k) lk[k] = (live && k <= kp - 1) ? len[(size_t)k * cm + to] : SMM_NEG_INF;
            double hq[M];                                     // h[n][to], slot n mod M
    #pragma unroll
            for (int i = 0; i < M; ++i) hq[i] = SMM_NEG_INF;
            hq[0] = live ? (chunk_flat ? 0.0 : init[to]) : SMM_NEG_INF;
            double cum = (chunk_flat && live) ? anchor[to] : 0.0;
            // every lane group computes every position; only group 0's results are wanted in LDS.  The other groups store
            // to a junk array of the same shape instead of being masked off: no exec juggling on the serial path (masking
            // measured slower: round 4, +8 % of the wave's busy time)
            double *const st_gam = half ? &sh_junk[0][0][to][0] : &sh_gam[to];
            double *const st_gh = half ? &sh_junk[0][0][to][0] : &sh_gh[0][0][to][0];
            constexpr int UC = M / B;                         // blocks per unrolled chain iteration: UC*B % M == 0, UC even
            SMM_PROF_DECL;
            // SPEC: the transition SPECULATED on one source.  beta[to] = max_c (gamma[c] + trans[to][c]) is the largest piece of
            // a position (12 adds, 12 maxes, 6 LDS reads behind an LDS round trip at 17..24 states), and on real data its
            // result is decided by ONE source nearly all the time: the state cs the frames currently belong to leads every
            // other gamma by tens to hundreds of nats.  A FAST position reads gamma[cs] from lane cs (one v_readlane per
            // half: no LDS on the serial path) and checks in every source lane c that gamma[c] - gamma[cs] <= dl[cs][c]
            // (sh_dl, see the prologue: source c then loses against cs at every target, in the rounded sums too, so the
            // maximum IS the cs term, bit for bit): one compare and one scalar branch; if every lane agrees,
            // beta[to] = gamma[cs] + trans[to][cs], else the position takes the full fold below -- the check comes before
            // anything is used, nothing is ever undone.  Behind a block with a full position the leader is looked for again
            // (arg-max of the block's last gamma row + the same check); lattices on which it never holds -- masked transition tables (the
            // check must hold for EVERY target), flat emissions -- try again with exponential back-off (every 32nd block at
            // most) and otherwise run exactly as before.
#ifdef SMM_PROFILE
            unsigned long long p_fastn = 0;                   // (diagnostic: positions of workgroup 0 that took the fast transition -> slot 34: wave 0 has no blocks with j mod 4 = 2)
#endif
            bool spec = false;                                // the leader cs is believed to hold
            int cs = 0, spec_wait = 0, spec_back = 1;         // leader; blocks until the next try; back-off
            int sb_wait = 0, sb_back = 1;                     // SPEC8: blocks until the next block-level try; back-off
#ifdef SMM_PROFILE
            unsigned long long p_sb_ok = 0, p_sb_fail = 0;    // (diagnostic: blocks of workgroup 0 that the block-level speculation decided / had to replay)
#endif
            double trS = SMM_NEG_INF, dlt = SMM_NEG_INF;      // trans[to][cs], dl[cs][to] (-inf: no leader holds, every position takes the full fold)
            const bool spec_off = !SPEC || (a.flags & 256);   // (SMM_SPEC=0: A/B aid)
            if (spec_off) spec_wait = 0x7fffffff;
            __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the tables have arrived; the loop is LDS-only
            for (int j0 = 0; j0 < J; j0 += UC) {
    #pragma unroll
                for (int jj = 0; jj < UC; ++jj) {
                    const int j = j0 + jj;
                    if (j >= J) break;
                    // A'[n][to] and elp[n-1][to] of the block's positions: read TWO positions ahead of their use (all sixteen at the
                    // block's start cost 28 registers of a wave that has none to spare)
                    // ... in the kernels that are short of registers (TIGHT: blocks of 8 positions, two lane groups, the
                    // speculative transition compiled in); the others read the whole block at its start, one wait for all
                    constexpr bool TIGHT = SPEC && B == 8 && HFC != 4;
                    double ap[B], ev[B];
                    auto stage = [&](int i) {
                        if (i < B) { ap[i] = sh_apart[jj & 1][i][to]; ev[i] = sh_e[jj & 1][i][to]; }
                    };
                    // Software pipeline inside the block: everything of position n+1 that does not depend on h[n] -- the
                    // candidates k = 2..K0, A'[n+1], cumE[n+1] -- is evaluated in the shadow of position n's LDS round trip
                    // (gamma broadcast), so that the serial path of a position is add, max, add, LDS, transition, sub.
                    auto partial = [&](int i) {                  // max(A'[n], max_{k=2..K0} h[n-k] + len[k]), n = jB+1+i
                        double acc = ap[i];
                        if constexpr (SMM_ABLATE & 1) return acc;
                        const int kmax = TRI ? B + D + i : K0;     // (i is a constant at every call site)
                        // groups of four candidates, fenced: all fourteen sums ahead of the first max (what the scheduler
                        // makes of a free hand) are 28 live registers in a wave that spills past 256 -- and a kernel with
                        // a private segment pays for it at dispatch (round 4: +0.3 ms on a launch beside another kernel)
    #pragma unroll
                        for (int k0 = K0; k0 >= 2; k0 -= 4) {
                            double sq[4];
    #pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (k0 - q >= 2 && k0 - q <= kmax) sq[q] = hq[(jj * B + 1 + i - (k0 - q) + 4 * M) % M] + lk[k0 - q];
    #pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (k0 - q >= 2 && k0 - q <= kmax) acc = smm_fmax(acc, sq[q]);
                            if constexpr (TIGHT) { if (k0 - 4 >= 2) __builtin_amdgcn_sched_barrier(0); }
                        }
                        return acc;
                    };
                    // SPEC8: the whole BLOCK speculated at once (round 5).  A fast position still hangs on its predecessor: h[n-1] feeds
                    // the k = 1 candidate of gamma[n], gamma[n] of the leader lane feeds h[n], and between two positions sit a
                    // v_readlane pair and the check's compare -> ballot -> branch -- ~330 cycles per position for ~40 instructions.
                    // But while a leader cs holds, its own gamma hardly ever comes from a source inside the current block (a
                    // segment that long does not restart every few frames), and gamma[cs] is all that the OTHER lanes' h depend on.
                    // So: (1) everything that needs only sources of earlier blocks -- cumE of the 8 positions and, per position, the
                    // maximum pp[i] over A'[i] and the candidates k > i -- is computed for all 8 positions, no dependence between
                    // them; (2) GUESS gamma[i][cs] = cumE[i][cs] + pp[i][cs] (no source of this block wins for the leader) and read all
                    // 8 from lane cs; (3) h[i][.] = (gs[i] + trans[.][cs]) - cumE[i][.] for all 8 at once; (4) with those, the
                    // in-block candidates k <= i and the TRUE gamma[i][.] of every lane; (5) ONE check for the block: no lane, at no
                    // position, has gamma[i][c] - gs[i] > dl[cs][c].  In lane cs the threshold is 0 and gamma >= the guess by
                    // construction, so passing there means the guess WAS gamma[i][cs]; by induction over i every h used was the true
                    // one, the per-position checks of the serial code would all have passed, and what was stored is what the serial
                    // code stores, bit for bit (the same candidates in the same expressions; max is exact and order-free).  A block
                    // that fails -- a boundary inside it, a competitor within reach -- has changed nothing but its own LDS rows and
                    // is run again by the serial code below.  (Back-off as for the leader search: a lattice on which blocks keep
                    // failing tries every 8th block.)
                    bool block_done = false;
                    if constexpr (SPEC && SMM_SPEC8) {
                        if (spec && sb_wait == 0) {
                            double cumv[B], pp[B];
                            {
                                // (the block's A' and elp rows are asked for first and needed last: the candidates of the earlier
                                // blocks' sources come from registers and run while the LDS answers)
                                double apv[B], evv[B];
    #pragma unroll
                                for (int i = 0; i < B; ++i) { apv[i] = sh_apart[jj & 1][i][to]; evv[i] = sh_e[jj & 1][i][to]; }
    #pragma unroll
                                for (int i = 0; i < B; ++i) {
                                    double acc = SMM_NEG_INF;
                                    const int kmax = TRI ? B + D + i : K0;
    #pragma unroll
                                    for (int k0 = K0; k0 >= 1; k0 -= 4) {
                                        double sq[4];
    #pragma unroll
                                        for (int q = 0; q < 4; ++q)
                                            if (k0 - q > i && k0 - q <= kmax) sq[q] = hq[(jj * B + 1 + i - (k0 - q) + 4 * M) % M] + lk[k0 - q];
    #pragma unroll
                                        for (int q = 0; q < 4; ++q)
                                            if (k0 - q > i && k0 - q <= kmax) acc = smm_fmax(acc, sq[q]);
                                    }
                                    pp[i] = acc;
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                double cr = cum;
    #pragma unroll
                                for (int i = 0; i < B; ++i) {
                                    cr = cr + evv[i];
                                    cumv[i] = cr;
                                    pp[i] = smm_fmax(pp[i], apv[i]);
                                }
                            }
                            double gsv[B], hg[B];
    #pragma unroll
                            for (int i = 0; i < B; ++i) gsv[i] = smm_readlane(cumv[i] + pp[i], cs);
    #pragma unroll
                            for (int i = 0; i < B; ++i) hg[i] = (gsv[i] + trS) - cumv[i];
                            double viol = SMM_NEG_INF;
    #pragma unroll
                            for (int i = 0; i < B; ++i) {
                                double acc = pp[i];
                                const int kmax = TRI ? B + D + i : K0;
    #pragma unroll
                                for (int k = 1; k <= i; ++k)
                                    if (k <= kmax) acc = smm_fmax(acc, hg[i - k] + lk[k]);
                                const double gm = cumv[i] + acc;
                                viol = smm_fmax(viol, gm - gsv[i]);
                                *reinterpret_cast<double2 *>(st_gh + ((jj & 1) * B + i) * 2 * SMM_MAX_STATES_DEV) = make_double2(gm, hg[i]);
                            }
                            if (__ballot(viol > dlt) == 0) {
    #pragma unroll
                                for (int i = 0; i < B; ++i) hq[(jj * B + 1 + i) % M] = hg[i];
                                cum = cumv[B - 1];
                                block_done = true;
                                sb_back = 1;
#ifdef SMM_PROFILE
                                p_fastn += B; ++p_sb_ok;
#endif
                            } else {
                                sb_wait = sb_back;
                                sb_back = sb_back < 8 ? 2 * sb_back : 8;
#ifdef SMM_PROFILE
                                ++p_sb_fail;
#endif
                            }
                        } else if (sb_wait > 0) {
                            --sb_wait;
                        }
                    }
                    if (!block_done) {
                    if constexpr (TIGHT) { stage(0); stage(1); stage(2); }
                    else {
    #pragma unroll
                        for (int i = 0; i < B; ++i) stage(i);
                    }
                    double pacc = partial(0);
                    double cumn = (SMM_ABLATE & 2) ? ev[0] : cum + ev[0];
                    const bool spec0 = spec;
                    double gm_last = SMM_NEG_INF;                // gamma of the block's last position, if that one took the full fold
                    // the full transition of position i of the block: beta[to] = max_from (gamma[from] + trans[to][from]); this
                    // lane group folds sources half*HFC ..  (gamma comes back from LDS: the row the chain wave has just written)
                    auto fold_read = [&](int i, double2 (&gv)[HFC / 2]) {
                        (void)i;
                        const double2 *gp = reinterpret_cast<const double2 *>(&sh_gam[half * HFC]);
    #pragma unroll
                        for (int q = 0; q < HFC / 2; ++q) gv[q] = gp[(SMM_ABLATE & 16) ? q % 2 : q];
                    };
                    auto fold_reduce = [&](const double2 (&gv)[HFC / 2]) {
                        double bq[4];                                // 4 independent max chains
    #pragma unroll
                        for (int q = 0; q < ((SMM_ABLATE & 16) ? 2 : HFC / 2); ++q) {
                            if (q < 2) {
                                bq[2 * q] = gv[q].x + tr[2 * q];
                                bq[2 * q + 1] = gv[q].y + tr[2 * q + 1];
                            } else {
                                bq[(2 * q) & 3] = smm_fmax(bq[(2 * q) & 3], gv[q].x + tr[2 * q]);
                                bq[(2 * q + 1) & 3] = smm_fmax(bq[(2 * q + 1) & 3], gv[q].y + tr[2 * q + 1]);
                            }
                        }
                        double bm = smm_fmax(smm_fmax(bq[0], bq[1]), smm_fmax(bq[2], bq[3]));
                        if constexpr (HFC == 4) bm = smm_max_rows16(bm);
                        return smm_max_halves(bm);
                    };
                    // MODE 0: every position takes the full fold (no leader holds; kernels without SPEC): the next position's
                    // candidates are evaluated in the shadow of the fold's LDS round trip.  MODE 1: a leader holds at the
                    // block's start: the fast transition, checked per position, the full fold (unshadowed: rare) where it fails.
                    // PIPE0: a full position evaluates the next position's candidates in the shadow of its fold's LDS round trip.
                    // With the speculative transition compiled in, the kernels with blocks of 8 positions and two lane groups
                    // cannot afford the registers (fold operands + candidate sums at once: a few spilled registers, i.e. a
                    // private segment, i.e. a slower DISPATCH of every launch): there the candidates follow the fold.
                    constexpr bool PIPE0 = !TIGHT || SMM_PIPE0_ALWAYS;
                    auto run_block = [&](auto mode_c) {
                        constexpr int MODE = decltype(mode_c)::value;
    #pragma unroll
                        for (int i = 0; i < B; ++i) {
                            // Every position of a block is computed, also those past T in the tail of the last block (their
                            // rows are never stored, published or read): no bounds test on the serial path -- every instruction
                            // of this wave, scalar compare and branch included, is a slot of the position's time (same-box A/B:
                            // cfg2 DP -1.2 %).  Position n = jB + 1 + i; n mod M == (jj*B + 1 + i) mod M.
                            if constexpr (TIGHT) stage(i + 3);
                            const double acc = smm_fmax(pacc, hq[(jj * B + i + 4 * M) % M] + lk[1]);
                            cum = cumn;
                            const double gm = cum + acc;
                            // (gamma[T] is read back from the block's rows behind the loop -- no test per position)
                            {
                                double hcur;
                                if constexpr (MODE == 1) {
                                    // (while no leader holds -- after a failed check earlier in this block -- dlt is -inf and lane cs
                                    // itself objects)
                                    const double gs = (SMM_ABLATE & 512) ? dlt : smm_readlane(gm, cs);
                                    const bool bad = (SMM_ABLATE & 512) ? spec_wait == 12345 : __ballot(gm - gs > dlt) != 0;
                                    const double hfast = (gs + trS) - cum;
                                    __builtin_amdgcn_sched_barrier(0);
                                    if (i + 1 < B) {       // the next position's candidates: between the compare and the branch on it
                                        pacc = partial(i + 1 < B ? i + 1 : 0);
                                        cumn = cum + ev[i + 1 < B ? i + 1 : 0];
                                    }
                                    __builtin_amdgcn_sched_barrier(0);
                                    // (pins: the compiler would otherwise sink all of this behind the branch, which then waits for
                                    // its compare with nothing else to issue)
                                    asm volatile("" : "+v"(pacc), "+v"(cumn));
                                    double hfast_p = hfast;
                                    asm volatile("" : "+v"(hfast_p));
                                    if (__builtin_expect(bad, 0)) {
                                        st_gam[0] = gm;                          // (the fold reads gamma back from its broadcast row)
                                        // the full fold in pieces of two sources, fenced (rare: registers matter here, time does not --
                                        // a spill anywhere in the kernel is a private segment, and that costs every launch)
                                        const double2 *gp = reinterpret_cast<const double2 *>(&sh_gam[half * HFC]);
                                        double bm = SMM_NEG_INF;
    #pragma unroll
                                        for (int q = 0; q < HFC / 2; ++q) {
                                            const double2 g2 = gp[q];
                                            bm = smm_fmax(bm, smm_fmax(g2.x + tr[2 * q], g2.y + tr[2 * q + 1]));
                                            __builtin_amdgcn_sched_barrier(0);
                                        }
                                        if constexpr (HFC == 4) bm = smm_max_rows16(bm);
                                        hcur = smm_max_halves(bm) - cum;
                                        if (spec) { spec = false; dlt = SMM_NEG_INF; }   // the leader lost its lead: the next one is looked for behind this block
                                        if (i == B - 1) gm_last = gm;
                                    } else {
                                        hcur = hfast_p;
#ifdef SMM_PROFILE
                                        ++p_fastn;
#endif
                                    }
                                } else {
                                    double2 gv[HFC / 2];
                                    st_gam[0] = gm;
                                    fold_read(i, gv);
                                    __builtin_amdgcn_sched_barrier(0);
                                    if constexpr (PIPE0) {
                                        if (i + 1 < B) {
                                            pacc = partial(i + 1 < B ? i + 1 : 0);
                                            cumn = (SMM_ABLATE & 2) ? ev[i + 1 < B ? i + 1 : 0] : cum + ev[i + 1 < B ? i + 1 : 0];
                                        }
                                        __builtin_amdgcn_sched_barrier(0);
                                    }
                                    hcur = fold_reduce(gv) - cum;
                                    if constexpr (!PIPE0) {
                                        __builtin_amdgcn_sched_barrier(0);
                                        if (i + 1 < B) {
                                            pacc = partial(i + 1 < B ? i + 1 : 0);
                                            cumn = cum + ev[i + 1 < B ? i + 1 : 0];
                                        }
                                    }
                                    if (i == B - 1) gm_last = gm;
                                }
                                hq[(jj * B + 1 + i) % M] = hcur;
                                *reinterpret_cast<double2 *>(st_gh + ((jj & 1) * B + i) * 2 * SMM_MAX_STATES_DEV) = make_double2(gm, hcur);
                            }
                        }
                    };
                    if (SPEC && spec) run_block(std::integral_constant<int, SPEC ? 1 : 0>{});
                    else run_block(std::integral_constant<int, 0>{});
                    if constexpr (SPEC) {
                        if (spec0 && !spec) { spec_wait = 0; spec_back = 1; }    // a leader lost its lead in this block: try at once
                        if (!spec && --spec_wait < 0) {
                            // who leads the block's last gamma row (lane group 0; the other groups hold copies), and does every
                            // other source lose against it at every target?  (a fast last position: its leader held, and the
                            // flag is still up)
                            const double v = (live && half == 0) ? gm_last : SMM_NEG_INF;
                            const double rmax = smm_row_max16(v);
                            const double best = fmax(smm_readlane(rmax, 0), smm_readlane(rmax, 16));
                            const unsigned long long lead_m = __ballot(v == best && live && half == 0);
                            bool ok = false;
                            if (lead_m) {
                                cs = __builtin_amdgcn_readfirstlane(__ffsll(lead_m) - 1);
                                int to_i = to < SN ? to : 0;
                                asm volatile("" : "+v"(to_i));       // (addresses made here, once per segment: hoisted out of the loop they were spilled)
                                trS = sh_tr[to_i * SN + cs];
                                dlt = sh_dl[cs * SN + to_i];
                                ok = __ballot(gm_last - best > dlt) == 0;
                            }
                            if (ok) { spec = true; spec_back = 1; }
                            else { dlt = SMM_NEG_INF; spec_wait = spec_back; spec_back = spec_back < 32 ? 2 * spec_back : 32; }
                        }
                    }
                    }   // (!block_done)
                    SMM_LDS_BARRIER();                           // end of block j (LDS-only: see smm_lds_barrier)
                }
            }
            // gamma[T] for the closing step: the row of position T in the last block's staging rows (written by lane
            // group 0; every wave reads sh_gfin behind the __syncthreads() that follows the loops)
            if (half == 0) sh_gfin[to] = sh_gh[(J - 1) & 1][(T - 1) % B][to][0];
#ifdef SMM_PROFILE
            p_ph[2] = p_fastn;                                // (slot 34 of the stamps: wave 0 has no blocks with j mod 4 = 2)
            p_ph[1] = p_sb_ok; p_ph[3] = p_sb_fail;           // (slots 33 / 35: blocks the block-level speculation decided / replayed)
#endif
            SMM_PROF_OUT();
        };
        if constexpr (BAND && HF > 4 && SMM_CHAIN_DUAL) {
            if (C <= 16) chain(std::integral_constant<int, 4>{});
            else chain(std::integral_constant<int, HF>{});
        } else {
            chain(std::integral_constant<int, HF>{});
        }
    } else if (BAND) {
      if constexpr (BAND) {
        // ============================================================================ pusher waves, BAND mode (see above)
        constexpr int SPS = SPW, RS = 2, NBD = SMM_BAND_N;
        static_assert(!BAND || SPS * NBD <= 64, "one lane per (state, band) of a pusher wave");
        // Wave 4 shares its SIMD with the chain wave: it only moves data (elp in, history out) and owns NO states -- every
        // instruction it issues can cost the chain wave an issue slot, and the chain wave's time is the frame time.  The
        // other six pushers own states rank, rank + 6, ...; waves (1,5), (2,6), (3,7) share a SIMD.
        constexpr int NPS = NP - 1;
        const int rank = (w == MW) ? (1 << 20) : ((w < MW) ? w - 1 : w - 2);  // (the mover: no state passes the c < C tests below)
        const double *lent = a.len_t + (size_t)g * cm * SMM_BAND_ROW;       // [c][k + 1]: a state's lengths are contiguous
        const double *btab = a.band_tab + (size_t)g * cm * SMM_BAND_TAB;
        const bool bound_ok = kp - 1 >= 32 + 15 + 127;                       // the lower bound's witness needs lengths up to 174
        double As[SPS][RS];
        // The rows a wave pushes (its states' h of the block, B x SPS values) come from LDS with ONE read per 16 values:
        // lane e % 16 of every row of register e / 16 holds element e = B js + i, and a push broadcasts its source along
        // the rows (smm_row_bcast).  Every lane reading every value (a ds_read2_b64 per state and pair of rows) was 16
        // times the LDS traffic, in one burst of all pusher waves right behind the barrier, in front of the chain wave's
        // own reads.
        constexpr int NHR = (SPS * B + 15) / 16;
        // A group (the unit of the skip test) is 16 sources = BPG blocks; block j pushes the sources (j-1)B + 1 - D ..
        // jB - D, so group g = sources 16g + 1 - D .. 16g + 16 - D is pushed in blocks g BPG + 1 .. (g+1) BPG, and the ring
        // indices 1 .. 2B - 1 + D (= K0) of a push are slots that were handed over already (K0 < SMM_BAND_LO = 16).
        constexpr int BPG = 16 / B;
        constexpr int UBB = UB;                                              // blocks per iteration of the block loop (one group)
        static_assert(!BAND || (UB == BPG && K0 < SMM_BAND_LO), "a group is one unrolled iteration of the block loop");
        // DOM: SOURCE DOMINANCE in band 0.  Candidate (s, k) and candidate (s + 1, k - 1) aim at the same target, so the
        // source s need not be pushed for state c when h[s+1][c] - h[s][c] > X_c (sh_xd, see the prologue): its successor
        // beats it at every target band 0 reaches -- and the successor's candidate IS evaluated, by a pusher or, at
        // k - 1 = B, by the chain wave; the relation is transitive along the block, whose last source is always pushed.
        // max is exact and the test is one-sided, so not a bit of A changes; what changes is the work: h = gamma - cumE of
        // a state that does not explain the current frames RISES by the margin of the state that does, frame after frame
        // (tens of nats on real features, X_c is a few), so on CrossTask-shaped data all but the leading state push ONE
        // source per block instead of eight (scripts/probe_dominance.py: 1.65 of 8 on cfg3; flat lattices: all of them,
        // as before).  The price: a state's length ring can no longer live in rotating registers (a skipped push would
        // still have to rotate it); it is read from LDS at the phase of every push that happens (sh_l0: 16 bytes per lane).
        static_assert(!BAND || D == 0, "DOM: the block's own sources, all of them in the rows read at its start");
        int hoff[NHR];
        double xdom[NHR];                                                    // lane = element (state, source): the state's X_c
#pragma unroll
        for (int r = 0; r < NHR; ++r) {
            const int e = 16 * r + (lane & 15), ejs = e / B, ec = ejs * NPS + rank;
            hoff[r] = ((e % B) * SMM_MAX_STATES_DEV + ((ejs < SPS && ec < C) ? ec : 0)) * 2 + 1;   // (the h half of the (gamma, h) pairs)
            xdom[r] = sh_xd[BAND ? ((ejs < SPS && ec < C) ? ec : 0) : 0];
        }
        // ANCHOR (round 5): dominance by an OLDER source.  DOM leaves a source out when its SUCCESSOR beats it at every target;
        // the state that currently explains the frames never passes that test (inside its segment h moves by len[m+1] - len[m],
        // a fraction of a nat), so its wave pushed all 8 sources of every block -- ~60 instructions and one or two dependent LDS
        // round trips, and that wave is the one the block barrier waits for.  But those sources are hopeless against the source
        // the segment STARTED from: candidate (s, k) and candidate (a, k + s - a) aim at the same target, and
        //     h[s][c] - h[a][c]  <  D_c[s - a]  =  (1 -+ 2^-49) min_{band 0's k} (len[k + s - a][c] - len[k][c])
        // (both sides rounded once; the deflated threshold makes it hold in real arithmetic for every length of the band, STRICTLY)
        // says that the older source a beats s at every target band 0 reaches -- a candidate that is strictly below another
        // candidate of its own target attains no maximum, whatever becomes of the other one (it may sit in a delayed band that
        // is skipped, or be left out by DOM: each of those is below yet another candidate; strict edges go to older sources,
        // the non-strict ones of DOM and of the band skip test to younger ones, every edge is an inequality of real numbers, so
        // a cycle would need x < x, and a chain of edges ends at a candidate that IS evaluated).  Inside a segment h[s] = h[a] +
        // len[s - a] + trans[c][c] for the segment's start a: a restart costs a self transition and a whole Poisson normaliser,
        // hundreds of nats below D.  The test lives in the RARE path only (a state with sources that DOM could not leave out):
        // per state of the wave one anchor (position, h) in wave-uniform registers, moved to the FIRST source that survives both
        // tests whenever there is one -- a state that becomes the leader drags a stale anchor along, fails the test once at its
        // segment's start, and has its anchor there from then on.  D_c is used through its minima over buckets of 16 distances
        // (sh_dlow, 64 per state, from smm_band_tables_kernel: a lower bound of D_c[d] is as good, and fits the LDS).  CPU probe
        // first (oracle/prune_probe.c: smm_anchor_probe): sources pushed per block by the leading state 7.9 -> 1.8 on cfg3.
        constexpr bool ANCHOR = SMM_ANCHOR != 0;
        static_assert(!BAND || !ANCHOR || (TRI ? B + D + 1 : 2 * B + D) >= 9, "dlow covers band 0's lengths from 9 on (smm_band_tables_kernel)");
        double anc_h[SPS];
        int anc_s[SPS];
#pragma unroll
        for (int js = 0; js < SPS; ++js) {
            const int ec = js * NPS + rank;
            anc_h[js] = (BAND && ec < C) ? init[ec] : SMM_NEG_INF;           // position 0: the start of every first segment
            anc_s[js] = 0;
        }
        const int dlim = ANCHOR ? kp - 1 - 127 : 0;                          // largest distance s - a whose candidates (a, k + s - a) all exist
        uint32_t nanch = 0;                                                  // (diagnostic: sources the anchor test left out)
        uint32_t npush = 0;                                                  // (diagnostic: sources this wave pushed into band 0)
        unsigned long long mcur = 0, mnext = 0;          // bit 8 js + m - 1: band m of the wave's js-th state is switched on for the
                                                         // current group / for the next one, once decided
        const int nvw = (rank < C) ? ((C - 1 - rank) / NPS + 1 < SPS ? (C - 1 - rank) / NPS + 1 : SPS) : 0;   // states of this wave
        uint32_t nact = 0;                                                   // (diagnostic: delayed band-blocks this wave pushed)
#pragma unroll
        for (int js = 0; js < SPS; ++js) {
#pragma unroll
            for (int r = 0; r < RS; ++r) As[js][r] = SMM_NEG_INF;
        }
        // The skip test.  The mover wave keeps max h over every group of 16 sources, one lane per state (ring sh_hm);
        // every pusher decides the bands of its own states, lane q = 8 js + (m - 1) for (state js, band m), in the block
        // BEFORE the last one of the previous group -- early enough for the delayed sources to be fetched two blocks ahead
        // of their pushes.  The witness of the lower bound is the group before the last one (G - 2: complete when the
        // decision is due), so the test reads
        //     hm[G - 7m] + max_{band m} len  >  hm[G - 2] + min_{33 <= k <= 174} len      <=>  band m on for group G.
        static_assert(!BAND || BPG == 2, "a group is two blocks: decided in the second block of the previous group");
        // the delayed sources are read from the history TWO blocks ahead of their pushes; the mover stores a block's rows
        // one step after the block and is known to have drained them two steps later (smm_lds_barrier): the newest row a
        // delayed band reads must be older than that by a margin
        static_assert(!BAND || (SMM_BAND_DELAY - 2 * B) / B - 1 >= 4 + 4, "delayed-band reads of the history: at least 4 block steps of slack behind the mover's drained stores");
        double hmx = SMM_NEG_INF;                                            // mover, lane = state: running max of the group
        // lane q = 8 js + (m - 1) decides (state js, band m) and fetches its delayed sources, two blocks ahead
        const int qjs = lane >> 3, qm = (lane & 7) + 1;
        const int qc = qjs * NPS + rank;
        const bool qok = qjs < SPS && qc < C && SMM_BAND_LO + SMM_BAND_DELAY * qm <= kp - 1;
        const double *qcol = hh + (size_t)((qjs < SPS && qc < C) ? qc : 0) * (T + 1);
        const double qlmx = qok ? btab[(size_t)qc * SMM_BAND_TAB + qm] : SMM_NEG_INF;
        const double qlbm = (qok && bound_ok) ? btab[(size_t)qc * SMM_BAND_TAB] : SMM_NEG_INF;
        const double *qhm = &sh_hm[BAND ? ((qjs < SPS && qc < C) ? qc : 0) : 0][0];
        // Round 4: EVERY complete group as a witness.  The best source s* of group G - delta (delta = 2 .. 55) is a real
        // candidate of every target the band-groups of group G reach, with 16 delta + 1 <= n - s* <= 16 delta + 142: so
        // hm[G - delta] + min len over that window is a lower bound of the final A[n] as well, and a band-group whose upper
        // bound is STRICTLY below any of them can be left out -- a candidate that is strictly below another candidate of
        // its own target attains no maximum, whatever becomes of the other one (no induction over witnesses needed; the
        // round-3 test with its one witness, group G - 2, skips ties too and argues through the witness's witness).  What
        // it buys: inside a long segment the sources behind the segment's START are beaten by the start itself (a restart
        // costs a self transition and a length of 1), so only the band that currently reaches back to the start has to
        // stay on, not every band up to it: 4.2 x fewer delayed band-groups on cfg3 (scripts/probe_dominance.py).  The
        // eight lanes of a state share the 54 windows: lane (js, m) looks at delta = 7 (m - 1) + 2 .. 7 (m - 1) + 8.
        constexpr int NWIT = 7;
        double qmlw[NWIT];
#pragma unroll
        for (int t = 0; t < NWIT; ++t) {
            const int dlt = 7 * (qm - 1) + 2 + t;
            qmlw[t] = (qjs < SPS && qc < C && dlt <= SMM_BAND_WIT && 16 * dlt + 142 <= kp - 1) ? btab[(size_t)qc * SMM_BAND_TAB + 16 + dlt] : SMM_NEG_INF;
        }
        double hq[2][B];                                                     // by parity of the block that pushes them
#pragma unroll
        for (int i = 0; i < B; ++i) { hq[0][i] = SMM_NEG_INF; hq[1][i] = SMM_NEG_INF; }
        constexpr int NPRE = 2;                   // rings fetched a block ahead: the wave's first NPRE (state, band) pairs
                                                  // (six measured 7 % SLOWER than two: registers and code for a case that is rare)
        double Lp[NPRE][RS];                      // in ascending 8 js + (m - 1), the order the push loop meets them in
#pragma unroll
        for (int k = 0; k < NPRE; ++k) { Lp[k][0] = SMM_NEG_INF; Lp[k][1] = SMM_NEG_INF; }
        int npre = 0, pk = 0;                     // pairs fetched for this block; pairs met so far in this block
        __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0): tables and rings have arrived
        SMM_PROF_DECL;
        for (int j0 = 0; j0 < J; j0 += UBB) {
#pragma unroll
            for (int jj = 0; jj < UBB; ++jj) {
                const int j = j0 + jj;
                if (j >= J) break;
                const int ph = jj % BPG;                                   // block of its group
                if (w == MW) {
                    // (the mover's block step: see mover_step) ... and the maxima of the skip test: the sources the pushers
                    // push in this block (positions (j-1)B + 1 .. jB), lane = state
                    const double gm = mover_step(j, jj, true);
                    if constexpr (!(SMM_ABLATE & 8)) {
                        hmx = smm_fmax(hmx, gm);
                        if (ph == 0) {                                         // group j/BPG - 1 is complete
                            // (j = 0: "group -1", slot 63: position 0, the start of every first segment, is its one real source)
                            if (lane < SN) sh_hm[BAND ? lane : 0][BAND ? ((j / BPG - 1) & 63) : 0] = hmx;
                            hmx = SMM_NEG_INF;
                        }
                    }
                }
                if (w != MW) {           // (the mover owns no states: none of what follows -- least of all the eight loads -- is for it)
                // this block pushes sources (j-1)B .. jB-1 (push steps jB ..): band 0 from LDS, the bands that are switched
                // on from the rows fetched during the previous block.  (Reading the rows of all states first and
                // interleaving the states' pushes step by step measured 3 % SLOWER: the wave is not bound by the latency
                // of one state's chain.)
                // All LDS reads of the block are issued up front: the rows (above) and -- first block of a group -- the
                // group's band switches, written by the mover wave during the previous group.
                double hvl[NHR];
                double dsrc = SMM_NEG_INF, dwit = SMM_NEG_INF;               // decision of group (j+1)/2, due in its block 2G - 1
                const int dG = (j + 1) / BPG;
#pragma unroll
                for (int r = 0; r < NHR; ++r) hvl[r] = (&sh_gh[(jj + 1) & 1][0][0][0])[hoff[r]];
                auto src_row = [&](int js, int i) {          // h[(j-1)B + 1 + i] of the wave's js-th state
                    return smm_row_bcast(hvl[(B * js + i) / 16], (B * js + i) % 16);
                };
                if (ph == 1 && !(SMM_ABLATE & 32)) {
                    dsrc = qhm[(dG - 7 * qm) & 63];
                    dwit = qhm[(dG - 2) & 63];
                }
                // the ring distance of this lane's first slot at the push of the block's LAST source (i = B - 1); source i: + B - 1 - i
                const int kr_last = (2 * lane + 1 + D - j * B) & 127;
                // ... and the ring pairs of those pushes -- every state pushes its last source --, read with the rows
                double2 lpl[SPS];
#pragma unroll
                for (int js = 0; js < SPS; ++js) {
                    const double *l0row = &sh_l0[BAND ? ((js < nvw) ? js * NPS + rank : 0) : 0][0];
                    lpl[js] = make_double2(l0row[kr_last], l0row[kr_last + 1]);
                }
                // DOM: which sources have to be pushed -- bit e = 8 js + i of keep[e / 16]: source i of the wave's js-th state is
                // not known to be beaten by its successor (lane e: h of the successor from the lane above, one compare)
                uint32_t keep[NHR], extra = 0;
#pragma unroll
                for (int r = 0; r < NHR; ++r) {
                    const double nx = smm_dpp<SMM_DPP_ROW_SHL(1)>(hvl[r]);         // (the row's last lane keeps its own: a last source)
                    const unsigned long long dom = (SMM_ABLATE & 1024) ? 0ull : __ballot(nx - hvl[r] > xdom[r]);
                    keep[r] = (uint32_t)~dom & 0x7f7fu;                            // (the last sources: pushed anyway, below)
                    extra |= keep[r];
                }
                if constexpr (!(SMM_ABLATE & 4)) {
                    // the last source of every state (straight-line; a wave with fewer states pushes into accumulators nobody reads)
#pragma unroll
                    for (int js = 0; js < SPS; ++js) {
                        const double hs = src_row(js, B - 1);
                        As[js][0] = smm_fmax(As[js][0], hs + lpl[js].x);
                        As[js][1] = smm_fmax(As[js][1], hs + lpl[js].y);
                    }
                    npush += nvw;
                    // the others that are not beaten: none for most states in most blocks
                    if (__builtin_expect(extra != 0, 0)) {
#pragma unroll
                        for (int js = 0; js < SPS; ++js) {
                            if (js >= nvw) break;
                            uint32_t k7 = (keep[(B * js) / 16] >> ((B * js) % 16)) & 0x7fu;
                            if (k7 == 0) continue;
                            if (ANCHOR && dlim > 0) {
                                // ANCHOR (above): the sources DOM could not leave out, against the state's anchor.  Lane (lane & 15) in
                                // [8 (js & 1), 8 (js & 1) + 8) of every row holds source i = lane & 7 of this state (block j pushes the
                                // sources (j-1)B + 1 .. jB); the other half-rows hold the neighbour state's and are masked off
                                const int dd = (j - 1) * B + 1 + (lane & 7) - anc_s[js];
                                const double dl = sh_dlow[BAND ? js * NPS + rank : 0][(dd >> 4) & 63];
                                const bool ad = dd >= 1 && dd <= dlim && hvl[(B * js) / 16] - anc_h[js] < dl;
                                const uint32_t a8 = (uint32_t)(__ballot(ad) >> ((B * js) % 16)) & 0x7fu;
                                nanch += __builtin_popcount(k7 & a8);
                                k7 &= ~a8;
                                if (k7 == 0) continue;
                                // the FIRST survivor is the state's next anchor: at the start of a segment the survivors are the start
                                // itself and the sources behind it, which the start beats from the next block on (an anchor INSIDE
                                // the segment would not: against it the self transition and the normaliser cancel)
                                const int il = __builtin_ctz(k7);
                                anc_s[js] = (j - 1) * B + 1 + il;
                                anc_h[js] = smm_readlane(hvl[(B * js) / 16], (B * js) % 16 + il);
                            }
                            const double *l0row = &sh_l0[BAND ? js * NPS + rank : 0][0];
                            if (__builtin_popcount(k7) < SMM_DOM_SPARSE) {
                                // a few: ring pair from the table at the push's phase
                                uint32_t rest = k7;
                                npush += __builtin_popcount(rest);
                                while (rest) {
                                    const int i = __builtin_ctz(rest);
                                    rest &= rest - 1;
                                    const int kr = (kr_last + B - 1 - i) & 127;
                                    const double2 lp = make_double2(l0row[kr], l0row[kr + 1]);
                                    const double hs = smm_readlane(hvl[(B * js) / 16], (B * js) % 16 + i);
                                    As[js][0] = smm_fmax(As[js][0], hs + lp.x);
                                    As[js][1] = smm_fmax(As[js][1], hs + lp.y);
                                }
                            } else {
                                // most of them (the state that leads; any state of a flat lattice): the ring at the phase of the
                                // block's first push, rotated in registers from source to source (smm_push), the sources before
                                // the last one pushed -- one that is beaten is still a candidate like any other
                                double Lr[RS];
                                const int kr0 = (kr_last + B - 1) & 127;
                                Lr[0] = l0row[kr0];
                                Lr[1] = l0row[kr0 + 1];
                                npush += B - 1;
#pragma unroll
                                for (int i = 0; i < B - 1; ++i)
                                    smm_push<RS>(As[js], Lr, src_row(js, i), (jj * B + i) % RS);
                            }
                        }
                    }
                }
                if (__builtin_expect(mcur != 0, 0)) {
                    // The bands that are switched on (rarely any, and then nearly always bands of ONE state: the one whose long
                    // segment is running).  ONE loop over the wave's (state, band) pairs q = 8 js + m - 1, ascending -- not a
                    // copy of the push code per state: a pair pushes its B sources into a fresh accumulator pair, which is then
                    // folded into the state's own (max is associative and exact: the same bits).  Round 3 had the loop once per
                    // state inside the unrolled block: four copies whose scalar registers the compiler spilled to lanes -- 130
                    // instructions per pair for 48 that push, and the wave that owns the running state was the one the block
                    // barrier waited for in most blocks of a long video (profiles/round4_*).
                    // Rings at this block's phase: the first NPRE pairs were fetched during the previous block (below), like the
                    // sources: a table row costs an L2 round trip.
                    const int off = (B + D - j * B) & 127;
                    unsigned long long rest = mcur;
                    nact += __builtin_popcountll(rest);
                    do {
                        const int q = __builtin_ctzll(rest);
                        rest &= rest - 1;
                        const int bjs = q >> 3, m = (q & 7) + 1;
                        double Lm[RS];
                        if (pk < npre) {                                 // (wave-uniform: scalar branches)
#pragma unroll
                            for (int k = 0; k < NPRE; ++k)
                                if (pk == k) { Lm[0] = Lp[k][0]; Lm[1] = Lp[k][1]; }
                        } else {
                            smm_band_ring_load(Lm, lent + (size_t)(bjs * NPS + rank) * SMM_BAND_ROW, off, m, kp, lane);
                        }
                        ++pk;
                        double At[RS];
                        At[0] = ninf; At[1] = ninf;
#pragma unroll
                        for (int i = 0; i < B; ++i)
                            smm_push<RS>(At, Lm, smm_readlane(hq[ph][i], q), (jj * B + i) % RS);
#pragma unroll
                        for (int js = 0; js < SPS; ++js) {
                            if (bjs == js) {
                                asm volatile("");   // (a real branch: not SPS x 4 conditional moves that every pair executes)
                                As[js][0] = smm_fmax(As[js][0], At[0]);
                                As[js][1] = smm_fmax(As[js][1], At[1]);
                            }
                        }
                    } while (rest);
                }
                // hand A' of block j+1 to the chain wave and clear those slots (the B slots are all registers of B/RS lanes) --
                // for all of the wave's states under ONE exec mask (a masked region per state was two branches per state)
                {
                    const int d = lane - (((j + 1) * B) & 127) / RS;
                    if (d >= 0 && d < B / RS) {
#pragma unroll
                        for (int js = 0; js < SPS; ++js) {
                            if (js >= nvw) break;
                            double *a_blk = &sh_apart[(jj + 1) & 1][0][js * NPS + rank];
#pragma unroll
                            for (int r = 0; r < RS; ++r) {
                                a_blk[(d * RS + r) * SMM_MAX_STATES_DEV] = As[js][r];
                                As[js][r] = ninf;
                            }
                        }
                    }
                    if constexpr (TRI) {
                        // the slots handed over a block ago (the B/RS lanes before these, around the ring): see TRI above
                        const int dp = (d + 64) & 63;
                        if (dp >= 64 - B / RS) {
#pragma unroll
                            for (int js = 0; js < SPS; ++js) {
#pragma unroll
                                for (int r = 0; r < RS; ++r) As[js][r] = ninf;
                            }
                        }
                    }
                }
                if (ph == 1 && !(SMM_ABLATE & 32)) {
                    // group dG = (j+1)/2 (pushed in blocks j+2, j+3): which of this wave's (state, band) pairs are on
                    bool on = qok && dG - 7 * qm >= -1 && dsrc + qlmx > dwit + qlbm;
                    if (__ballot(on) != 0 && SMM_BAND_ALLWIT) {
                        // a band would be switched on: look at the older (and the nearer) witnesses as well (see qmlw)
                        double lb = SMM_NEG_INF;
#pragma unroll
                        for (int t = 0; t < NWIT; ++t) {
                            const int gw = dG - (7 * (qm - 1) + 2 + t);
                            const double hw = qhm[gw & 63];
                            if (gw >= -1) lb = smm_fmax(lb, hw + qmlw[t]);      // (groups before the video: the ring holds nothing)
                        }
                        // ... the best of the state's eight lanes, in every one of them
                        lb = smm_fmax(lb, smm_dpp<0xB1>(lb));     // quad_perm [1, 0, 3, 2]
                        lb = smm_fmax(lb, smm_dpp<0x4E>(lb));     // quad_perm [2, 3, 0, 1]
                        lb = smm_fmax(lb, smm_dpp<0x141>(lb));    // row_half_mirror
                        on = on && !(dsrc + qlmx < lb);
                    }
                    mnext = __ballot(on);                         // (bit 8 js + m - 1: lane q = 8 js + (m - 1) decides (state js, band m))
                }
                const unsigned long long lanes2 = mnext;          // group (j+1)/2 (decided in either block of a group)
                if (ph == 0) mcur = mnext;                        // the next block starts group j/2
                // the rings of the next block's first NPRE (state, band) pairs at that block's phase (the tables stay in L2)
                {
                    const unsigned long long lanes = mcur;
                    npre = 0; pk = 0;
                    if (lanes) {                                  // (nearly always nothing is switched on)
                        const int offn = (B + D - (j + 1) * B) & 127;
                        unsigned long long rest = lanes;
#pragma unroll
                        for (int k = 0; k < NPRE; ++k) {
                            if (rest && !(SMM_ABLATE & 128)) {
                                const int pq = __builtin_ctzll(rest);
                                smm_band_ring_load(Lp[k], lent + (size_t)((pq >> 3) * NPS + rank) * SMM_BAND_ROW, offn, (pq & 7) + 1, kp, lane);
                                rest &= rest - 1;
                                npre = k + 1;
                            }
                        }
                    }
                }
                // The delayed sources of block j+2 (positions (j+1)B + 1 - 112m .. + B-1 of (state, band) = lane), TWO blocks
                // ahead: they come from HBM or the far cache, and a fetch that is not back when its block starts makes
                // every wave wait (the chain wave waited 700 cycles per block for the pushers, 280 without these loads).
                // Issued in EVERY block, after the ring loads, whether a band is on or not: the wave's loads complete in
                // order and a wait can only name a count, so the wait for the next block's rings must know that exactly
                // these B loads are younger -- behind a branch the compiler has to wait for everything.
                if (!(SMM_ABLATE & 64) && (SMM_HQ_ALWAYS || lanes2 != 0)) {
                    const bool mine = (lanes2 >> lane) & 1ull;
                    const int s0 = (j + 1) * B + 1 - SMM_BAND_DELAY * qm;
                    // (s0 in 1-B .. -1: the words before the row are read and masked below -- a video with a band switched
                    // on has >= 128 positions of history in front of that row.  Lanes with nothing to fetch read the head of
                    // the length table: always there, whatever the video's size)
                    const double *src = (mine && s0 > -B) ? qcol + s0 : lent;
#pragma unroll
                    for (int i = 0; i < B; ++i) hq[ph][i] = smm_ld_agent(src + i);
                    if ((j + 1) * B + 1 < SMM_BAND_DELAY * SMM_BAND_N) {               // (positions before 0 do not exist)
#pragma unroll
                        for (int i = 0; i < B; ++i)
                            if (s0 + i < 0) hq[ph][i] = SMM_NEG_INF;
                    }
                }
                }
                SMM_LDS_BARRIER();                               // end of block j
            }
        }
        SMM_PROF_OUT();
        if (w == MW) (void)mover_step(J, J & 1, false);                       // the last block's history
        if (lane == 0 && nact) atomicAdd(a.err + 3, (int)nact);             // error block word 3: see ops.error_words
        if (lane == 0 && npush) atomicAdd(a.err + 2, (int)npush);           // ... and word 2: sources pushed into band 0
        (void)nanch;
#ifdef SMM_PROFILE
        if (lane == 0 && blockIdx.x == 0 && w != MW && w < 7) reinterpret_cast<unsigned long long *>(a.err)[w < MW ? w + 1 : w] = nact;   // slots 2..6: workgroup 0's waves 1, 2, 3, 5, 6
#endif
      }
    } else {
        // ============================================================================ pusher waves
        // pusher rank: the wave that shares a SIMD with the chain wave (wave 4 when there are 8) goes last, so that it
        // owns the fewest states
        int rank = w - 1;
        // 8 waves, and the states fit six pushers at the same states-per-wave: wave 4 -- the chain wave's SIMD partner, at
        // the lower priority -- owns NO states and only moves data (as in BAND mode).  With two states it was the wave the
        // block barrier waited for (stamps on cfg2's shape: busy 3720 cycles per block of 8, the chain wave 3470, the
        // other pushers <= 2600).
        const bool stateless = NW == 8 && C <= (NP - 1) * SPW && !(SMM_MOVER_STATES);
        const int NPd = stateless ? NP - 1 : NP;                           // state-owning pushers
        if (NW == 8) {
            if (stateless) {
                rank = (w == MW) ? (1 << 20) : ((w < MW) ? w - 1 : w - 2);
            } else {
                rank = (w == 4) ? NP - 1 : ((w == NW - 1) ? 3 : w - 1);
                // ... unless that leaves one SIMD with two of the fuller waves (smm_device.h: smm_rebalanced_rank)
                const int swap = smm_rebalanced_rank(C);
                if (swap >= 0) rank = (rank == swap) ? NP - 1 : ((rank == NP - 1) ? swap : rank);
            }
        }
        const int nv_all = (C - rank + NPd - 1) / NPd;                     // states rank, rank+NPd, ...
        const int nv = nv_all < 0 ? 0 : (nv_all > SPW ? SPW : nv_all);
        double A[SPW][R], L[SPW][R], hd[SPW];
#pragma unroll
        for (int js = 0; js < SPW; ++js) {
            smm_ring_init<R, B, D, TRI>(A[js], L[js], len + (js < nv ? js * NPd + rank : 0), cm, kp, js < nv, lane);
            hd[js] = SMM_NEG_INF;
        }
        // Everything loaded so far (tables, rings) has to have arrived before the loop: the compiler's wait-count
        // bookkeeping would otherwise carry "maybe pending" into every iteration and make the waves that store the
        // history wait for their own stores.
        __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0)
        SMM_PROF_DECL;
        for (int j0 = 0; j0 < J; j0 += UB) {
#pragma unroll
            for (int jj = 0; jj < UB; ++jj) {
                const int j = j0 + jj;
                if (j >= J) break;
                if (w == MW) (void)mover_step(j, jj, true);                 // (see mover_step)
#pragma unroll
                for (int js = 0; js < SPW; ++js) {
                    if (js >= nv) break;
                    const int c = js * NPd + rank;
                    smm_ring_block<R, B, D, TRI>(A[js], L[js], hd[js], &sh_gh[(jj + 1) & 1][0][c][1], &sh_apart[(jj + 1) & 1][0][c], j, jj, lane, ninf);
                }
                SMM_LDS_BARRIER();                               // end of block j
            }
        }
        SMM_PROF_OUT();
        if (w == MW) (void)mover_step(J, J & 1, false);                       // the last block's history
    }

    if (chunk) return;                        // a unit of a time-split video ends with its history (every wave gets here: no barrier follows)
    // -------------------------------------------------------------------------------- last position
    // sh_gfin holds gamma[T][.]; candidates fin[to], to = 0..C (C = EOS): first maximal entry wins.
    __syncthreads();
#ifdef SMM_PROFILE
    const unsigned long long p_kern2 = __builtin_readcyclecounter();
#endif
    const bool no_eos = (a.flags & 8) != 0;    // add_eos=False: T counts the frames before the last one (see smmdp.h)
    if (w == 0) {
        double f = SMM_NEG_INF;
        const int last = no_eos ? C - 1 : C;   // candidates: the real labels, and EOS unless there is none
        if (lane <= last) {
            for (int c = 0; c < C; ++c) {
                const double wgt = (lane == C) ? (endpen ? endpen[c] : 0.0) : trans[(size_t)lane * cm + c];
                f = fmax(f, sh_gfin[c] + wgt);
            }
            if (no_eos) f = f + elp[(size_t)T * cm + lane];   // the closing label only emits frame T
            else if (lane < C) f = f + SMM_BIG_NEG;
        }
        int kk = 0, cc = (lane <= last) ? lane : 0x7fffffff;
        if (lane > last) f = SMM_NEG_INF;
        smm_wave_best3(f, kk, cc);
        if (lane == 0) {
            sh_c = cc;
            if (a.best) a.best[vid] = f;
            if (spans) spans[T] = cmap ? cmap[cc] : cc;
            if (no_eos && labels) labels[T] = cmap ? cmap[cc] : cc;
        }
    }
    __threadfence_block();
    __syncthreads();

    // -------------------------------------------------------------------------------- back-trace
#ifdef SMM_DEV
    if (a.flags & 1) return;                   // (profiling builds: stop after the forward pass, outputs undefined)
#endif
    // At a span start (n, to) the predecessor is the FIRST (k ascending, then from ascending) whose
    //   (cumE[n][from] + (h[n-k][from] + len[k][from])) + w(to, from)  equals the maximum.
    // Adding is monotone, so a hit needs gamma[n][from] + w(to, from) == maximum: phase A (every wave, redundantly,
    // one lane per state) finds the maximum and the usually single state that attains it from 8*C bytes of the
    // gamma history; phase B scans only that state's row for the first k (16 B per candidate), all waves abreast.
    int n = T, to = sh_c, nseg = 0, round = 0;
    if constexpr (R == 1) {
        // -------- short segments (kp <= 64: the reference's default --sm_max_span_length 20, cfg4's 64): WINDOW back-trace.
        // The general path below makes one or two round trips to the history per SEGMENT (1.6 us each) with the whole
        // workgroup; with short segments a video is hundreds of them.  Here a window of the history (W positions: the
        // gamma and cumE rows, the h rows, 24 C bytes per position) is staged in LDS in bulk by all waves, then wave 0
        // walks every segment that ends inside it from LDS alone -- lane = state for the maximum, lane = length for the
        // first length that attains it: no barrier, no memory round trip per segment.  Same expressions, same
        // first-(k, state) order as the general path.  a.bt_window = W (0: the general path; the host sizes the dynamic LDS).
        const int W = a.bt_window;
        if (W > 0) {
            extern __shared__ __attribute__((aligned(16))) double smm_dyn[];
            double *win_g = smm_dyn, *win_c = win_g + (size_t)W * C, *win_h = win_c + (size_t)W * C;
            double *tab_t = win_h + (size_t)W * C, *tab_l = tab_t + (size_t)C * C;     // trans [to][from], len [k][c]
            for (int e = threadIdx.x; e < C * C; e += blockDim.x) tab_t[e] = trans[(size_t)(e / C) * cm + e % C];
            for (int e = threadIdx.x; e < kp * C; e += blockDim.x) tab_l[e] = len[(size_t)(e / C) * cm + e % C];
            // lane c: the global id of local label c (a load per segment from the class map would be a trip to L2 each)
            const int64_t gid_l = (lane <= C) ? (cmap ? cmap[lane] : (int64_t)lane) : 0;
            const int gid_lo = (int)(gid_l & 0xffffffff), gid_hi = (int)(gid_l >> 32);
            // ... and lane c: the end penalty of state c.  (No global LOAD inside the walk: a wave that waits for a load
            // waits for its older stores too -- vmcnt counts both, in order -- and every segment stores its labels.)
            const double ep_l = (endpen && lane < C) ? endpen[lane] : 0.0;
            bool bad = false;
            while (n > 0) {
                // window = positions n0 .. n  (W of them, or all that are left)
                const int n0 = (n - W + 1 > 0) ? n - W + 1 : 0, rows = n - n0 + 1;
                __syncthreads();                                   // the previous window has been walked
                for (int e = threadIdx.x; e < rows * C; e += blockDim.x) {
                    win_g[e] = hgam[(size_t)n0 * C + e];
                    win_c[e] = hcum[(size_t)n0 * C + e];
                }
                for (int e = threadIdx.x; e < rows * C; e += blockDim.x) {
                    const int c2 = e / rows, i = e % rows;         // (h is state-major in HBM: consecutive threads, consecutive positions)
                    win_h[(size_t)i * C + c2] = hh[(size_t)c2 * (T + 1) + n0 + i];
                }
                __syncthreads();
                if (w == 0) {
                    // every segment whose candidates all lie inside the window
                    while (n > 0 && (n0 == 0 || n - (kp - 1) >= n0)) {
                        const int kmax = (kp - 1 < n) ? kp - 1 : n;
                        const int row = n - n0;
                        double wgt = 0.0, gmv = SMM_NEG_INF, cnl = 0.0;
                        if (lane < C) {
                            cnl = win_c[(size_t)row * C + lane];
                            wgt = (to == C) ? ep_l : tab_t[(size_t)to * C + lane];
                            gmv = win_g[(size_t)row * C + lane] + wgt;
                        }
                        // a NaN in the row (the inputs broke the contract): by its bits -- this unit is compiled with
                        // -fno-honor-nans, a compare that is meant to fail proves nothing here (smm_nan_bits)
                        if (__ballot(lane < C && smm_nan_bits(gmv)) != 0) { bad = true; break; }
                        const double rmax = smm_row_max16(gmv);
                        const double best = fmax(smm_readlane(rmax, 0), smm_readlane(rmax, 16));
                        unsigned long long fmask = __ballot(lane < C && gmv == best);
                        int k = 0x7fffffff, c = 0x7fffffff;
                        while (fmask) {
                            const int f = __builtin_amdgcn_readfirstlane(__ffsll(fmask) - 1);
                            fmask &= fmask - 1;
                            const double cn = smm_readlane(cnl, f), wf = smm_readlane(wgt, f);
                            const int lim = (kmax < k - 1) ? kmax : k - 1;      // an equal k with a larger state loses
                            const int kk = lane + 1;
                            bool hit = false;
                            if (kk <= lim) hit = ((cn + (win_h[(size_t)(row - kk) * C + f] + tab_l[(size_t)kk * C + f])) + wf) == best;
                            const unsigned long long m = __ballot(hit);
                            if (m) { k = __ffsll(m); c = f; }
                        }
                        if (k < 1 || k > kmax || c < 0 || c >= C) { bad = true; break; }   // NaN / inf - inf in the inputs
                        const int s0 = n - k;
                        const int64_t gid = ((int64_t)__builtin_amdgcn_readlane(gid_hi, c) << 32) |
                                            (uint32_t)__builtin_amdgcn_readlane(gid_lo, c);
                        if (labels && lane < k) labels[s0 + lane] = gid;
                        if (spans && lane == 0) spans[s0] = gid;
                        ++nseg;
                        n = s0;
                        to = c;
                    }
                    if (lane == 0) { sh_kmin[0] = (unsigned)n; sh_kmin[1] = (unsigned)to; sh_kmin[2] = bad ? 1u : 0u; }
                }
                __syncthreads();
                n = (int)sh_kmin[0];
                to = (int)sh_kmin[1];
                if (sh_kmin[2]) {
                    if (threadIdx.x == 0) atomicExch(a.err, 1);
                    break;
                }
            }
            if (a.n_segs && threadIdx.x == 0) a.n_segs[vid] = nseg + (no_eos ? 1 : 0);
            return;
        }
    }
#ifdef SMM_PROFILE
    unsigned long long bt_a = 0, bt_b = 0, bt_c = 0, bt_t = __builtin_readcyclecounter();   // phase A / B / labels, workgroup 0
#define SMM_BT_STAMP(acc) do { const unsigned long long t_ = __builtin_readcyclecounter(); acc += t_ - bt_t; bt_t = t_; } while (0)
#else
#define SMM_BT_STAMP(acc) do { } while (0)
#endif
    if (threadIdx.x == 0) { sh_kmin[0] = 0xffffffffu; sh_kmin[1] = 0xffffffffu; sh_kmin[2] = 0xffffffffu; }
    // Phase B depends on phase A (which state's column to scan), and each is a round trip to a history that has long
    // left the caches (~2000 cycles per trip).  The column of the state that preceded `to` the LAST time (at the
    // start: the most likely one a priori, arg-max of its transition row) is therefore fetched SPECULATIVELY together
    // with phase A's rows; when phase A confirms the guess -- nearly always on ordered tasks -- the segment costs one
    // round trip.  Same expressions, same first-(k, state) order: a wrong guess only costs the second trip.
    // Round 4: (1) the lengths of a column come from the state-major table in BAND mode (the [k][c] table is one cache line
    // per candidate: 512 lines per round against 64); (2) the loop's barriers wait for LDS only, every wave keeps the guess
    // table current by itself, and (3) the NEXT segment's trip is issued in front of this segment's label stores: a segment
    // no longer waits for its predecessor's stores to be acknowledged, then for its own loads.  cfg3's longest video (92
    // segments): 5040 -> 3380 cycles per segment, of which ~2000 are the one trip (profiles/round4_backtrace.txt).
    if (threadIdx.x <= (unsigned)C) {
        const int t2 = threadIdx.x;
        int bi = C - 1;
        if (t2 < C) {
            double bv = SMM_NEG_INF;
            bi = 0;
            for (int c2 = 0; c2 < C; ++c2) {
                const double v2 = trans[(size_t)t2 * cm + c2];
                if (v2 > bv) { bv = v2; bi = c2; }
            }
        } else if (endpen) {
            for (int c2 = 0; c2 < C; ++c2)
                if (endpen[c2] == 0.0) bi = c2;
        }
        sh_guess[t2] = bi;
    }
    __syncthreads();
    const double *lcol0 = nullptr;                            // BAND: [c][k + 1] = len[k][c], a state's lengths contiguous
    if constexpr (BAND) lcol0 = a.len_t + (size_t)g * cm * SMM_BAND_ROW + 1;
    // one round trip for everything phase A and B need of position n_: gamma row, cumE row, weights, and the first round
    // of phase B for the guessed state (clamped, unconditional loads)
    int fg = 0, kmax = 0;
    double wgt = 0.0, g0 = SMM_NEG_INF, cnl = 0.0, sp_h = 0.0, sp_l = 0.0;
    auto trip = [&](int n_, int to_) {
        fg = sh_guess[to_];                                  // (this wave's own lane 0 wrote it last: see below)
        kmax = (kp - 1 < n_) ? kp - 1 : n_;
        if (lane < C) {
            g0 = hgam[(size_t)n_ * C + lane];
            cnl = hcum[(size_t)n_ * C + lane];
            wgt = (to_ == C) ? (endpen ? endpen[lane] : 0.0) : trans[(size_t)to_ * cm + lane];
        }
        const int kk0 = w * 64 + lane + 1;
        const int kc = kk0 <= kmax ? kk0 : kmax;
        sp_h = hh[(size_t)fg * (T + 1) + n_ - kc];
        sp_l = BAND ? lcol0[(size_t)fg * SMM_BAND_ROW + kc] : len[(size_t)kc * cm + fg];
    };
    // (lane c keeps the global id of state c: a load of cmap[c] behind the trip's loads would wait for all of them)
    const int64_t gid_l = cmap ? cmap[lane < C ? lane : C] : (int64_t)lane;
    if (n > 0) trip(n, to);
    while (n > 0) {
        const double gmv = (lane < C) ? g0 + wgt : SMM_NEG_INF;
        // a NaN in the row (the inputs broke the contract): detected by its bits, in every wave alike (smm_nan_bits)
        const bool nan_row = __ballot(lane < C && smm_nan_bits(gmv)) != 0;
        const double rmax = smm_row_max16(gmv);
        const double best = fmax(smm_readlane(rmax, 0), smm_readlane(rmax, 16));
        unsigned long long fmask = nan_row ? 0ull : __ballot(lane < C && gmv == best);
        int k = 0x7fffffff, c = 0x7fffffff;
        SMM_BT_STAMP(bt_a);
        while (fmask) {
            const int f = __builtin_amdgcn_readfirstlane(__ffsll(fmask) - 1);
            fmask &= fmask - 1;
            const double cn = smm_readlane(cnl, f);
            const double wf = smm_readlane(wgt, f);
            const double *hcol = hh + (size_t)f * (T + 1);     // contiguous: this state's h over all positions
            const double *lcol = BAND ? lcol0 + (size_t)f * SMM_BAND_ROW : len + f;
            const int lim = (kmax < k - 1) ? kmax : k - 1;      // an equal k with a larger state loses
            for (int kb = 0; kb < lim; kb += NW * 64) {
                const int kk = kb + w * 64 + lane + 1;
                bool hit = false;
                if (kb == 0 && f == fg) {                       // the guess was right: its first round is already here
                    if (kk <= lim) hit = ((cn + (sp_h + sp_l)) + wf) == best;
                } else if (kk <= lim) hit = ((cn + (hcol[n - kk] + lcol[BAND ? (size_t)kk : (size_t)kk * cm])) + wf) == best;
                const unsigned long long m = __ballot(hit);
                const int slot = round % 3;
                if (lane == 0 && m) atomicMin(&sh_kmin[slot], (unsigned)(kb + w * 64 + __ffsll(m)));
                if (threadIdx.x == 0) sh_kmin[(round + 1) % 3] = 0xffffffffu;
                smm_lds_barrier();                              // (LDS only: the label stores below are never waited for)
                const unsigned kf = sh_kmin[slot];
                ++round;
                if (kf != 0xffffffffu) {
                    if ((int)kf < k) { k = (int)kf; c = f; }
                    break;
                }
            }
        }
        SMM_BT_STAMP(bt_b);
        if (nan_row || k < 1 || k > kmax || c < 0 || c >= C) {   // NaN in the inputs / no candidate attains the maximum: stop, flag, never spin
            if (threadIdx.x == 0) atomicExch(a.err, 1);
            break;
        }
        const int s = n - k;
        // every wave writes the guess it has just learned (the same value in all of them) and reads its OWN write back at
        // its next visit of that state: LDS operations of a wave execute in order, no barrier needed.  (An entry another
        // wave wrote was written a segment or more ago, i.e. in front of a barrier this wave has passed since.)
        if (lane == 0) sh_guess[to] = c;
        const int n0 = n;
        n = s;
        to = c;
        if (n > 0) trip(n, to);                                // the next segment's trip goes out in front of the stores
        const int64_t gid = ((int64_t)__builtin_amdgcn_readlane((int)(gid_l >> 32), c) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)gid_l, c);
        if (labels)
            for (int f = s + threadIdx.x; f < n0; f += blockDim.x) labels[f] = gid;
        if (threadIdx.x == 0 && spans) spans[s] = gid;
        ++nseg;
        SMM_BT_STAMP(bt_c);
    }
#ifdef SMM_PROFILE
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long *pp = reinterpret_cast<unsigned long long *>(a.err);
        pp[40] = bt_a; pp[41] = bt_b; pp[42] = bt_c; pp[43] = (unsigned long long)nseg;
        pp[44] = p_kern1 - p_kern0; pp[45] = p_kern2 - p_kern1; pp[46] = __builtin_readcyclecounter() - p_kern2;
    }
#endif
    if (a.n_segs && threadIdx.x == 0) a.n_segs[vid] = nseg + (no_eos ? 1 : 0);
#ifdef SMM_PROFILE_END   // diagnostic build: when did each video's (leader) workgroup finish?  (100 MHz wall clock into best[])
    if (a.best && threadIdx.x == 0) a.best[vid] = (double)wall_clock64();
#endif
}

// ------------------------------------------------------------------------------------------------ band tables
// Per (parameter group, state): the state-major copy of the length table (a band that is switched back on reads its ring
// from it, 128 consecutive lengths per wave instruction) and the bounds of the skip test: [0] min len over 33..174,
// [m] max len over band m = 16+112m .. 127+112m (clipped to the table; a video's own kp only makes the bound looser).
__global__ void __launch_bounds__(256) smm_band_tables_kernel(const double *len, const int32_t *n_states, double *len_t,
                                                              double *band_tab, double *dmin_t, int cm, int k_rows)
{
    // the state's column of the length table, once from memory (strided by c_max: every load is a trip to L2), then
    // every bound from LDS: this kernel sits in front of the DP kernel on the critical path of every decode (round 3: nine
    // serial loops of ~140 strided loads took 35 us; round 4's 54 witness windows straight from memory 23 us)
    __shared__ double col[SMM_BAND_ROW];                          // col[k] = len[k] (-inf beyond the table / for a dead state)
    const int g = blockIdx.x / cm, c = blockIdx.x % cm;
    const double *src = len + (size_t)g * k_rows * cm + c;
    double *dst = len_t + ((size_t)g * cm + c) * SMM_BAND_ROW;     // row[k + 1] = len[k]; -inf beyond the table
    const bool live = c < n_states[g];
    for (int k = threadIdx.x; k < SMM_BAND_ROW; k += blockDim.x) {
        const double v = (live && k >= 1 && k - 1 < k_rows) ? src[(size_t)(k - 1) * cm] : SMM_NEG_INF;
        dst[k] = v;
        if (k >= 1) col[k - 1] = v;
    }
    if (threadIdx.x == 0) col[SMM_BAND_ROW - 1] = SMM_NEG_INF;
    __syncthreads();
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    // [0] min len over 33..174 (clipped to the table: the kernel checks the video's own limit), [m] max len over band m
    for (int m = wv; m <= SMM_BAND_N; m += 4) {
        const int k0 = m ? SMM_BAND_LO + SMM_BAND_DELAY * m : 33, k1 = m ? 127 + SMM_BAND_DELAY * m : 174;
        double v = m ? SMM_NEG_INF : __builtin_huge_val();
        for (int k = k0 + ln; k <= k1 && k < k_rows; k += 64) {
            const double x = col[k];
            v = m ? fmax(v, x) : fmin(v, x);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double o = __shfl_xor(v, off);
            v = m ? fmax(v, o) : fmin(v, o);
        }
        if (!live) v = SMM_NEG_INF;
        if (ln == 0) band_tab[((size_t)g * cm + c) * SMM_BAND_TAB + m] = v;
    }
    // ANCHOR (round 5, see the pusher waves): D[d] = min over band 0's lengths k = 9 .. 127 of (len[k + d] - len[k]), pushed a
    // part in 2^49 DOWN (the test h[s] - h[a] < D[s - a] must hold in real arithmetic, strictly), and of that the minimum
    // over each bucket of 16 distances, dmin_t[q] = min_{16 q <= d < 16 q + 16, d >= 1} D[d]: a lower bound of D[d] serves as
    // well and 64 numbers per state fit the LDS.  -inf where some k + d of the bucket leaves the table (no such candidate:
    // never beaten); a length the table does not reach (len[k] = -inf) asks for nothing.
    if constexpr (SMM_ANCHOR != 0) {
        const int klo = 9, khi = (k_rows - 1 < 127) ? k_rows - 1 : 127;
        double *drow = dmin_t + ((size_t)g * cm + c) * 64;
        if (threadIdx.x < 64) {
            const int q = threadIdx.x;
            double mq = __builtin_huge_val();
            for (int d = (q ? 16 * q : 1); d < 16 * q + 16; ++d) {
                double m = __builtin_huge_val();
                if (!live || khi + d > k_rows - 1) m = SMM_NEG_INF;
                else
                    for (int k = klo; k <= khi; ++k) {
                        const double lb = col[k];
                        if (lb == SMM_NEG_INF) continue;
                        m = fmin(m, col[k + d] - lb);
                    }
                mq = fmin(mq, m);
            }
            drow[q] = mq > 0.0 ? mq * (1.0 - 0x1p-49) : mq * (1.0 + 0x1p-49);
        }
    }
    // the witnesses of the skip test (round 4): min len over the lengths 16 delta + 1 .. 16 delta + 142, a wave per window
    // (-inf when the window leaves the table: no witness there)
    for (int d = 2 + wv; d <= SMM_BAND_WIT; d += 4) {
        const int k0 = 16 * d + 1, k1 = 16 * d + 142;
        double v = __builtin_huge_val();
        for (int k = k0 + ln; k <= k1; k += 64) v = fmin(v, k < k_rows ? col[k] : SMM_NEG_INF);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v = fmin(v, __shfl_xor(v, off));
        if (!live) v = SMM_NEG_INF;
        if (ln == 0) band_tab[((size_t)g * cm + c) * SMM_BAND_TAB + 16 + d] = v;
    }
}

void smm_launch_band_tables(const double *len, const int32_t *n_states, double *len_t, double *band_tab, double *dmin_t,
                            int n_groups, int cm, int k_rows, hipStream_t stream)
{
    hipLaunchKernelGGL(smm_band_tables_kernel, dim3(n_groups * cm), dim3(256), 0, stream, len, n_states, len_t, band_tab, dmin_t, cm, k_rows);
}

// ------------------------------------------------------------------------------------------------ dispatch
#include <cstdlib>
#include "../../include/smmdp.h"
#include "smm_launch.h"

// Configuration: 1 chain wave + NP pusher waves x SPW states.  VALU code can only address the 256 architected VGPRs (the
// other half of the unified file are AGPRs) and a pusher needs ~4*R*SPW + 40 of them.  Fewer, fatter pushers are
// preferred (fewer waves per barrier), but at least one pusher per SIMD.  K <= 512 (R <= 8): up to 32 states on 8 waves.
template <int R, int SPW, int NW>
static int launch_if(const SmmDpArgs &a, int spw, int nw, int c_need, hipStream_t stream)
{
    if (spw != SPW || nw != NW) return 0;
    constexpr int B = (R == SMM_B8_R && NW == 8) ? 8 : SMM_B;
    // R = 1 (kp <= 64): the window back-trace stages history and tables in dynamic LDS (a.bt_dyn_bytes, up to ~125 KB:
    // the 64 KB default limit of dynamic LDS is lifted once per kernel and device)
    const size_t dyn = (R == 1 && a.bt_window > 0) ? (size_t)a.bt_dyn_bytes : 0;
    auto go = [&](auto kernel) {
        if constexpr (R == 1) {
            static bool raised[64] = {};
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
            if (!raised[dev]) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                raised[dev] = true;
            }
        }
        hipLaunchKernelGGL(kernel, dim3(a.b), dim3(NW * 64), dyn, stream, a);
    };
    // HF: source states per lane group of the chain wave (4: four groups of 16 lanes; else two groups, 2 HF >= states)
    // blocks of 8: the pushers push the block's own sources (D = 0, as in BAND mode): with D = 1 the chain wave keeps 32
    // positions of h instead of 16, and with the speculative transition on top it spills
    constexpr int DD = (B == 8) ? 0 : SMM_D;
    if (c_need <= 16) go(smm_viterbi_kernel<R, SPW, NW, 4, B, DD>);
    else if (c_need <= 24) go(smm_viterbi_kernel<R, SPW, NW, 12, B, DD>);
    else go(smm_viterbi_kernel<R, SPW, NW, 16, B, DD>);
    return 1;
}

// BAND mode (K > 512): 8 waves, six state-owning pusher waves with 3 states each up to 18 states, 4 up to 24, 5 up to 30,
// 6 up to 32; HF = source states per lane group of the chain wave
template <int TAG>
static int launch_band(const SmmDpArgs &a, int c_need, hipStream_t stream)
{
    const dim3 grid(a.b), block(512);
#ifdef SMM_DEV_BAND_ONE   // development builds: one instantiation
    (void)c_need;
#ifndef SMM_DEV_BAND_SPW
#define SMM_DEV_BAND_SPW 4
#define SMM_DEV_BAND_HF 12
#endif
#ifndef SMM_DEV_BAND_NW
#define SMM_DEV_BAND_NW 8
#endif
    hipLaunchKernelGGL((smm_viterbi_kernel<16, SMM_DEV_BAND_SPW, SMM_DEV_BAND_NW, SMM_DEV_BAND_HF, 8, 0, true, TAG>), grid, dim3(SMM_DEV_BAND_NW * 64), 0, stream, a);
    return SMM_OK;
#else
    if (c_need <= 16) hipLaunchKernelGGL((smm_viterbi_kernel<16, 3, 8, 4, 8, 0, true, TAG>), grid, block, 0, stream, a);
    else if (c_need <= 18) hipLaunchKernelGGL((smm_viterbi_kernel<16, 3, 8, 12, 8, 0, true, TAG>), grid, block, 0, stream, a);
    else if (c_need <= 24) hipLaunchKernelGGL((smm_viterbi_kernel<16, 4, 8, 12, 8, 0, true, TAG>), grid, block, 0, stream, a);
    else if (c_need <= 30) hipLaunchKernelGGL((smm_viterbi_kernel<16, 5, 8, 16, 8, 0, true, TAG>), grid, block, 0, stream, a);
    else if (c_need <= 32) hipLaunchKernelGGL((smm_viterbi_kernel<16, 6, 8, 16, 8, 0, true, TAG>), grid, block, 0, stream, a);
    else return SMM_ERR_UNSUPPORTED;
    return SMM_OK;
#endif
}

template <int R>
static int launch_r(const SmmDpArgs &a, int c_need, hipStream_t stream)
{
    // 8 waves (256 VGPRs per wave): 7 pushers x up to 5 states; 4 waves for the short rings of small launches
    const int nw = 8;
    const int spw = (c_need + nw - 2) / (nw - 1);
    int hit = 0;
    if constexpr (R <= 4) {
        hit = launch_if<R, 1, 8>(a, spw, nw, c_need, stream) || launch_if<R, 2, 8>(a, spw, nw, c_need, stream) ||
              launch_if<R, 3, 8>(a, spw, nw, c_need, stream) || launch_if<R, 4, 8>(a, spw, nw, c_need, stream) ||
              launch_if<R, 5, 8>(a, spw, nw, c_need, stream);
    } else if constexpr (R == 8) {
        // (five states per pusher with 512-slot rings do not fit the registers: 29..32 states at K > 256 run in BAND mode,
        // smm_api.hip: band_mode)
        hit = launch_if<R, 1, 8>(a, spw, nw, c_need, stream) || launch_if<R, 2, 8>(a, spw, nw, c_need, stream) ||
              launch_if<R, 3, 8>(a, spw, nw, c_need, stream) || launch_if<R, 4, 8>(a, spw, nw, c_need, stream);
    }
    return hit ? SMM_OK : SMM_ERR_UNSUPPORTED;
}

// SMALL workgroups (round 5): chain wave, mover wave and TWO pusher waves with up to 8 states each -- four waves, <= 16 states,
// 238 VGPRs at two waves per SIMD and 54 KB of LDS, so that a CU holds TWO of them.  One video on its own CU runs 25..35 %
// slower this way (the pushers' per-state work doubles; four waves wait for each other as eight did), a launch part with
// more videos than CUs 1.2..1.3 x faster (profiles/round5_two_per_cu.txt): the host uses it for the <= 16-state videos of
// such parts only (smm_api.hip: run_viterbi), beside the eight-wave launch of the others.  Same code, same bits.
int smm_launch_viterbi_small(const SmmDpArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL((smm_viterbi_kernel<16, 8, 4, 4, 8, 0, true, 0>), dim3(a.b), dim3(256), 0, stream, a);
    return SMM_OK;
}

// The repair launch of a time-split decode (smm_chunk.hip): one workgroup per split video, at work only for the videos the
// stitch could not certify -- the launch's own BAND configuration under its second name (TAG = 1)
int smm_launch_viterbi_repair(const SmmDpArgs &a, int c_need, hipStream_t stream)
{
    return launch_band<1>(a, c_need, stream);
}

int smm_launch_viterbi(const SmmDpArgs &a, int r, int c_need, hipStream_t stream)
{
#ifdef SMM_DEV_BAND_ONLY   // development builds: instantiate the BAND kernels only (quick compile-measure cycles)
    (void)r;
    return launch_band<0>(a, c_need, stream);
#else
    // SMM_DEV_R (development builds only): instantiate one ring size, for quick compile-measure cycles
#ifndef SMM_DEV_R
#define SMM_DEV_R 0
#endif
#define SMM_CASE_R(rr) case rr: if constexpr (SMM_DEV_R == 0 || SMM_DEV_R == rr) return launch_r<rr>(a, c_need, stream); else break;
    switch (r) {
    SMM_CASE_R(1)
    SMM_CASE_R(2)
    SMM_CASE_R(4)
    SMM_CASE_R(8)
    case 16: if constexpr (SMM_DEV_R == 0 || SMM_DEV_R == 16) return launch_band<0>(a, c_need, stream); else break;
    default: break;
    }
#undef SMM_CASE_R
    return SMM_ERR_UNSUPPORTED;
#endif
}
