// smm_device.h -- shared device-side helpers and launch metadata for libsmmdp (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SMM_BIG_NEG (-1e9)   // reference BIG_NEG, semimarkov_modules.py:20
#define SMM_NEG_INF (-__builtin_huge_val())
#define SMM_MAX_STATES_DEV 32
// Viterbi BAND mode (smm_viterbi.hip)
#define SMM_BAND_DELAY 112
#define SMM_BAND_LO 16
#define SMM_BAND_N 8          // delayed bands
#define SMM_BAND_ROW 1026      // doubles per state of the shifted state-major length table (row[k + 1] = len[k], k <= 1024)
#define SMM_L0_ROW 130         // doubles per state of band 0's length table in LDS (ring distances 0..128 + padding)
#define SMM_BAND_TAB 80       // doubles per (group, state) in SmmDpArgs::band_tab: [0] min len over 33..174, [m] max len over band m,
                              // [16 + delta] min len over 16 delta + 1 .. 16 delta + 142 for delta = 2 .. SMM_BAND_WIT (the lengths that
                              // connect the sources of group g - delta with the targets the band-groups of group g reach)
#define SMM_BAND_WIT 55       // oldest witness group of the band skip test: 16 * 55 + 142 = 1022 <= the longest length

// One entry per video, built on the host by smm_plan() and staged into the workspace.
struct SmmVideo {
    int64_t frame_off;   // first frame on the packed frame axis
    int64_t hist_off;    // offset (in doubles) of this video's history block in the workspace
    int32_t T;           // frames
    int32_t group;       // parameter group
    int32_t kp;          // usable segment lengths are 1..kp-1   (min(K, Tmax of the reference batch))
    int32_t pad;         // Viterbi, time-split videos (round 5, smm_viterbi.hip: CHUNK units): bit 0 = this entry is a unit (a
                         // forward pass over part of a video: no closing step, no outputs), bit 1 = it starts from the guess
                         // (not the video's first unit), bits 2.. = its first position within the parent video
};

#define SMM_CHUNK_MAX_UNITS 64   // units per time-split video (smm_chunk.hip)
// One per time-split video: its units are videos[first_unit .. first_unit + n_chunks) of the extended array, in time order
struct SmmChunkVideo {
    int32_t vid;         // the parent video
    int32_t first_unit;  // index into the extended SmmVideo array (>= b)
    int32_t n_chunks;
    int32_t ov;          // positions a unit runs before its own part begins (warm-up + kp - 1)
};

struct SmmDpArgs {
    const SmmVideo *videos;    // [b]
    const int32_t *order;      // [b] block -> video (longest first)
    const int32_t *n_states;   // [n_groups]
    const double *elp;         // [total_frames][c_max]
    const double *trans;       // [g][c_max][c_max]  [to][from]
    const double *init;        // [g][c_max]
    const double *len;         // [g][k_rows][c_max]
    const double *endpen;      // [b][c_max] or null
    const int64_t *class_map;  // [g][c_max+1] or null
    double *hist;              // per video (Viterbi): cum[T+1][C], h[C][T+1], gamma[T+1][C]; C = the video's states
    int64_t *spans;            // [b][t_max+1] or null
    int64_t *labels;           // [total_frames] or null
    double *best;              // [b] or null
    int32_t *n_segs;           // [b] or null
    int32_t *err;              // [0] sticky error flag (NaN in the inputs); [1] always 0 (rounds 1-3: gang time-outs); [2] Viterbi BAND mode, diagnostic: sources pushed into band 0;
                               // [3] Viterbi BAND mode, diagnostic: delayed band-blocks (the sources of one hand-over block, 8 or under SMM_BAND_B=4 4, x one band of one state) evaluated
    int32_t c_max, k_rows, t_max, b;
    int32_t flags;             // bit 0: profiling only, -DSMM_DEV builds -- stop after the forward pass (outputs undefined);
                               // bit 1: logZ backward; bit 3 (8): no EOS (add_eos=False; SmmVideo::T = frames - 1);
                               // bit 6 (64): logZ forward and time-reversed runs in one launch (2 workgroups per video);
                               // bit 7 (128): Viterbi BAND mode; bit 8 (256): Viterbi without the speculative transition (A/B aid)
    const double *trans_t;     // logZ, both directions in one launch (flags bit 6): transposed tables [g][c_max][c_max]
    double *logz_b;            // ... and where the reversed runs put their closing value [b]
    const double *len_t;       // Viterbi BAND mode (flags bit 7, 128): [g][c_max][k_rows] state-major length table ...
    const double *band_tab;    // ... and [g][c_max][16] bounds of the band skip test (smm_viterbi.hip: smm_band_tables_kernel)
    const double *dmin_t;      // ... and [g][c_max][64] thresholds of the anchor test, minima over buckets of 16 distances (round 5: ANCHOR in the BAND pushers, -DSMM_ANCHOR=1 builds only)
    const int32_t *redo;       // Viterbi, the repair launch of a time-split decode: [grid] 0 = this workgroup has nothing to do
    const double *chunk_anchor;// Viterbi, time-split videos: [units][c_max] cumE at each unit's first position (smm_cum_anchor_kernel)
    int32_t b_videos;          // ... and the number of real videos (units are entries b_videos.. of `videos`)
    int32_t bt_window;         // Viterbi, kp <= 64: positions per LDS window of the back-trace (0: the general back-trace) ...
    int32_t bt_dyn_bytes;      // ... and the dynamic LDS it needs: (3 W + c + kp) c doubles for the launch's largest c, kp
};

// One-CU videos on 8 waves: the rank (0..6) that trades places with the chain wave's partner (rank 6, the lightest) so
// that no SIMD carries two of the fuller pusher waves; -1: none.  Ranks below C % 7 own one state more and the waves of
// ranks (0,4), (1,5), (2,3) share a SIMD: with FOUR fuller ranks (18 states: 3+2 | 3+2 | 3+3 | chain+2) the partner
// takes rank 3 instead (3+2 | 3+2 | 3+2 | chain+3: 438 -> 402 ns per frame).  With ONE fuller rank (15 states) the same
// trade measured worse (391 against 365 ns: the chain wave and the mover duty of its partner weigh ~2.5 states).
// Shared with the host's cost model (smm_api.hip: frame_ns_single).
__host__ __device__ inline int smm_rebalanced_rank(int c)
{
    return (c >= 14 && c % 7 == 4) ? 3 : -1;
}

// ---- 64-bit register helpers -------------------------------------------------------------------
__device__ __forceinline__ double smm_pack(int lo, int hi) { return __hiloint2double(hi, lo); }

__device__ __forceinline__ double smm_readlane(double x, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return smm_pack(lo, hi);
}

// DPP move of a double; lanes whose source is out of range keep their own value (bound_ctrl = 0, old = src).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double smm_dpp(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return smm_pack(lo, hi);
}

#define SMM_DPP_ROW_SHR(n) (0x110 + (n))
#define SMM_DPP_ROW_SHL(n) (0x100 + (n))   // lane i <- lane i + n of its row of 16
#define SMM_DPP_WAVE_ROR1 0x13C
#define SMM_DPP_ROW_BCAST15 0x142
#define SMM_DPP_ROW_BCAST31 0x143

// lane i <- lane i-1, lane 0 <- lane 63
__device__ __forceinline__ double smm_wave_ror1(double x) { return smm_dpp<SMM_DPP_WAVE_ROR1>(x); }

__device__ __forceinline__ float smm_wave_ror1f(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), SMM_DPP_WAVE_ROR1, 0xf, 0xf, false));
}

// NaN by its BITS (exponent all ones, mantissa non-zero).  The Viterbi translation unit is compiled with -fno-honor-nans
// (see smm_fmax): a floating-point compare that is meant to FAIL for a NaN is formally poison there and may be folded, so
// the error-word contract of smmdp.h ("a NaN reached the DP -> err[0] = 1, the back-trace stops") rests on integer tests
// only.  The empty asm keeps the compiler from reasoning about where the bits came from.
__device__ __forceinline__ bool smm_nan_bits(double x)
{
    int hi = __double2hiint(x), lo = __double2loint(x);
    asm volatile("" : "+v"(hi), "+v"(lo));
    return (hi & 0x7ff00000) == 0x7ff00000 && (((hi & 0x000fffff) | lo) != 0);
}

// v_max_f64 without the canonicalising self-max hipcc puts in front of fmax() operands it cannot prove quiet
// (inputs are finite or -inf by contract; a NaN that enters anyway -- or an inf - inf of the two one-sided tests that subtract
// possibly infinite values, DOM's h[s+1] - h[s] and SPEC's gamma[c] - gamma[cs]: their compares read a NaN as "not beaten" /
// "objects", the conservative side -- is swallowed by v_max_f64 or caught by smm_nan_bits in the back-trace).
__device__ __forceinline__ double smm_fmax(double a, double b)
{
#ifdef SMM_FMAX_BUILTIN
    // translation units compiled with -fno-honor-nans: fmax lowers to a bare v_max_f64 there, and -- unlike the inline asm --
    // the compiler knows its latency and hazards (no s_nop behind every max, free scheduling)
    return __builtin_fmax(a, b);
#else
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}

// max(x[lane], x[lane ^ 32]) in every lane: one v_permlane32_swap per 32-bit half
__device__ __forceinline__ double smm_max_halves(double x)
{
    const int lo = __double2loint(x), hi = __double2hiint(x);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return smm_fmax(smm_pack(a[0], b[0]), smm_pack(a[1], b[1]));
}

// x[lane] + x[lane ^ 32] in every lane (fp32): one v_permlane32_swap
__device__ __forceinline__ float smm_sum_halvesf(float x)
{
    const int v = __float_as_int(x);
    auto a = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return __int_as_float(a[0]) + __int_as_float(a[1]);
}

// x[lane] + x[lane ^ 16] in every lane (fp32): one v_permlane16_swap
__device__ __forceinline__ float smm_sum_rows16f(float x)
{
    const int v = __float_as_int(x);
    auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return __int_as_float(a[0]) + __int_as_float(a[1]);
}

// max(x[lane], x[lane ^ 16]) in every lane: one v_permlane16_swap per 32-bit half (rows 1 <-> 0 and 3 <-> 2 trade places)
__device__ __forceinline__ double smm_max_rows16(double x)
{
    const int lo = __double2loint(x), hi = __double2hiint(x);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return smm_fmax(smm_pack(a[0], b[0]), smm_pack(a[1], b[1]));
}

// Max over each DPP row (16 consecutive lanes) of an unsigned; every lane of the row receives it.
// v_max_u32 with a rotated DPP source, 4 levels.  (2 wait states between a VALU write and a DPP read of
// the same VGPR: hipcc pads nothing inside asm, so the s_nop's are part of the sequence.)
__device__ __forceinline__ unsigned smm_row_umax16(unsigned x)
{
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0"
                 : "+v"(x));
    return x;
}

// Max of a double over each row of 16 lanes (no NaNs); every lane of the row receives it.  fp64 has no DPP
// form, so the reduction runs on order-preserving 32-bit keys: the high word first, then the low word among
// the lanes that tie on the high word.
__device__ __forceinline__ double smm_row_max16(double x)
{
    const int hi = __double2hiint(x), lo = __double2loint(x);
    const int sgn = hi >> 31;                                  // all ones for negative values
    const unsigned khi = (unsigned)(hi ^ (sgn | (int)0x80000000));
    const unsigned klo = (unsigned)(lo ^ sgn);
    const unsigned mhi = smm_row_umax16(khi);
    const unsigned mlo = smm_row_umax16(khi == mhi ? klo : 0u);
    const int neg = (mhi & 0x80000000u) ? 0 : -1;              // key without the top bit <=> negative value
    return smm_pack((int)mlo ^ neg, (int)(mhi ^ ((unsigned)neg | 0x80000000u)));
}
