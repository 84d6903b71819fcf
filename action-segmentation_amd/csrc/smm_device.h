// smm_device.h -- shared device-side helpers and launch metadata for libsmmdp (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SMM_BIG_NEG (-1e9)   // reference BIG_NEG, semimarkov_modules.py:20
#define SMM_NEG_INF (-__builtin_huge_val())
#define SMM_MAX_STATES_DEV 32

// One entry per video, built on the host by smm_plan() and staged into the workspace.
struct SmmVideo {
    int64_t frame_off;   // first frame on the packed frame axis
    int64_t hist_off;    // offset (in doubles) of this video's history block in the workspace
    int32_t T;           // frames
    int32_t group;       // parameter group
    int32_t kp;          // usable segment lengths are 1..kp-1   (min(K, Tmax of the reference batch))
    int32_t pad;
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
    double *hist;              // per video: cum[c_max][T+1] then h[c_max][T+1]
    int64_t *spans;            // [b][t_max+1] or null
    int64_t *labels;           // [total_frames] or null
    double *best;              // [b] or null
    int32_t *n_segs;           // [b] or null
    int32_t *err;              // [1] sticky error flag (NaN in the inputs)
    int32_t c_max, k_rows, t_max, b;
};

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
#define SMM_DPP_WAVE_ROR1 0x13C
#define SMM_DPP_ROW_BCAST15 0x142
#define SMM_DPP_ROW_BCAST31 0x143

// lane i <- lane i-1, lane 0 <- lane 63
__device__ __forceinline__ double smm_wave_ror1(double x) { return smm_dpp<SMM_DPP_WAVE_ROR1>(x); }

// max over lanes 0..31 of x (lanes 32..63 ignored); result is wave-uniform.
__device__ __forceinline__ double smm_wave_max32(double x)
{
    x = fmax(x, smm_dpp<SMM_DPP_ROW_SHR(1)>(x));
    x = fmax(x, smm_dpp<SMM_DPP_ROW_SHR(2)>(x));
    x = fmax(x, smm_dpp<SMM_DPP_ROW_SHR(4)>(x));
    x = fmax(x, smm_dpp<SMM_DPP_ROW_SHR(8)>(x));            // lane 15 / 31 hold their row's max
    x = fmax(x, smm_dpp<SMM_DPP_ROW_BCAST15, 0xA>(x));      // row 1 (and 3) <- lane 15 of the row before
    return smm_readlane(x, 31);
}
