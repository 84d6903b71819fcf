// smm_emission.hip -- diagonal-Gaussian emission scorer for gfx950 (fp64 FMA on the VALU, HBM-streaming).
//
// Replaces SemiMarkovModule.emission_log_probs / _emission_log_probs_with_means (reference
// semimarkov_modules.py:324-381): elp[t][c] = log N(x_t; mu_c, diag(sigma^2)) (+ constraints[t][c], :379-380),
// evaluated in the expanded form
//     elp[t][c] = cst[c] + sum_d x[t][d] * w[d][c] - 0.5 * sum_d x[t][d]^2 * inv_var[d]
// (w = mu/sigma^2, cst = -0.5 sum mu^2/sigma^2 - sum log sigma - D/2 log 2pi; host-side, fp64), which is one
// v_fma_f64 per (frame, d, state).  In fp64 the expansion costs nothing in accuracy (|terms| ~ 1e3, eps 1e-16).
//
// Mapping: see the comment above smm_emission_kernel (fp64 MFMA tiles, weights in LDS, 64-B row pieces of x).
#include <cstdlib>
#include "smm_launch.h"

// ---------------------------------------------------------------------------------------------------------------
// elp = x . w is a [frames x D] x [D x C] product: it runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64: a
// 16-frame x 16-state tile per instruction, 4 features deep).  fp64 MFMA has the fp64 VALU's FLOP rate on gfx950, but
// one instruction replaces 1024 FMAs, needs no scalar operand stream, and leaves the VALU free for the x^2 term.
//   A operand (lane l): x[frame f0 + (l & 15)][feature]  -- each lane loads 16 B = 4 consecutive features of its frame,
//       so a wave reads 16 rows x 64 contiguous bytes per macro-step of 16 features (whole 64-B sectors, once);
//       the 4 features of a lane feed 4 MFMAs (k index = l >> 4), i.e. MFMA j of a macro-step contracts features
//       {d0 + 4k + j}.  Any feature <-> (j, k) assignment is valid as long as B uses the same one.
//   B operand (lane l): w[d0 + 4 (l >> 4) + j][s0 + (l & 15)]  from an LDS copy of the group's weight table.
//   C/D (lane l, reg i): frame f0 + (l >> 4) + 4 i, state s0 + (l & 15)  -> 128-B contiguous row pieces on store.
// One workgroup = 8 waves on one video sharing the LDS copy of the weights (2 workgroups per CU at D = 200, 32 states:
// 4 waves per SIMD); every wave walks 16-frame tiles with a grid stride.  The kernel is bound by HBM latency unless
// enough bytes are in flight: x is fetched in CHUNKS of 4 macro-steps (4 KB per wave) through a 3-deep register
// pipeline that runs across tile boundaries -- two chunks are always in flight while the third feeds the MFMAs.
typedef double smm_d4 __attribute__((ext_vector_type(4)));
// LDS row stride of the weight table (doubles) by the launch's 4-state groups behind the first 16 states: the smallest
// stride == 4 (mod 8) that holds the 16 + 4 ng columns -- 4 rows apart (the two k groups of a ds_read_b64 half-wave) are then
// 128 B apart modulo the 256-B bank span.  (Round 5: it was 36 for every class set above 16 states; at D = 300 that is 90 KB
// and ONE workgroup per CU -- 3.0 TB/s at 23 states where D = 256 runs at 3.7, profiles/round5_emission_d.txt; CrossTask's
// largest tasks have 23 states: stride 28 = 70 KB leaves two.)
constexpr int smm_em_row_stride(int ng) { return ng <= 1 ? 20 : (ng <= 3 ? 28 : 36); }
typedef float smm_f4 __attribute__((ext_vector_type(4)));

#define SMM_EM_WAVES 8
#ifndef SMM_EM_PAIR
#define SMM_EM_PAIR 1            // tiles in pairs per wave (smm_emission_pair_kernel); 0: one at a time (A/B aid)
#endif
#ifndef SMM_EM_ABLATE
#define SMM_EM_ABLATE 0           // development builds only (results WRONG, timing experiments): bit 0 no x^2 term,
                                  // bit 1 no loads of x, bit 2 no MFMA (one VALU add each instead), bit 3 no 4x4x4 MFMAs
#endif
#ifndef SMM_EM_TILES_PER_WAVE
#define SMM_EM_TILES_PER_WAVE 4   // same-box A/B on cfg3 (scripts/ab_prof.sh): 2: 0.611 ms, 3: 0.598, 4: 0.593, 8: 0.601, 16: 0.64
#endif

// largest i in [0, n) with cum[i] <= b   (cum[0] = 0, cum ascending, cum[n] = grid size)
__device__ __forceinline__ int smm_em_find_video(const int32_t *__restrict__ cum, int n, int b)
{
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] <= b) lo = mid; else hi = mid;
    }
    return lo;
}

// NT   state tiles of 16 (1: C <= 16, 2: C <= 32)        VEC  D % 4 == 0: 16-byte loads of x
// CONS narration constraints are added (they travel through the same pipeline as x: a load in the epilogue would make
//      the wave wait for every x load in flight)
// NG  (NT = 2) 4-state groups behind the first 16 states that the launch's largest class set needs: 1..4
template <int NT, bool VEC, bool CONS, int NG>
__global__ void __launch_bounds__(SMM_EM_WAVES * 64)
smm_emission_kernel(const SmmVideo *__restrict__ videos, const int32_t *__restrict__ order, const int32_t *__restrict__ n_states,
                    const float *__restrict__ xall, const double *__restrict__ wall, const double *__restrict__ cstall,
                    const double *__restrict__ iv, const float *__restrict__ cons, double *__restrict__ elp64,
                    float *__restrict__ elp32, int D, int cm, int tpw, const int32_t *__restrict__ blk_cum, int nvid, int blk_base)
{
    // LDS: this group's weights (zero padded), row d = w[d][0 .. 16 NT), then inv_var[D16].  Row stride 20 (NT = 1) /
    // 36 (NT = 2) doubles: rows 4 apart -- the two k groups of a ds_read_b64 half-wave -- land 128 B apart modulo the
    // 256-B bank span and do not collide.
    // NT = 2 (17..32 states): states 0..15 are one 16x16x4 tile as for NT = 1; states 16.. go through
    // v_mfma_f64_4x4x4_4b_f64 in groups of FOUR: its four blocks are the four 4-frame groups of the tile, all with the
    // same B (A[i][k] of block b: lane 16 k + 4 b + i -- the lane <-> (frame, k) mapping of the A operand above, so the
    // x registers serve both forms; B[k][j]: lane 16 k + 4 b + j; D[i][j]: lane 16 i + 4 b + j).  16 cycles per group
    // and 4 features against 64 for a second 16-state tile (same FLOP rate at a quarter of the granularity,
    // profiles/round2_ubench_mfma_f64_4x4.txt): 21..24 states cost 96 cycles instead of 128, 17..20 cost 80.
    extern __shared__ __attribute__((aligned(16))) double wl[];
    // flat grid: blk_cum[i] = workgroups of the videos order[0..i) (longest videos first: no long workgroup starts late)
    // (blk_base: a launch may cover only the videos order[v0 .. v0 + nvid) -- blk_cum and order arrive offset by v0 and the
    // cumulative block counts stay absolute)
    const int bid = blockIdx.x + blk_base;
    const int slot = smm_em_find_video(blk_cum, nvid, bid);
    const int chunk = bid - blk_cum[slot];
    const int vid = order[slot];
    const SmmVideo mv = videos[vid];
    const int T = mv.T, g = mv.group;
    const int C = n_states[g];
    const int ntiles = (T + 15) >> 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // this video's share of the grid: ~tpw tiles per wave (so short videos do not fill LDS for one tile per wave)
    const int nbv = blk_cum[slot + 1] - blk_cum[slot];
    const int D16 = (D + 15) & ~15;
    constexpr int WS = smm_em_row_stride(NT == 2 ? NG : 0), NC = (NT == 2) ? 16 + 4 * NG : 16;   // LDS row stride, columns filled (doubles)
    const int nt = (NT == 2 && C > 16) ? 2 : 1;                      // more than the first 16-state tile?
    const int ng1 = (NT == 2 && C > 16) ? (C - 13) >> 2 : 0;         // groups of 4 states behind it: ceil((C - 16) / 4)
    double *ivl = wl + (size_t)D16 * WS;
    {
        const double *__restrict__ w = wall + (size_t)g * D * cm;
        for (int i = threadIdx.x; i < D16 * NC; i += SMM_EM_WAVES * 64) {
            const int d = i / NC, c = i - d * NC;
            wl[(size_t)d * WS + c] = (d < D && c < C) ? w[(size_t)d * cm + c] : 0.0;
        }
        for (int d = threadIdx.x; d < D16; d += SMM_EM_WAVES * 64) ivl[d] = (d < D) ? iv[d] : 0.0;
    }
    __syncthreads();
    const int fr = lane & 15, kq = lane >> 4;
    const int jj = lane & 3, row1 = (fr & 12) + kq;                  // 4x4x4 result: state 16 + 4 g + jj of frame row1
    const float *__restrict__ xv = xall + (size_t)mv.frame_off * D;
    const double *__restrict__ cst = cstall + (size_t)g * cm;

    const int tile0 = chunk * SMM_EM_WAVES + wave, tstride = nbv * SMM_EM_WAVES;
    if (tile0 >= ntiles) return;
    const int nmy = (ntiles - tile0 + tstride - 1) / tstride;        // tiles of this wave
    const int nms = D16 >> 4;                                        // macro-steps of 16 features per tile
    const int nch = (nms + 3) >> 2;                                  // chunks of 4 macro-steps per tile
    const int total = nmy * nch;                                     // chunk sequence of this wave

    // fetch side of the pipeline: chunk (lt, lc) = tile tile0 + lt * tstride, macro-steps 4 lc .. 4 lc + 3
    int lt = 0, lc = 0;
    // (branch-free on the vector path: tiles past the end re-read the last tile, macro-steps past the end re-read the
    // last macro-step -- cache hits that are never consumed -- so that the compiler can count the loads in flight
    // instead of waiting for all of them)
    auto fetch = [&](float4 (&buf)[4], float (&cb)[CONS ? 4 * NT : 1]) {
        const int ltc = lt < nmy ? lt : nmy - 1;
        const int f0 = (tile0 + ltc * tstride) << 4;
        const int f = (f0 + fr < T) ? f0 + fr : T - 1;               // clamp: rows past the end are computed, not stored
        const float *__restrict__ xrow = xv + (size_t)f * D;
        if constexpr (CONS) {
            // constraints of this chunk's tile in the accumulator layout (used by the tile's last chunk only)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ff = f0 + kq + 4 * i;
                const size_t rowo = (size_t)(mv.frame_off + (ff < T ? ff : T - 1)) * cm;
                cb[i] = cons[rowo + (fr < cm ? fr : cm - 1)];
            }
            if constexpr (NT == 2) {                                 // states 16 + 4 g + jj of frame row1 (4x4x4 result layout)
                const int ff = f0 + row1;
                const size_t rowo = (size_t)(mv.frame_off + (ff < T ? ff : T - 1)) * cm;
#pragma unroll
                for (int g4 = 0; g4 < NG; ++g4) {
                    const int c = 16 + 4 * g4 + jj;
                    cb[4 + g4] = cons[rowo + (c < cm ? c : cm - 1)];
                }
            }
        }
        if constexpr (VEC) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int ms = (4 * lc + m < nms) ? 4 * lc + m : nms - 1;
                int db = 16 * ms + 4 * kq;                           // this lane's 4 features of the macro-step
                db = db + 3 < D ? db : D - 4;                        // (the zero-padded weights ignore what is read there)
                if (SMM_EM_ABLATE & 2) buf[m] = make_float4((float)db, 1.f, 2.f, (float)f);
                else buf[m] = *reinterpret_cast<const float4 *>(xrow + db);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int db = 16 * (4 * lc + m) + 4 * kq;
                float4 r;
                r.x = (db + 0 < D) ? xrow[db + 0] : 0.f;
                r.y = (db + 1 < D) ? xrow[db + 1] : 0.f;
                r.z = (db + 2 < D) ? xrow[db + 2] : 0.f;
                r.w = (db + 3 < D) ? xrow[db + 3] : 0.f;
                buf[m] = r;
            }
        }
        if (++lc == nch) { lc = 0; ++lt; }
    };

    // compute side
    int ct = 0, cc = 0;
    smm_d4 acc = (smm_d4){0.0, 0.0, 0.0, 0.0};
    constexpr int G1 = (NT == 2) ? NG : 1;                           // (NT = 1: unused dummies)
    double acc1[G1];
#pragma unroll
    for (int g4 = 0; g4 < G1; ++g4) acc1[g4] = 0.0;
    double q = 0.0;
    const double cstv = (fr < C) ? cst[fr] : 0.0;
    double cst1[G1];
#pragma unroll
    for (int g4 = 0; g4 < G1; ++g4) cst1[g4] = (NT == 2 && 16 + 4 * g4 + jj < C) ? cst[16 + 4 * g4 + jj] : 0.0;
    // B operands of the 16x16x4 tile for one macro-step, d = 16 ms + 4 kq + j: w[d][fr].  They are read from LDS one
    // macro-step ahead of the MFMAs that use them (wcur lives across consume() calls and across tiles: the weights do
    // not depend on the tile), so an MFMA never waits for its own LDS read.  The operands of the 4x4x4 groups,
    // w[d][16 + 4 g + jj], are read at the top of their j step and used behind its 16x16x4 instruction (64 cycles of
    // cover) -- prefetching them a macro-step ahead as well would cost 16 NG registers and the second workgroup per CU.
    // inv_var[d] is read at the top of the macro-step and used after its MFMAs have been issued.
    double wcur[4];
    auto load_ops = [&](int ms, double (&wv)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[j] = wl[(size_t)(16 * ms + 4 * kq + j) * WS + fr];
    };
    load_ops(0, wcur);
    auto consume = [&](const float4 (&buf)[4], const float (&cb)[CONS ? 4 * NT : 1]) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int ms = 4 * cc + m;
            if (ms < nms) {
                double wnext[4], iv4[4];
                load_ops(ms + 1 < nms ? ms + 1 : 0, wnext);
#pragma unroll
                for (int j = 0; j < 4; ++j) iv4[j] = ivl[16 * ms + 4 * kq + j];
                double av[4];
                av[0] = (double)buf[m].x; av[1] = (double)buf[m].y; av[2] = (double)buf[m].z; av[3] = (double)buf[m].w;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double wg[G1];
                    if constexpr (NT == 2) {
#pragma unroll
                        for (int g4 = 0; g4 < NG; ++g4)
                            wg[g4] = wl[(size_t)(16 * ms + 4 * kq + j) * WS + 16 + 4 * (g4 < ng1 ? g4 : 0) + jj];
                    }
                    if (SMM_EM_ABLATE & 4) { acc[j] += av[j] + wcur[j]; if constexpr (NT == 2) acc1[0] += wg[0] + wg[NG - 1]; }
                    else {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j], wcur[j], acc, 0, 0, 0);
                    if constexpr (NT == 2) {
#pragma unroll
                        for (int g4 = 0; g4 < NG; ++g4)
                            if (g4 < ng1 && !(SMM_EM_ABLATE & 8)) acc1[g4] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[j], wg[g4], acc1[g4], 0, 0, 0);
                    }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!(SMM_EM_ABLATE & 1)) q = fma(av[j] * iv4[j], av[j], q);
                    wcur[j] = wnext[j];
                }
            }
        }
        if (++cc < nch) return;
        // tile finished.  q: this lane summed the features with k index kq of frame fr; add the four k groups
        const int f0 = (tile0 + ct * tstride) << 4;
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = kq + 4 * i;                              // frame of accumulator register i
            const double qr = __shfl(q, row);
            const int ff = f0 + row;
            if (ff < T && fr < C) {
                const size_t o = (size_t)(mv.frame_off + ff) * cm + fr;
                double v = (cstv + acc[i]) - 0.5 * qr;
                if constexpr (CONS) v += (double)cb[i];
                if (elp64) elp64[o] = v;
                if (elp32) elp32[o] = (float)v;
            }
        }
        if constexpr (NT == 2) {                                     // 4x4x4 groups: frame row1, states 16 + 4 g + jj
            const double qr = __shfl(q, row1);
            const int ff = f0 + row1;
#pragma unroll
            for (int g4 = 0; g4 < NG; ++g4) {
                const int c = 16 + 4 * g4 + jj;
                if (g4 < ng1 && ff < T && c < C) {
                    const size_t o = (size_t)(mv.frame_off + ff) * cm + c;
                    double v = (cst1[g4] + acc1[g4]) - 0.5 * qr;
                    if constexpr (CONS) v += (double)cb[4 + g4];
                    if (elp64) elp64[o] = v;
                    if (elp32) elp32[o] = (float)v;
                }
                acc1[g4] = 0.0;
            }
        }
        acc = (smm_d4){0.0, 0.0, 0.0, 0.0};
        q = 0.0;
        cc = 0;
        ++ct;
    };

    // two chunk buffers: one in flight while the other feeds the MFMAs.  (A third -- the 4x4x4 form left the registers for
    // it -- measured 1 % SLOWER on the same box, cfg2 and cfg3: two chunks per wave x 16 waves per CU already cover the
    // latency; the extra prologue fetch only delays the first MFMA.)
    float4 b0[4], b1[4];
    float c0[CONS ? 4 * NT : 1], c1[CONS ? 4 * NT : 1];
    fetch(b0, c0);
    for (int it = 0; it < total; it += 2) {
        fetch(b1, c1);
        consume(b0, c0);
        if (it + 1 >= total) break;
        fetch(b0, c0);
        consume(b1, c1);
    }
}

// The same kernel on PAIRS of tiles (round 4).  Template arguments as above.
// CONS narration constraints are added (they travel through the same pipeline as x: a load in the epilogue would make
//      the wave wait for every x load in flight)
// NG  (NT = 2) 4-state groups behind the first 16 states that the launch's largest class set needs: 1..4
template <int NT, bool VEC, bool CONS, int NG>
__global__ void __launch_bounds__(SMM_EM_WAVES * 64)
smm_emission_pair_kernel(const SmmVideo *__restrict__ videos, const int32_t *__restrict__ order, const int32_t *__restrict__ n_states,
                    const float *__restrict__ xall, const double *__restrict__ wall, const double *__restrict__ cstall,
                    const double *__restrict__ iv, const float *__restrict__ cons, double *__restrict__ elp64,
                    float *__restrict__ elp32, int D, int cm, int tpw, const int32_t *__restrict__ blk_cum, int nvid, int blk_base)
{
    // LDS: this group's weights (zero padded), row d = w[d][0 .. 16 NT), then inv_var[D16].  Row stride 20 (NT = 1) /
    // 36 (NT = 2) doubles: rows 4 apart -- the two k groups of a ds_read_b64 half-wave -- land 128 B apart modulo the
    // 256-B bank span and do not collide.
    // NT = 2 (17..32 states): states 0..15 are one 16x16x4 tile as for NT = 1; states 16.. go through
    // v_mfma_f64_4x4x4_4b_f64 in groups of FOUR: its four blocks are the four 4-frame groups of the tile, all with the
    // same B (A[i][k] of block b: lane 16 k + 4 b + i -- the lane <-> (frame, k) mapping of the A operand above, so the
    // x registers serve both forms; B[k][j]: lane 16 k + 4 b + j; D[i][j]: lane 16 i + 4 b + j).  16 cycles per group
    // and 4 features against 64 for a second 16-state tile (same FLOP rate at a quarter of the granularity,
    // profiles/round2_ubench_mfma_f64_4x4.txt): 21..24 states cost 96 cycles instead of 128, 17..20 cost 80.
    extern __shared__ __attribute__((aligned(16))) double wl[];
    // flat grid: blk_cum[i] = workgroups of the videos order[0..i) (longest videos first: no long workgroup starts late)
    // (blk_base: a launch may cover only the videos order[v0 .. v0 + nvid) -- blk_cum and order arrive offset by v0 and the
    // cumulative block counts stay absolute)
    const int bid = blockIdx.x + blk_base;
    const int slot = smm_em_find_video(blk_cum, nvid, bid);
    const int chunk = bid - blk_cum[slot];
    const int vid = order[slot];
    const SmmVideo mv = videos[vid];
    const int T = mv.T, g = mv.group;
    const int C = n_states[g];
    const int ntiles = (T + 15) >> 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // this video's share of the grid: ~tpw tiles per wave (so short videos do not fill LDS for one tile per wave)
    const int nbv = blk_cum[slot + 1] - blk_cum[slot];
    const int D16 = (D + 15) & ~15;
    constexpr int WS = smm_em_row_stride(NT == 2 ? NG : 0), NC = (NT == 2) ? 16 + 4 * NG : 16;   // LDS row stride, columns filled (doubles)
    const int nt = (NT == 2 && C > 16) ? 2 : 1;                      // more than the first 16-state tile?
    const int ng1 = (NT == 2 && C > 16) ? (C - 13) >> 2 : 0;         // groups of 4 states behind it: ceil((C - 16) / 4)
    double *ivl = wl + (size_t)D16 * WS;
    {
        const double *__restrict__ w = wall + (size_t)g * D * cm;
        for (int i = threadIdx.x; i < D16 * NC; i += SMM_EM_WAVES * 64) {
            const int d = i / NC, c = i - d * NC;
            wl[(size_t)d * WS + c] = (d < D && c < C) ? w[(size_t)d * cm + c] : 0.0;
        }
        for (int d = threadIdx.x; d < D16; d += SMM_EM_WAVES * 64) ivl[d] = (d < D) ? iv[d] : 0.0;
    }
    __syncthreads();
    const int fr = lane & 15, kq = lane >> 4;
    const int jj = lane & 3, row1 = (fr & 12) + kq;                  // 4x4x4 result: state 16 + 4 g + jj of frame row1
    const float *__restrict__ xv = xall + (size_t)mv.frame_off * D;
    const double *__restrict__ cst = cstall + (size_t)g * cm;

    // PAIRS of tiles (32 consecutive frames): the B operands of a macro-step -- 4 + 4 NG ds_read_b64 per wave, the kernel's
    // LDS traffic -- are read once and feed the MFMAs of both tiles
    const int npairs = (ntiles + 1) >> 1;
    const int tile0 = chunk * SMM_EM_WAVES + wave, tstride = nbv * SMM_EM_WAVES;   // (in pairs)
    if (tile0 >= npairs) return;
    const int nmy = (npairs - tile0 + tstride - 1) / tstride;        // pairs of this wave
    const int nms = D16 >> 4;                                        // macro-steps of 16 features per tile
    constexpr int MC = 2;                                            // macro-steps per chunk (x 2 tiles: 4 KB per wave, as before)
    const int nch = (nms + MC - 1) / MC;                             // chunks per pair
    const int total = nmy * nch;                                     // chunk sequence of this wave

    // fetch side of the pipeline: chunk (lt, lc) = tile tile0 + lt * tstride, macro-steps 4 lc .. 4 lc + 3
    int lt = 0, lc = 0;
    // (branch-free on the vector path: tiles past the end re-read the last tile, macro-steps past the end re-read the
    // last macro-step -- cache hits that are never consumed -- so that the compiler can count the loads in flight
    // instead of waiting for all of them)
    auto fetch = [&](float4 (&buf)[2][MC], float (&cb)[2][CONS ? 4 * NT : 1]) {
        const int ltc = lt < nmy ? lt : nmy - 1;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int f0 = (2 * (tile0 + ltc * tstride) + tl) << 4;
            const int f = (f0 + fr < T) ? f0 + fr : T - 1;           // clamp: rows past the end are computed, not stored
            const float *__restrict__ xrow = xv + (size_t)f * D;
            if constexpr (CONS) {
                // constraints of this chunk's tile in the accumulator layout (used by the pair's last chunk only)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ff = f0 + kq + 4 * i;
                    const size_t rowo = (size_t)(mv.frame_off + (ff < T ? ff : T - 1)) * cm;
                    cb[tl][i] = cons[rowo + (fr < cm ? fr : cm - 1)];
                }
                if constexpr (NT == 2) {                             // states 16 + 4 g + jj of frame row1 (4x4x4 result layout)
                    const int ff = f0 + row1;
                    const size_t rowo = (size_t)(mv.frame_off + (ff < T ? ff : T - 1)) * cm;
#pragma unroll
                    for (int g4 = 0; g4 < NG; ++g4) {
                        const int c = 16 + 4 * g4 + jj;
                        cb[tl][4 + g4] = cons[rowo + (c < cm ? c : cm - 1)];
                    }
                }
            }
            if constexpr (VEC) {
#pragma unroll
                for (int m = 0; m < MC; ++m) {
                    const int ms = (MC * lc + m < nms) ? MC * lc + m : nms - 1;
                    int db = 16 * ms + 4 * kq;                       // this lane's 4 features of the macro-step
                    db = db + 3 < D ? db : D - 4;                    // (the zero-padded weights ignore what is read there)
                    buf[tl][m] = *reinterpret_cast<const float4 *>(xrow + db);
                }
            } else {
#pragma unroll
                for (int m = 0; m < MC; ++m) {
                    const int db = 16 * (MC * lc + m) + 4 * kq;
                    float4 r;
                    r.x = (db + 0 < D) ? xrow[db + 0] : 0.f;
                    r.y = (db + 1 < D) ? xrow[db + 1] : 0.f;
                    r.z = (db + 2 < D) ? xrow[db + 2] : 0.f;
                    r.w = (db + 3 < D) ? xrow[db + 3] : 0.f;
                    buf[tl][m] = r;
                }
            }
        }
        if (++lc == nch) { lc = 0; ++lt; }
    };

    // compute side
    int ct = 0, cc = 0;
    smm_d4 acc[2] = {(smm_d4){0.0, 0.0, 0.0, 0.0}, (smm_d4){0.0, 0.0, 0.0, 0.0}};
    constexpr int G1 = (NT == 2) ? NG : 1;                           // (NT = 1: unused dummies)
    double acc1[2][G1];
#pragma unroll
    for (int g4 = 0; g4 < G1; ++g4) { acc1[0][g4] = 0.0; acc1[1][g4] = 0.0; }
    double q[2] = {0.0, 0.0};
    const double cstv = (fr < C) ? cst[fr] : 0.0;
    double cst1[G1];
#pragma unroll
    for (int g4 = 0; g4 < G1; ++g4) cst1[g4] = (NT == 2 && 16 + 4 * g4 + jj < C) ? cst[16 + 4 * g4 + jj] : 0.0;
    // B operands: as in smm_emission_kernel (the 16x16x4 tile's a macro-step ahead, the 4x4x4 groups' at the top of their
    // j step), each read serving the MFMAs of BOTH tiles of the pair
    double wcur[4];
    auto load_ops = [&](int ms, double (&wv)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[j] = wl[(size_t)(16 * ms + 4 * kq + j) * WS + fr];
    };
    load_ops(0, wcur);
    auto consume = [&](const float4 (&buf)[2][MC], const float (&cb)[2][CONS ? 4 * NT : 1]) {
#pragma unroll
        for (int m = 0; m < MC; ++m) {
            const int ms = MC * cc + m;
            if (ms < nms) {
                double wnext[4], iv4[4];
                load_ops(ms + 1 < nms ? ms + 1 : 0, wnext);
#pragma unroll
                for (int j = 0; j < 4; ++j) iv4[j] = ivl[16 * ms + 4 * kq + j];
                double av[2][4];
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    av[tl][0] = (double)buf[tl][m].x; av[tl][1] = (double)buf[tl][m].y;
                    av[tl][2] = (double)buf[tl][m].z; av[tl][3] = (double)buf[tl][m].w;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double wg[G1];
                    if constexpr (NT == 2) {
#pragma unroll
                        for (int g4 = 0; g4 < NG; ++g4)
                            wg[g4] = wl[(size_t)(16 * ms + 4 * kq + j) * WS + 16 + 4 * (g4 < ng1 ? g4 : 0) + jj];
                    }
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) {
                        acc[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[tl][j], wcur[j], acc[tl], 0, 0, 0);
                        if constexpr (NT == 2) {
#pragma unroll
                            for (int g4 = 0; g4 < NG; ++g4)
                                if (g4 < ng1) acc1[tl][g4] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[tl][j], wg[g4], acc1[tl][g4], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    q[0] = fma(av[0][j] * iv4[j], av[0][j], q[0]);
                    q[1] = fma(av[1][j] * iv4[j], av[1][j], q[1]);
                    wcur[j] = wnext[j];
                }
            }
        }
        if (++cc < nch) return;
        // pair finished.  q: this lane summed the features with k index kq of frame fr; add the four k groups
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int f0 = (2 * (tile0 + ct * tstride) + tl) << 4;
            double qq = q[tl];
            qq += __shfl_xor(qq, 16);
            qq += __shfl_xor(qq, 32);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = kq + 4 * i;                          // frame of accumulator register i
                const double qr = __shfl(qq, row);
                const int ff = f0 + row;
                if (ff < T && fr < C) {
                    const size_t o = (size_t)(mv.frame_off + ff) * cm + fr;
                    double v = (cstv + acc[tl][i]) - 0.5 * qr;
                    if constexpr (CONS) v += (double)cb[tl][i];
                    if (elp64) elp64[o] = v;
                    if (elp32) elp32[o] = (float)v;
                }
            }
            if constexpr (NT == 2) {                                 // 4x4x4 groups: frame row1, states 16 + 4 g + jj
                const double qr = __shfl(qq, row1);
                const int ff = f0 + row1;
#pragma unroll
                for (int g4 = 0; g4 < NG; ++g4) {
                    const int c = 16 + 4 * g4 + jj;
                    if (g4 < ng1 && ff < T && c < C) {
                        const size_t o = (size_t)(mv.frame_off + ff) * cm + c;
                        double v = (cst1[g4] + acc1[tl][g4]) - 0.5 * qr;
                        if constexpr (CONS) v += (double)cb[tl][4 + g4];
                        if (elp64) elp64[o] = v;
                        if (elp32) elp32[o] = (float)v;
                    }
                    acc1[tl][g4] = 0.0;
                }
            }
            acc[tl] = (smm_d4){0.0, 0.0, 0.0, 0.0};
            q[tl] = 0.0;
        }
        cc = 0;
        ++ct;
    };

    // two chunk buffers: one in flight while the other feeds the MFMAs.  (A third -- the 4x4x4 form left the registers for
    // it -- measured 1 % SLOWER on the same box, cfg2 and cfg3: two chunks per wave x 16 waves per CU already cover the
    // latency; the extra prologue fetch only delays the first MFMA.)
    float4 b0[2][MC], b1[2][MC];
    float c0[2][CONS ? 4 * NT : 1], c1[2][CONS ? 4 * NT : 1];
    fetch(b0, c0);
    for (int it = 0; it < total; it += 2) {
        fetch(b1, c1);
        consume(b0, c0);
        if (it + 1 >= total) break;
        fetch(b0, c0);
        consume(b1, c1);
    }
}

#ifdef SMM_DEV   // (development builds only: the variant lost its A/B; the product library does not carry it)
// ---------------------------------------------------------------------------------------------------------------
// Kernel v2 (D % 4 == 0): x goes through LDS in whole 128-byte lines.
//
// The MFMA A layout gives a row of x only 4 lanes, so loading it straight into A registers (the kernel above) touches
// half a 128-B line per row and request: 0.65 ms on cfg3 where the bytes alone need 0.38 ms.  Here a wave's 16-frame
// tile -- 16 * D * 4 CONTIGUOUS bytes of the packed frame axis -- is fetched with 16-byte-per-lane loads of 1 KB per
// instruction (every line whole, once), parked in a per-wave LDS buffer with a row stride == 8 (mod 16) floats (the
// A-operand ds_read_b128 of lane (frame l & 15, k l >> 4) is then conflict-free; D = 200 needs no padding at all), and
// read back in the A layout.  One tile per wave is in flight in registers while the previous one feeds the MFMAs.
// LDS: weights (D16 * 33 doubles) + 8 waves * 16 * RS floats = 55 + 102 KB at D = 200: one workgroup per CU, 100 KB of
// loads in flight per CU.
template <int NT, bool CONS, int NLD>
__global__ void __launch_bounds__(SMM_EM_WAVES * 64)
smm_emission_lds_kernel(const SmmVideo *__restrict__ videos, const int32_t *__restrict__ order, const int32_t *__restrict__ n_states,
                        const float *__restrict__ xall, const double *__restrict__ wall, const double *__restrict__ cstall,
                        const double *__restrict__ iv, const float *__restrict__ cons, double *__restrict__ elp64,
                        float *__restrict__ elp32, int D, int cm, int RS, int64_t x_floats,
                        const int32_t *__restrict__ blk_cum, int nvid, int blk_base)
{
    extern __shared__ __attribute__((aligned(16))) double wl[];
    // (blk_base: a launch may cover only the videos order[v0 .. v0 + nvid) -- blk_cum and order arrive offset by v0 and the
    // cumulative block counts stay absolute)
    const int bid = blockIdx.x + blk_base;
    const int slot = smm_em_find_video(blk_cum, nvid, bid);
    const int chunk = bid - blk_cum[slot];
    const int vid = order[slot];
    const SmmVideo mv = videos[vid];
    const int T = mv.T, g = mv.group;
    const int C = n_states[g];
    const int ntiles = (T + 15) >> 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nbv = blk_cum[slot + 1] - blk_cum[slot];
    const int D16 = (D + 15) & ~15;
    constexpr int WS = (NT == 2) ? 32 : 20;                          // LDS row stride of the weights (doubles)
    const int nt = (NT == 2 && C > 16) ? 2 : 1;
    double *ivl = wl + (size_t)D16 * WS;
    float *xt = reinterpret_cast<float *>(ivl + D16) + (size_t)wave * 16 * RS;   // this wave's tile: 16 rows x RS floats
    {
        const double *__restrict__ w = wall + (size_t)g * D * cm;
        for (int i = threadIdx.x; i < D16 * 16 * NT; i += SMM_EM_WAVES * 64) {
            const int d = i / (16 * NT), r = i - d * (16 * NT);
            const int c = (NT == 2) ? (r >> 1) + 16 * (r & 1) : r;
            wl[(size_t)d * WS + r] = (d < D && c < C) ? w[(size_t)d * cm + c] : 0.0;
        }
        for (int d = threadIdx.x; d < D16; d += SMM_EM_WAVES * 64) ivl[d] = (d < D) ? iv[d] : 0.0;
    }
    __syncthreads();
    const int fr = lane & 15, kq = lane >> 4;
    const double *__restrict__ cst = cstall + (size_t)g * cm;
    const int tile0 = chunk * SMM_EM_WAVES + wave, tstride = nbv * SMM_EM_WAVES;
    if (tile0 >= ntiles) return;
    const int nmy = (ntiles - tile0 + tstride - 1) / tstride;
    const int nms = D16 >> 4;
    // NLD: 16-byte loads per lane and tile, >= 16 D / 256 (the surplus reads the next tile's first bytes; never parked)
    smm_f4 pre[NLD];
    float cb[CONS ? 4 * NT : 1];
    const int64_t last4 = x_floats - 4;                              // clamp: never past the end of x

    // element j of this lane in a tile: linear float offset 4 * (64 j + lane) -> offset in the LDS image (-1: not in the
    // tile); computed once -- a division per element and tile would cost a third of the MFMA time
    int lo[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int e = 4 * (64 * j + lane);
        const int row = e / D, col = e - row * D;                    // D % 4 == 0: a 16-byte piece never straddles rows
        lo[j] = row < 16 ? row * RS + col : -1;
    }
    double cstv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) cstv[t] = (16 * t + fr < C) ? cst[16 * t + fr] : 0.0;
    auto load_ops = [&](int ms, double (&wv)[4][NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = 16 * ms + 4 * kq + j;
            if constexpr (NT == 2) {
                const double2 t2 = *reinterpret_cast<const double2 *>(&wl[(size_t)d * WS + 2 * fr]);
                wv[j][0] = t2.x;
                wv[j][NT - 1] = t2.y;
            } else {
                wv[j][0] = wl[(size_t)d * WS + fr];
            }
        }
    };
    // ct = -1: prologue (fetch tile 0 only); single call sites keep the tile registers out of scratch
    for (int ct = -1; ct < nmy; ++ct) {
        float cbt[CONS ? 4 * NT : 1];
        if constexpr (CONS) {
#pragma unroll
            for (int i = 0; i < 4 * NT; ++i) cbt[i] = cb[i];
        }
        if (ct >= 0) {
            // registers -> LDS image of the tile (waits for this tile's loads)
#pragma unroll
            for (int j = 0; j < NLD; ++j)
                if (lo[j] >= 0) *reinterpret_cast<smm_f4 *>(xt + lo[j]) = pre[j];
        }
        {
            // the next tile is in flight while this one is computed
            const int lt = ct + 1;
            const int ltc = lt < nmy ? lt : nmy - 1;
            const int f0n = (tile0 + ltc * tstride) << 4;
            const int64_t base = (int64_t)(mv.frame_off + f0n) * D;
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                int64_t e = base + 4 * (64 * j + lane);
                e = e < last4 ? e : last4;                           // (rows past the video / the buffer are never stored)
                pre[j] = *reinterpret_cast<const smm_f4 *>(xall + e);
            }
            if constexpr (CONS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ff = f0n + kq + 4 * i;
                    const size_t rowo = (size_t)(mv.frame_off + (ff < T ? ff : T - 1)) * cm;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int c = 16 * t + fr;
                        cb[4 * t + i] = cons[rowo + (c < cm ? c : cm - 1)];
                    }
                }
            }
        }
        if (ct < 0) continue;
        smm_d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (smm_d4){0.0, 0.0, 0.0, 0.0};
        double q = 0.0;
        double wcur[4][NT];
        load_ops(0, wcur);
        const float *xrow = xt + fr * RS;
        float4 a4 = *reinterpret_cast<const float4 *>(xrow + (4 * kq + 3 < D ? 4 * kq : D - 4));
        for (int ms = 0; ms < nms; ++ms) {
            double wnext[4][NT], iv4[4];
            load_ops(ms + 1 < nms ? ms + 1 : 0, wnext);
#pragma unroll
            for (int j = 0; j < 4; ++j) iv4[j] = ivl[16 * ms + 4 * kq + j];
            int cn = 16 * (ms + 1) + 4 * kq;                         // next macro-step's features of this lane
            cn = cn + 3 < D ? cn : D - 4;                            // (zero-padded weights ignore what is read there)
            const float4 an = *reinterpret_cast<const float4 *>(xrow + cn);
            double av[4];
            av[0] = (double)a4.x; av[1] = (double)a4.y; av[2] = (double)a4.z; av[3] = (double)a4.w;
            if (nt == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j], wcur[j][0], acc[0], 0, 0, 0);
                    acc[NT - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j], wcur[j][NT - 1], acc[NT - 1], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j], wcur[j][0], acc[0], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                q = fma(av[j] * iv4[j], av[j], q);
#pragma unroll
                for (int t = 0; t < NT; ++t) wcur[j][t] = wnext[j][t];
            }
            a4 = an;
        }
        // tile finished.  q: this lane summed the features with k index kq of frame fr; add the four k groups
        const int f0 = (tile0 + ct * tstride) << 4;
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = kq + 4 * i;
            const double qr = __shfl(q, row);
            const int ff = f0 + row;
            if (ff < T) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int c = 16 * t + fr;
                    if (c < C) {
                        const size_t o = (size_t)(mv.frame_off + ff) * cm + c;
                        double v = (cstv[t] + acc[t][i]) - 0.5 * qr;
                        if constexpr (CONS) v += (double)cbt[4 * t + i];
                        if (elp64) elp64[o] = v;
                        if (elp32) elp32[o] = (float)v;
                    }
                }
            }
        }
    }
}
#endif   // SMM_DEV

// fp32 -> fp64 widening of a [n] array (smm_viterbi_f32 boundary)
__global__ void smm_widen_kernel(const float *src, double *dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (double)src[i];
}

int smm_emission_tiles_per_wave(int64_t total_frames, int b)
{
    // up to SMM_EM_TILES_PER_WAVE tiles per wave (amortises the LDS fill of the weights), fewer when that would leave CUs without a
    // workgroup: aim at >= 2 workgroups on each of the 256 CUs
    int64_t tpw = (total_frames / 16 + b) / (512 * SMM_EM_WAVES);
    return (int)(tpw < 1 ? 1 : (tpw > SMM_EM_TILES_PER_WAVE ? SMM_EM_TILES_PER_WAVE : tpw));
}

size_t smm_emission_lds_bytes(int d, int c_need)
{
    const int d16 = (d + 15) & ~15, ng = c_need <= 16 ? 0 : (c_need - 13) >> 2;
    return sizeof(double) * d16 * (smm_em_row_stride(ng) + 1);
}

int smm_emission_blocks(int t, int tpw)
{
    const int tiles = (t + 15) / 16;
    return (tiles + SMM_EM_WAVES * tpw - 1) / (SMM_EM_WAVES * tpw);
}

void smm_launch_emission(const SmmEmArgs &a, int ct, int tpw, int n_blocks, const int32_t *blk_cum, int64_t total_frames,
                         hipStream_t stream, int blk_base, int vid0, int nvid)
{
    // blk_base / vid0 / nvid: only the videos order[vid0 .. vid0 + nvid) = the workgroups blk_base .. blk_base + n_blocks
    // of the flat grid (nvid < 0: all of them)
    if (nvid < 0) { vid0 = 0; nvid = a.b; blk_base = 0; }
    if (n_blocks <= 0 || nvid <= 0) return;
    const int32_t *order_v = a.order + vid0;
    blk_cum += vid0;
    const int d16 = (a.d + 15) & ~15;
    dim3 grid(n_blocks), block(SMM_EM_WAVES * 64);
    const size_t lds_w = smm_emission_lds_bytes(a.d, ct);                      // weights (row stride 20 / 28 / 36) + inv_var
    const size_t lds_w2 = sizeof(double) * d16 * (ct <= 16 ? 21 : 33);       // (v2's layout)
    const bool vec = (a.d & 3) == 0;
    // v2 (x staged through LDS in whole lines): row stride == 8 (mod 16) floats, >= D
    int rs = (a.d / 16) * 16 + 8;
    if (rs < a.d) rs += 16;
    const int nld = (16 * a.d / 4 + 63) / 64;                                 // 16-byte loads per lane and tile
    const size_t lds_x = sizeof(float) * 16 * rs * SMM_EM_WAVES;
    // v2 is opt-in (SMM_EMISSION_V2=1): measured on cfg3 (rocprofv3, profiles/round2_emission_v1_v2.txt) it takes 0.77 ms
    // against v1's 0.62 ms -- its LDS footprint (weights + 8 tile buffers = 157 KB) leaves ONE 8-wave workgroup per CU
    // where v1 runs two, and the kernel turned out not to be bound by the 64-byte row pieces of v1's loads.
    const bool v2 = vec && nld <= 20 && lds_w2 + lds_x <= 160 * 1024 && smm_env_emission_v2();
    const int nsel = nld <= 4 ? 0 : (nld <= 8 ? 1 : (nld <= 13 ? 2 : 3));     // compiled load counts: 4, 8, 13, 20
    auto go1 = [&](auto kern) {
        if (lds_w > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w);
        hipLaunchKernelGGL(kern, grid, block, lds_w, stream, a.videos, order_v, a.n_states, a.x, a.w, a.cst, a.inv_var, a.cons,
                           a.elp64, a.elp32, a.d, a.c_max, tpw, blk_cum, nvid, blk_base);
    };
    auto go2 = [&](auto kern) {
        const size_t lds = lds_w2 + lds_x;
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, block, lds, stream, a.videos, order_v, a.n_states, a.x, a.w, a.cst, a.inv_var, a.cons,
                           a.elp64, a.elp32, a.d, a.c_max, rs, (int64_t)total_frames * a.d, blk_cum, nvid, blk_base);
    };
#ifdef SMM_DEV
    if (v2) {
#define SMM_EM_V2(NLD_)                                                                         \
        switch ((ct <= 16 ? 0 : 2) + (a.cons ? 1 : 0)) {                                        \
        case 0: go2(smm_emission_lds_kernel<1, false, NLD_>); break;                            \
        case 1: go2(smm_emission_lds_kernel<1, true, NLD_>); break;                             \
        case 2: go2(smm_emission_lds_kernel<2, false, NLD_>); break;                            \
        default: go2(smm_emission_lds_kernel<2, true, NLD_>); break;                            \
        }
        switch (nsel) {
        case 0: SMM_EM_V2(4) break;
        case 1: SMM_EM_V2(8) break;
        case 2: SMM_EM_V2(13) break;
        default: SMM_EM_V2(20) break;
        }
#undef SMM_EM_V2
        return;
    }
#else
    (void)v2; (void)nsel; (void)go2;
#endif
    const int ng = ct <= 16 ? 0 : (ct - 13) >> 2;                            // 4-state groups behind the first 16 states
    // tiles in PAIRS (each read of the weights from LDS feeds two tiles' MFMAs: cfg3 0.645 -> 0.584 ms, bit-identical) where the
    // second tile's registers leave 4 waves per SIMD (<= 128 VGPRs: 16-byte loads, <= 28 states, constraints only up to 16
    // states) and the launch is large enough for >= 2 tiles per wave (a small one loses parallelism to pairs: cfg4)
    const bool pair = SMM_EM_PAIR && vec && tpw >= 2 && (a.cons ? ng == 0 : ng <= 3);
#define SMM_EM_V1(NG_)                                                                          \
    switch ((vec ? 2 : 0) + (a.cons ? 1 : 0)) {                                                 \
    case 0: go1(smm_emission_kernel<2, false, false, NG_>); break;                             \
    case 1: go1(smm_emission_kernel<2, false, true, NG_>); break;                              \
    case 2: if (pair) go1(smm_emission_pair_kernel<2, true, false, (NG_ <= 3 ? NG_ : 3)>);     \
            else go1(smm_emission_kernel<2, true, false, NG_>);                                \
            break;                                                                             \
    default: go1(smm_emission_kernel<2, true, true, NG_>); break;                              \
    }
    switch (ng) {
    case 0:
        switch ((vec ? 2 : 0) + (a.cons ? 1 : 0)) {
        case 0: go1(smm_emission_kernel<1, false, false, 0>); break;
        case 1: go1(smm_emission_kernel<1, false, true, 0>); break;
        case 2: if (pair) go1(smm_emission_pair_kernel<1, true, false, 0>); else go1(smm_emission_kernel<1, true, false, 0>); break;
        default: if (pair) go1(smm_emission_pair_kernel<1, true, true, 0>); else go1(smm_emission_kernel<1, true, true, 0>); break;
        }
        break;
    case 1: SMM_EM_V1(1) break;
    case 2: SMM_EM_V1(2) break;
    case 3: SMM_EM_V1(3) break;
    default: SMM_EM_V1(4) break;
    }
#undef SMM_EM_V1
}

// ------------------------------------------------------------------------------------------------ chain rule (training)
// elp[t][c] = cst[g][c] + sum_d x[t][d] w[g][d][c] - 0.5 sum_d x[t][d]^2 inv_var[d]   ==>   with ge = dL/d elp:
//   g_w[g][c][d] = sum_t x[t][d] ge[t][c]      g_cst[g][c] = sum_t ge[t][c]      g_iv[d] = -0.5 sum_t x[t][d]^2 sum_c ge[t][c]
// (t over the frames of the videos of group g).  Replaces what autograd does behind the reference's
// emission_log_probs (semimarkov_modules.py:324-381) in loss.backward() (semimarkov.py:286); it was six skinny fp64
// torch GEMMs (K = frames, M x N = D x C) plus fp64 copies of x -- 1.5 ms of a 4 ms training step on cfg4.
// Thread = feature column (a wave's loads of one frame are 256 contiguous bytes), C fp64 accumulators in registers; the
// ge rows of a 64-frame tile sit in LDS and are read as broadcasts.  Column D is the constant-1 feature: its sums are
// g_cst.  A workgroup owns a contiguous range of 256-frame chunks and leaves through atomics when the group changes
// (consecutive lanes = consecutive d of one class row: 512 contiguous bytes per atomic instruction).
#define SMM_EB_CHUNK 256
#define SMM_EB_TILE 64

int smm_emission_bwd_chunk() { return SMM_EB_CHUNK; }

template <int CT>
__global__ void __launch_bounds__(256) smm_emission_bwd_kernel(SmmEmBwdArgs a)
{
    __shared__ __attribute__((aligned(16))) double s_ge[SMM_EB_TILE * CT];
    __shared__ double s_rs[SMM_EB_TILE];
    const int tid = threadIdx.x;
    const int D = a.d, cm = a.c_max;
    const int ipw = (a.n_chunks + (int)gridDim.x - 1) / (int)gridDim.x;
    const int it0 = blockIdx.x * ipw, it1 = min(a.n_chunks, it0 + ipw);
    for (int c0 = 0; c0 <= D; c0 += 256) {                       // column passes (one for D < 256)
        const int col = c0 + tid;
        const bool real = col < D, ones = col == D;
        const int colc = real ? col : 0;                         // (idle lanes read column 0 and drop the result)
        double acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = 0.0;
        double ivacc = 0.0;
        int cur_g = -1, cur_c = 0;
        auto flush = [&]() __attribute__((always_inline)) {
            if (cur_g >= 0 && (real || ones)) {
                double *dst = real ? a.g_w + (size_t)cur_g * cm * D + col : a.g_cst + (size_t)cur_g * cm;
                const size_t step = real ? (size_t)D : 1;
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    if (c < cur_c) __hip_atomic_fetch_add(dst + c * step, acc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = 0.0;
        };
        for (int it = it0; it < it1; ++it) {
            const int slot = smm_em_find_video(a.cum, a.b, it);
            const SmmVideo mv = a.videos[a.order[slot]];
            if (mv.group != cur_g) {
                flush();
                cur_g = mv.group;
                cur_c = min(a.n_states[cur_g], CT);
            }
            const int r0 = (it - a.cum[slot]) * SMM_EB_CHUNK, r1 = min(mv.T, r0 + SMM_EB_CHUNK);
            for (int t0 = r0; t0 < r1; t0 += SMM_EB_TILE) {
                const int nt = min(SMM_EB_TILE, r1 - t0);
                const double *__restrict__ ge = a.g_elp + (size_t)(mv.frame_off + t0) * cm;
                const float *__restrict__ xp = a.x + (size_t)(mv.frame_off + t0) * D + colc;
                __syncthreads();                                 // the previous tile has been consumed
                for (int i = tid; i < SMM_EB_TILE * CT; i += 256) {
                    const int f = i / CT, c = i - f * CT;
                    s_ge[i] = (f < nt && c < cur_c) ? ge[(size_t)f * cm + c] : 0.0;
                }
                __syncthreads();
                if (tid < SMM_EB_TILE) {
                    double r = 0.0;
#pragma unroll
                    for (int c = 0; c < CT; ++c) r += s_ge[tid * CT + ((c + tid) % CT)];   // (rotated: spreads the banks)
                    s_rs[tid] = r;
                }
                __syncthreads();
                for (int f = 0; f < nt; f += 8) {
                    float xv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) xv[u] = xp[(size_t)min(f + u, nt - 1) * D];   // rows past nt: ge row is 0
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double xd = ones ? 1.0 : (double)xv[u];
                        const double *gr = s_ge + (f + u) * CT;  // (f + u < 64: rows past nt are zero)
#pragma unroll
                        for (int c = 0; c < CT; ++c) acc[c] = __builtin_fma(xd, gr[c], acc[c]);
                        ivacc = __builtin_fma(xd * xd, s_rs[f + u], ivacc);
                    }
                }
            }
        }
        flush();
        if (real) __hip_atomic_fetch_add(a.g_iv + col, -0.5 * ivacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
}

void smm_launch_emission_bwd(const SmmEmBwdArgs &a, int c_need, hipStream_t stream)
{
    const int grid = a.n_chunks < 1024 ? a.n_chunks : 1024;
    if (c_need <= 8) hipLaunchKernelGGL(smm_emission_bwd_kernel<8>, dim3(grid), dim3(256), 0, stream, a);
    else if (c_need <= 16) hipLaunchKernelGGL(smm_emission_bwd_kernel<16>, dim3(grid), dim3(256), 0, stream, a);
    else if (c_need <= 24) hipLaunchKernelGGL(smm_emission_bwd_kernel<24>, dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(smm_emission_bwd_kernel<32>, dim3(grid), dim3(256), 0, stream, a);
}

void smm_launch_widen(const float *src, double *dst, size_t n, hipStream_t stream)
{
    if (n == 0) return;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(smm_widen_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, n);
}
