// EXPERIMENT, NOT BUILT (round 5; record: profiles/round5_emission_stream.txt).  The parts that csrc/smm_emission.hip had at the end of the
// experiment and has no more (smm_emission_stream_kernel, its launch, the item cost): the pair kernel's arithmetic (bit-identical output) behind buffer loads with
// hand-placed waits, operand-ordered weights in LDS (ds_read_b128), one straight-line body per group count, persistent workgroups on
// cost-weighted ranges of a class-set-sorted item list, waves that take pairs from a counter in LDS, loads running on across pairs.
// It needed three host-side additions that are not in the tree either (smm_api.hip: em_order = each launch part's videos sorted by
// class set, em_cost = items weighted by class set, both in the metadata block; smm_launch.h: t_max / cost arguments of
// smm_launch_emission).  It passed every emission test and measured 0.60 (static items) .. 0.70 ms (cost-weighted ranges) on cfg3
// where the shipped pair kernel takes 0.60; on one-class-set corpora 0.535 / 0.545 / 0.576 / 0.605 ms at 11 / 15 / 19 / 23 states
// against the pair kernel's 0.518 / 0.535 / 0.579 / 0.681.  Development switches: -DSMM_EM_ABLATE (bits), -DSMM_EM_STAMP (cycle
// stamps, scripts/prof_emission_stamps.py), -DSMM_EM_PRIO, -DSMM_EM_CHUNK_ITEMS, -DSMM_EM_STAGGER.

// ---------------------------------------------------------------------------------------------------------------
// Round 5: the pair kernel's arithmetic with a third of its instruction stream (smm_emission_stream_kernel).
//
// The pair kernel issues ~170 instructions beside the 24 MFMAs of a macro-step (16 features x 32 frames, 21..24 states): clamped
// 64-bit addresses per load, a scalar branch around every 4x4x4 MFMA (the group count is a property of the video), 8-byte LDS
// reads of the operands, branches around every store.  The ablation of round 4 put that stream ALONE at 0.30 ms of the kernel's
// 0.60 on cfg3 -- as much as the matrix work or the loads of x, and not overlapped with either.  Here:
//   * the group count NG of the workgroup's video selects one of the straight-line bodies below ONCE (template argument);
//   * x arrives through buffer loads: one descriptor per macro-step built from wave-uniform scalars (base = the pair's first
//     feature of that macro-step, records = the bytes left in the video), so rows past the video's end read zeros and the
//     per-lane offset (row * D + 4 k) * 4 is ONE register for the whole kernel -- no per-load VALU address arithmetic, no clamps;
//     a ring of four macro-steps per wave, three in flight (6 KB) while the fourth feeds the MFMAs;
//   * the weights sit in LDS in the order the operands are read: per macro-step [k][j pair][state][2] for the 16-state tile,
//     [group][k][state & 3][j] for the 4-state groups, [k][j] for inv_var -- every read is a conflict-free or broadcast
//     ds_read_b128 (4 + 2 NG per macro-step instead of 8 + 4 NG ds_read_b64), 28..55 KB instead of 34..60;
//   * elp leaves through buffer stores whose descriptor ends at the video's last frame: no branch per row.
// The sequence of MFMAs and of the x^2 additions per accumulator is the pair kernel's, so the output is bit-identical.
typedef unsigned int smm_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int smm_u2 __attribute__((ext_vector_type(2)));
#define SMM_BUF_FLAGS 0x00020000

#ifdef SMM_EM_STAMP   // development builds only: s_memtime stamps of the first workgroups' waves, item by item (scripts/prof_emission_stamps.py)
#define SMM_EM_STAMP_WGS 8
#define SMM_EM_STAMP_ITEMS 12
__device__ unsigned long long smm_em_stamps[SMM_EM_STAMP_WGS][SMM_EM_WAVES][SMM_EM_STAMP_ITEMS][8];
extern "C" int smm_dev_em_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(smm_em_stamps), sizeof(unsigned long long) * SMM_EM_STAMP_WGS * SMM_EM_WAVES * SMM_EM_STAMP_ITEMS * 8);
}
#define SMM_STAMP(k) do { if (stamp) stamp[k] = __builtin_readcyclecounter(); } while (0)
#else
#define SMM_STAMP(k) do { } while (0)
#endif

// An item of the flat grid = chunk `chunk` of a video: the pairs [chunk * ppc, (chunk + 1) * ppc) of its (ntiles + 1) / 2 pairs of
// 16-frame tiles, ppc = 4 * tiles-per-wave (the host's blk_cum counts ceil(tiles / (8 tpw)) items per video).
struct SmmEmItem {
    const char *xvb;       // the video's first feature
    double *erow;          // the video's first elp row
    int T, xbytes, p0, np, group;   // frames, bytes of x, first pair and pairs of the item, class set
    int pad_;                       // (no padding bytes: hipcc copies them through scratch)
};

__device__ __forceinline__ SmmEmItem smm_em_item(int item, const SmmVideo *__restrict__ videos, const int32_t *__restrict__ order,
                                                 const int32_t *__restrict__ blk_cum, int nvid, int blk_base, int ppc,
                                                 const float *__restrict__ xall, double *__restrict__ elp64, int D, int cm)
{
    const int bid = item + blk_base;
    const int slot = smm_em_find_video(blk_cum, nvid, bid);
    const int chunk = bid - blk_cum[slot];
    const SmmVideo mv = videos[order[slot]];
    SmmEmItem it;
    it.xvb = reinterpret_cast<const char *>(xall + (size_t)mv.frame_off * D);
    it.erow = elp64 + (size_t)mv.frame_off * cm;
    it.T = mv.T;
    it.xbytes = mv.T * D * 4;
    const int npairs = (((mv.T + 15) >> 4) + 1) >> 1;
    it.p0 = chunk * ppc;
    it.np = npairs - it.p0 < ppc ? npairs - it.p0 : ppc;
    it.group = mv.group;
    it.pad_ = 0;
    return it;
}

// One SEGMENT of a persistent workgroup's range of items: the items [item, item_end) as long as they belong to the class set of the
// first.  The weights go to LDS once; then every wave is on its own: it takes pairs from the segment's counter in LDS (one
// ds_add_rtn per pair), so the waves of a workgroup finish together however unevenly the SIMDs' arbiters treat them (waves 4..7 of a
// workgroup run a macro-step in 2 900 cycles where waves 0..3 need 2 150), and its ring of loads runs on ACROSS pairs: the pair
// after the current one is taken when the current one starts.  Returns the first item behind the segment.
template <int NG>
__device__ __forceinline__ int smm_em_stream_segment(double *__restrict__ wl, int item, const int item_end, const SmmEmItem first,
                                                     const SmmVideo *__restrict__ videos, const int32_t *__restrict__ order,
                                                     const int32_t *__restrict__ n_states, const int32_t *__restrict__ blk_cum,
                                                     const int nvid, const int blk_base, const int ppc,
                                                     const float *__restrict__ xall, const double *__restrict__ wall,
                                                     const double *__restrict__ cstall, const double *__restrict__ iv,
                                                     double *__restrict__ elp64, const int D, const int cm, unsigned long long *stamp)
{
    constexpr int MS = 256 + 64 * NG + 16;                          // doubles per macro-step in LDS
    const int g = first.group;
    const int C = n_states[g];
    const double *__restrict__ w = wall + (size_t)g * D * cm;
    const double *__restrict__ cst = cstall + (size_t)g * cm;
    const int D16 = (D + 15) & ~15;
    const int nms = D16 >> 4;
    SMM_STAMP(1);
    // ---- weights -> LDS, in operand order (consecutive threads write consecutive doubles)
    __syncthreads();                                                 // every wave has left the previous segment's table
    // (four elements per thread and round: the four loads are issued together -- one at a time, each behind its own wait, the fill of
    // 5 200 doubles took 11 round trips to L2, 15..30 k cycles: profiles/round5_emission_stream.txt)
    for (int i0 = threadIdx.x; i0 < nms * MS; i0 += 4 * SMM_EM_WAVES * 64) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * SMM_EM_WAVES * 64;
            const int ic = i < nms * MS ? i : 0;
            const int ms = ic / MS, r = ic - ms * MS;
            int d, c;
            if (r < 256) {                                           // [kq][jp][fr][e]: w[16 ms + 4 kq + 2 jp + e][fr]
                d = 16 * ms + 4 * (r >> 6) + 2 * ((r >> 5) & 1) + (r & 1);
                c = (r >> 1) & 15;
            } else if (r < 256 + 64 * NG) {                          // [g][kq][jj][j]: w[16 ms + 4 kq + j][16 + 4 g + jj]
                const int q = r - 256;
                d = 16 * ms + 4 * ((q >> 4) & 3) + (q & 3);
                c = 16 + 4 * (q >> 6) + ((q >> 2) & 3);
            } else {                                                 // [kq][j]: inv_var[16 ms + 4 kq + j]
                d = 16 * ms + (r - 256 - 64 * NG);
                c = -1;
            }
            const bool ok = d < D && c < C;
            const size_t off = ok ? (c < 0 ? (size_t)d : (size_t)d * cm + c) : 0;
            const double *src = c < 0 ? iv : w;
            const double x = (SMM_EM_ABLATE & 64) ? (double)(d + c) : src[off];   // (unconditional load from a clamped address)
            v[u] = ok ? x : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * SMM_EM_WAVES * 64;
            if (i < nms * MS) wl[i] = v[u];
        }
    }
    // (cst too: a global load behind the barrier would be hipcc's to wait for -- vmcnt(0) with the fetches in flight included)
    if (threadIdx.x < 32) wl[nms * MS + threadIdx.x] = (int)threadIdx.x < C ? cst[threadIdx.x] : 0.0;
    int *const ctr = reinterpret_cast<int *>(wl + nms * MS + 32);    // the segment's pair counter
    if (threadIdx.x == 0) *ctr = 0;
    __syncthreads();
    SMM_STAMP(2);
    const double *__restrict__ lcst = wl + nms * MS;
    // (the lane-derived offsets below are recomputed per segment: hoisted out of the kernel's loop, those of all the bodies of a
    // kernel are alive at once and some end up in scratch)
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int fr = lane & 15, kq = lane >> 4;
    const int jj = lane & 3, row1 = (fr & 12) + kq;                  // 4x4x4 result: state 16 + 4 g + jj of frame row1
    const double *__restrict__ l16 = wl + kq * 64 + fr * 2;          // + 32 jp      (+ ms * MS)
    const double *__restrict__ lg = wl + 256 + (kq * 4 + jj) * 4;    // + 64 g
    const double *__restrict__ liv = wl + 256 + 64 * NG + kq * 4;
    const int voff0 = (fr * D + 4 * kq) * 4, voff1 = voff0 + 64 * D;

    // ---- the wave's walk over the segment's pairs: q = ds_add_rtn(counter) is pair q - q_base of item `cur`
    SmmEmItem cur = first;
    int q_base = 0;
    bool more = true;                                                // (uniform) the segment has items behind `cur`
    struct Pair { const char *xvb; double *erow; int T, xbytes, f0, ok; };   // (ok an int: no padding bytes, see SmmEmItem)
    auto take = [&]() -> Pair {
        Pair pr{nullptr, nullptr, 0, 0, 0, 0};
        if (!more) return pr;
        int q1 = 0;                                                  // (one lane asks: every lane of the wave would add its own 1)
        if ((threadIdx.x & 63) == 0) q1 = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int q = __builtin_amdgcn_readfirstlane(q1);
        while (q >= q_base + cur.np) {                               // on to the item that holds pair q
            q_base += cur.np;
            if (++item >= item_end) { more = false; return pr; }
            cur = smm_em_item(item, videos, order, blk_cum, nvid, blk_base, ppc, xall, elp64, D, cm);
            if (cur.group != g) { more = false; return pr; }
        }
        pr.xvb = cur.xvb; pr.erow = cur.erow; pr.T = cur.T; pr.xbytes = cur.xbytes;
        pr.f0 = 32 * (cur.p0 + q - q_base);
        pr.ok = 1;
        return pr;
    };
    // (every wave has to find the segment's end by itself: `item` is left at the first item behind it)
    auto finish = [&]() {
        while (more) {
            if (++item >= item_end) break;
            const SmmEmItem nx = smm_em_item(item, videos, order, blk_cum, nvid, blk_base, ppc, xall, elp64, D, cm);
            if (nx.group != g) break;
        }
    };

    // ---- fetch side: macro-step f_ms of pair fp.  The loads are inline assembly and so are their waits: hipcc's own wait at the
    // head of the loop -- the merge of the first round and the steady state, with the stores of a finished pair pending on one
    // path -- let only 2..3 of the loads in flight stay there.  Hand-placed: one fetch per consume, always, so the step about to be
    // consumed has exactly two fetches (4 loads) behind it: vmcnt(4).  (Stores of a finished pair are younger than those loads;
    // they can only make the wait longer, not wrong: loads return in order.)
    Pair cp = take();                                                // the pair being consumed
    Pair np = cp.ok ? take() : cp;                                   // the one behind it
    Pair fp = cp;                                                    // the one being fetched
    bool f_on_next = false;                                          // fp is np (the fetch side runs two steps ahead)
    int f_ms = 0;
    auto fetch = [&](smm_u4 (&b)[2]) {
        const int off = fp.f0 * D * 4 + 64 * f_ms;
        const int left = fp.ok ? fp.xbytes - off : 0;                // (<= 0 behind the video's end or the segment's: nothing is read)
        const uint64_t base = reinterpret_cast<uint64_t>(fp.xvb) + (uint64_t)(left > 0 ? off : 0);
        smm_u4 rs;                                                   // buffer descriptor: base, stride 0, records, flags
        rs.x = (unsigned)base;
        rs.y = (unsigned)(base >> 32) & 0xffffu;
        rs.z = (unsigned)(left > 0 ? left : 0);
        rs.w = SMM_BUF_FLAGS;
        if (SMM_EM_ABLATE & 2) { b[0] = (smm_u4){(unsigned)voff0, 1u, 2u, (unsigned)left}; b[1] = (smm_u4){(unsigned)voff1, 3u, 4u, (unsigned)left}; }
        else {
            asm volatile("buffer_load_dwordx4 %0, %2, %4, 0 offen\n\tbuffer_load_dwordx4 %1, %3, %4, 0 offen"
                         : "=&v"(b[0]), "=&v"(b[1]) : "v"(voff0), "v"(voff1), "s"(rs) : "memory");
        }
        if (++f_ms == nms) { f_ms = 0; fp = np; f_on_next = true; }  // (np is at least one pair ahead of what is consumed)
    };
    auto arrived = [&](smm_u4 (&b)[2]) {                             // the fetch two fetches back has landed
        if (!(SMM_EM_ABLATE & 2)) asm volatile("s_waitcnt vmcnt(4)" : "+v"(b[0]), "+v"(b[1]) : : "memory");
    };

    // ---- compute side
    int c_ms = 0;
    smm_d4 acc[2] = {(smm_d4){0.0, 0.0, 0.0, 0.0}, (smm_d4){0.0, 0.0, 0.0, 0.0}};
    constexpr int G1 = NG > 0 ? NG : 1;
    double acc1[2][G1];
#pragma unroll
    for (int g4 = 0; g4 < G1; ++g4) { acc1[0][g4] = 0.0; acc1[1][g4] = 0.0; }
    double q2[2] = {0.0, 0.0};

    // (the 16-state tile's operands are read a macro-step ahead of the MFMAs that use them where the registers allow; the groups' and
    // inv_var at the top of their own macro-step, used behind its first 16x16x4 instructions / its last)
    double wcur[4];
    auto load16 = [&](int ms, double (&wv)[4]) {
        const double2 a0 = *reinterpret_cast<const double2 *>(l16 + ms * MS);
        const double2 a1 = *reinterpret_cast<const double2 *>(l16 + ms * MS + 32);
        wv[0] = a0.x; wv[1] = a0.y; wv[2] = a1.x; wv[3] = a1.y;
    };
    // Operand timing.  No groups: the next macro-step's 16-state operands are read while this one's MFMAs run (8 registers).  With
    // groups there is a better place and no register to spare: the macro-step issues its eight 16x16x4 MFMAs FIRST (512 cycles),
    // reads the next macro-step's 16-state operands into the registers they have just left, and then issues the 4x4x4 MFMAs of its
    // groups (128..256 cycles: the read lands under them), whose own operands were read at the macro-step's top, 512 cycles
    // earlier -- no MFMA waits for LDS.  (Interleaved j by j with the operands read just in front, the 17..24-state class sets ran
    // their matrix pipes at 46 % where the <= 16-state ones reach 73 %: profiles/round5_emission_stream.txt.)  Per accumulator the
    // order of the MFMAs is the same either way.
    constexpr bool AHEAD = NG == 0;
    load16(0, wcur);
    auto consume = [&](const smm_u4 (&b)[2]) {
        double wnext[4], ivc[4];
        if constexpr (AHEAD) load16(c_ms + 1 < nms ? c_ms + 1 : 0, wnext);
        {
            const double2 i0 = *reinterpret_cast<const double2 *>(liv + c_ms * MS);
            const double2 i1 = *reinterpret_cast<const double2 *>(liv + c_ms * MS + 2);
            ivc[0] = i0.x; ivc[1] = i0.y; ivc[2] = i1.x; ivc[3] = i1.y;
        }
        double wg[G1][4];
        if constexpr (NG > 0) {
#pragma unroll
            for (int g4 = 0; g4 < NG; ++g4) {
                const double2 a0 = *reinterpret_cast<const double2 *>(lg + c_ms * MS + 64 * g4);
                const double2 a1 = *reinterpret_cast<const double2 *>(lg + c_ms * MS + 64 * g4 + 2);
                wg[g4][0] = a0.x; wg[g4][1] = a0.y; wg[g4][2] = a1.x; wg[g4][3] = a1.y;
            }
        }
        double av[2][4];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            // (by value through __uint_as_float: __builtin_bit_cast of a vector COMPONENT reads component 0 whichever is named)
            if (SMM_EM_ABLATE & 16) {
                av[tl][0] = __hiloint2double(0x3ff00000, b[tl].x); av[tl][1] = __hiloint2double(0x3ff00000, b[tl].y);
                av[tl][2] = __hiloint2double(0x3ff00000, b[tl].z); av[tl][3] = __hiloint2double(0x3ff00000, b[tl].w);
            } else {
            av[tl][0] = (double)__uint_as_float(b[tl].x); av[tl][1] = (double)__uint_as_float(b[tl].y);
            av[tl][2] = (double)__uint_as_float(b[tl].z); av[tl][3] = (double)__uint_as_float(b[tl].w);
            }
        }
        if constexpr (NG > 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (SMM_EM_ABLATE & 4) { acc[0][j] += av[0][j] + wcur[j]; acc[1][j] += av[1][j]; }
            else {
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0][j], wcur[j], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1][j], wcur[j], acc[1], 0, 0, 0);
            }
        }
        if constexpr (NG > 0) {
            __builtin_amdgcn_sched_barrier(0);
            load16(c_ms + 1 < nms ? c_ms + 1 : 0, wcur);             // the next macro-step's, into the registers just read
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(SMM_EM_ABLATE & 8)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int g4 = 0; g4 < NG; ++g4) {
                        acc1[0][g4] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[0][j], wg[g4][j], acc1[0][g4], 0, 0, 0);
                        acc1[1][g4] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[1][j], wg[g4][j], acc1[1][g4], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!(SMM_EM_ABLATE & 1)) {
            q2[0] = fma(av[0][j] * ivc[j], av[0][j], q2[0]);
            q2[1] = fma(av[1][j] * ivc[j], av[1][j], q2[1]);
            }
            if constexpr (AHEAD) wcur[j] = wnext[j];
        }
        if (++c_ms < nms) return;
        // pair finished.  q2: this lane summed the features with k index kq of frame fr; add the four k groups
        // (the constants and the store offsets are made here, once per pair, not kept in registers across the macro-steps)
        const double cstv = lcst[fr];
        double cst1[G1];
#pragma unroll
        for (int g4 = 0; g4 < G1; ++g4) cst1[g4] = NG > 0 ? lcst[16 + 4 * (g4 < 3 ? g4 : 3) + jj] : 0.0;
        const int so = (fr < C) ? fr * 8 : -1;                       // store offsets: a column past the class set is out of range
        int so1[G1];
#pragma unroll
        for (int g4 = 0; g4 < G1; ++g4) so1[g4] = (NG > 0 && 16 + 4 * g4 + jj < C) ? (row1 * cm + 16 + 4 * g4 + jj) * 8 : -1;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int f0 = cp.f0 + 16 * tl;
            const int rows = cp.T - f0;                              // (<= 0: the pair's second tile lies behind the video)
            const int64_t nb = (int64_t)rows * cm * 8;
            const unsigned n64 = rows > 0 ? (nb < 0x7fffffffll ? (unsigned)nb : 0x7fffffffu) : 0u;
            double *const row0 = cp.erow + (size_t)(rows > 0 ? f0 : 0) * cm;
            double qq = q2[tl];
            qq += __shfl_xor(qq, 16);
            qq += __shfl_xor(qq, 32);
            double v4[4], v1[G1];
#pragma unroll
            for (int i = 0; i < 4; ++i) v4[i] = (cstv + acc[tl][i]) - 0.5 * __shfl(qq, kq + 4 * i);
            if constexpr (NG > 0) {
                const double qr = __shfl(qq, row1);
#pragma unroll
                for (int g4 = 0; g4 < NG; ++g4) v1[g4] = (cst1[g4] + acc1[tl][g4]) - 0.5 * qr;
            }
            {
                __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(row0, 0, (SMM_EM_ABLATE & 32) ? (n64 & 8) : n64, SMM_BUF_FLAGS);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(smm_u2, v4[i]), r, so < 0 ? -1 : so + (kq + 4 * i) * cm * 8, 0, 0);
                if constexpr (NG > 0) {
#pragma unroll
                    for (int g4 = 0; g4 < NG; ++g4)
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(smm_u2, v1[g4]), r, so1[g4], 0, 0);
                }
            }
            acc[tl] = (smm_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int g4 = 0; g4 < G1; ++g4) acc1[tl][g4] = 0.0;
            q2[tl] = 0.0;
        }
        c_ms = 0;
        // on to the next pair: the fetch side is there already (or about to be); take the one behind it
        cp = np;
        if (!f_on_next) fp = np;                                     // (cannot happen with nms >= 2; keeps the two sides together)
        f_on_next = false;
        np = cp.ok ? take() : cp;
    };

    // a ring of three macro-steps: two in flight (4 KB per wave, 64 KB per CU) while the third feeds the MFMAs (a fourth stage does
    // not fit 128 VGPRs at two groups), running on across the wave's pairs
    SMM_STAMP(3);
    smm_u4 b0[2], b1[2], b2[2];
    fetch(b0);
    fetch(b1);
    while (cp.ok) {
        fetch(b2); arrived(b0); consume(b0);
        if (!cp.ok) break;
        fetch(b0); arrived(b1); consume(b1);
        if (!cp.ok) break;
        fetch(b1); arrived(b2); consume(b2);
    }
    SMM_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");              // (the fetches behind the last step: nothing read, but counted)
    finish();
    SMM_STAMP(6);
#ifdef SMM_EM_STAMP
    if (stamp) stamp[7] = __builtin_amdgcn_s_memrealtime();         // (100 MHz: the shader clock follows from stamp 6 - stamp 0)
#endif
    return item;
}

// PERSISTENT workgroups (two per CU), each on a contiguous range of the flat grid's items (an item = a chunk of a video; the
// launch's videos arrive sorted by class set: smm_api.hip, em_order).  Why, from the ablation builds and the cycle stamps of round 5
// (profiles/round5_emission_stream.txt): a one-item-per-workgroup launch spent a third of every workgroup's life outside the
// streaming loop -- dispatch, the search of blk_cum, 7..20 k cycles of weight fill, and the wait for the workgroup's slowest wave --
// so that prologues, streaming and stores ADDED UP to the kernel's time with the matrix work hidden completely.
template <int NGM>
__global__ void __launch_bounds__(SMM_EM_WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 4)))   // (two workgroups per CU: <= 128 VGPRs)
smm_emission_stream_kernel(const SmmVideo *__restrict__ videos, const int32_t *__restrict__ order, const int32_t *__restrict__ n_states,
                           const float *__restrict__ xall, const double *__restrict__ wall, const double *__restrict__ cstall,
                           const double *__restrict__ iv, double *__restrict__ elp64, int D, int cm,
                           const int32_t *__restrict__ blk_cum, int nvid, int blk_base, int n_items, int ppc,
                           const int32_t *__restrict__ cost_cum, int cost_base, int cost_total)
{
    extern __shared__ __attribute__((aligned(16))) double wl[];
    // the workgroup's range of items: an equal share of the launch's COST (items weighted by their class set's matrix work: by item
    // counts the workgroups of 21..24-state class sets had half as much again to do as those of <= 16 states)
    if (cost_total < 0) cost_total = cost_cum[nvid] - cost_base;     // (a launch over all videos)
    auto bound = [&](int b) -> int {
        if (b <= 0) return 0;
        if (b >= (int)gridDim.x) return n_items;
        const int tgt = cost_base + (int)((int64_t)b * cost_total / gridDim.x);
        const int slot = smm_em_find_video(cost_cum, nvid, tgt);
        const int n_it = blk_cum[slot + 1] - blk_cum[slot];
        const int per = n_it > 0 ? (cost_cum[slot + 1] - cost_cum[slot]) / n_it : 1;
        return blk_cum[slot] - blk_base + (tgt - cost_cum[slot]) / per;
    };
#ifdef SMM_EM_CHUNK_ITEMS
  // (development: chunks of SMM_EM_CHUNK_ITEMS consecutive items dealt round-robin instead of one cost-weighted range per workgroup)
  for (int ch = blockIdx.x; ch * SMM_EM_CHUNK_ITEMS < n_items; ch += gridDim.x) {
    const int item0 = ch * SMM_EM_CHUNK_ITEMS, item1 = item0 + SMM_EM_CHUNK_ITEMS < n_items ? item0 + SMM_EM_CHUNK_ITEMS : n_items;
    (void)bound;
#else
  {
    const int item0 = bound(blockIdx.x), item1 = bound(blockIdx.x + 1);
#endif
    int item = item0;
#ifdef SMM_EM_STAMP
    int seg_no = 0;
#endif
    while (item < item1) {
        unsigned long long *stamp = nullptr;
#ifdef SMM_EM_STAMP
        if (blockIdx.x % 73 == 0 && blockIdx.x / 73 < SMM_EM_STAMP_WGS && seg_no < SMM_EM_STAMP_ITEMS && (threadIdx.x & 63) == 0)
            stamp = &smm_em_stamps[blockIdx.x / 73][threadIdx.x >> 6][seg_no][0];    // (workgroups 0, 73, .. 511)
        ++seg_no;
        SMM_STAMP(0);
        if (stamp) stamp[4] = __builtin_amdgcn_s_memrealtime();
#endif
        const SmmEmItem first = smm_em_item(item, videos, order, blk_cum, nvid, blk_base, ppc, xall, elp64, D, cm);
        const int C = n_states[first.group];
        const int ng1 = C > 16 ? (C - 13) >> 2 : 0;
#define SMM_EM_SEG(NG_) smm_em_stream_segment<NG_>(wl, item, item1, first, videos, order, n_states, blk_cum, nvid, blk_base, ppc, xall, wall, \
                                                   cstall, iv, elp64, D, cm, stamp)
        if (NGM == 0 || ng1 == 0) item = SMM_EM_SEG(0);
        else if (NGM == 1 || ng1 == 1) item = SMM_EM_SEG((NGM >= 1 ? 1 : 0));
        else item = SMM_EM_SEG((NGM >= 2 ? 2 : 0));
#undef SMM_EM_SEG
    }
  }
}


// ---- the item cost the host weighted the ranges by
int smm_emission_item_cost(int n_states)
{
    // measured, not the matrix cycles (4 : 5 : 6): cycle stamps of workgroups that served one class set each on cfg3 put an item of
    // 17..20 states at 1.76 and one of 21..24 states at 2.25 items of <= 16 states (profiles/round5_emission_stream.txt)
    static const int cost[5] = {4, 5, 6, 7, 8};
    const int ng = n_states > 16 ? (n_states - 13) >> 2 : 0;
    return cost[ng < 4 ? ng : 4];
}


// ---- inside smm_launch_emission, in front of the pair kernel's dispatch
#if 0
    // round 5: the same pairs through the streaming kernel (buffer loads, operand-ordered weights, one body per group count)
    // (its byte offsets into a video are 32-bit)
    if (SMM_EM_STREAM && pair && ng <= 2 && !a.cons && !a.elp32 && cost_cum && t_max > 0 && ((int64_t)t_max + 64) * a.d * 4 < (1ll << 30)) {
        auto go3 = [&](auto kern, int ngm) {
            const size_t lds = sizeof(double) * ((d16 / 16) * (256 + 64 * ngm + 16) + 32 + 2);
            if (lds > 48 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            const int n_wg = n_blocks < 2 * smm_em_cus() ? n_blocks : 2 * smm_em_cus();       // persistent: two workgroups per CU
            hipLaunchKernelGGL(kern, dim3(n_wg), block, lds, stream, a.videos, order_v, a.n_states, a.x, a.w, a.cst, a.inv_var, a.elp64,
                               a.d, a.c_max, blk_cum, nvid, blk_base, n_blocks, 4 * tpw, cost_cum, cost_base, cost_total);
        };
        switch (ng) {
        case 0: go3(smm_emission_stream_kernel<0>, 0); break;
        case 1: go3(smm_emission_stream_kernel<1>, 1); break;
        default: go3(smm_emission_stream_kernel<2>, 2); break;      // (three groups: 128 VGPRs are not enough -- the pair kernel)
        }
        return;
    }
#endif
