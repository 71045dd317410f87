// fp32 contraction on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16), gfx950.
// Included by igemm_split_tu.hip inside namespace gan, after igemm_common.h (IgemmArgs / ClassInfo / fdiv).
//
// The fp32 MFMA runs at the vector rate (64 FLOP/clk/SIMD); the bf16 MFMA is 16 times faster.
// An fp32 value x is written EXACTLY as the sum of three bf16 values h + m + l (each holds the next
// 8 significant bits of x: 24 = 3 x 8), so a product a*b is the sum of nine bf16 x bf16 products,
// each of which the matrix pipe forms exactly and adds into an fp32 accumulator.  The three
// smallest ones (m*l, l*m, l*l <= 2^-23 |a*b| together) are dropped, which is below the rounding
// error of a single fp32 product; the remaining six cost 6/16 of the fp32 MFMA's time.
//
// The split happens when a tile is written to LDS: the loaders fetch fp32 exactly like the fp32
// kernel (same gather arithmetic, same masking), split every value into its three parts with four
// integer / fp32 vector instructions, and store three bf16 images per operand.  LDS image of one
// operand and one part: [k / 8][row position][8 consecutive k] (16 bytes per row and 8-deep group),
// so a fragment of the 32x32x16 MFMA (lane: row l % 32, k = 8 * (l / 32) .. +7) is one 16-byte read
// from 32 consecutive addresses per half wave.
//
// Loader shapes (BK = 16 deep slices, 256 threads):
//   * operand contiguous along k in memory (A of FPROP / TPROP, B of TPROP): a thread loads one
//     float4 = 4 consecutive k of one row per pass and writes 8 bytes per part;
//   * operand contiguous along its rows (B of FPROP, A and B of WGRAD): a thread loads 4 consecutive
//     k rows x 2 adjacent rows (four float2), transposes in registers and writes 8 bytes per part
//     and row.  Adjacent lanes hold the two 4-deep halves of one 8-deep group, and the row pair
//     (2c, 2c+1) is stored at positions (c, ROWS/2 + c), so a wave's stores cover consecutive
//     addresses; the epilogue undoes that row permutation.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Probe builds only (make STAMPS=1 -> libgan_hip_stamps.so, tools/phase_probe.py): thread 0 of every workgroup
// leaves the constant-rate 100 MHz clock at six points of the kernel in p.stamps[workgroup][0..5], and the shader
// clock counter around the slice loop in [6..7] (their ratio is the clock the chip holds inside the loop).
#ifdef GAN_PHASE_STAMPS
#define GAN_STAMP(i) do { if (threadIdx.x == 0 && p.stamps) { p.stamps[(long)blockIdx.x * 8 + (i)] = (long long)wall_clock64(); \
        if ((i) == 2 || (i) == 3) p.stamps[(long)blockIdx.x * 8 + 4 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } } while (0)
#else
#define GAN_STAMP(i) do { } while (0)
#endif

// The gathers go through buffer descriptors: a masked-off lane (padding tap, row beyond the problem,
// tail of K) presents an offset beyond the buffer and the hardware range check returns zeros, so
// neither the address nor the loaded values need a select per element.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned kOutOfRange = 0xfffffff0u;

// x0, x1 -> packed bf16 pairs (x0 in the low half) of the three parts
__device__ __forceinline__ void split3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
#ifdef GAN_ABLATE_SPLIT   // probe builds: one instruction per part instead of the exact split (wrong results)
    h = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
    m = h ^ 0x00010001u;
    l = h ^ 0x00020002u;
    return;
#endif
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u);   // exact: the low 16 significand bits
    const float r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u);   // exact: at most 8 significant bits left
    const float s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);            // (u1 & 0xffff0000) | (u0 >> 16)
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// position in the LDS image -> row of the tile, for an operand filled by the row-pair loader
template <int ROWS>
__device__ __forceinline__ int unpermute(int pos) {
    return 2 * (pos % (ROWS / 2)) + pos / (ROWS / 2);
}

// FAST selects the cheap gather arithmetic (the host checks its condition):
//   FPROP / TPROP: the gathered image's channel count is a multiple of 16, so a 16-deep slice lies
//     inside ONE filter tap: the tap decode depends only on the slice number and runs on the scalar
//     unit, and a thread's offset is a constant of the thread plus a per-slice scalar;
//   WGRAD: the small image's width is a multiple of 4, so the 4 consecutive pixels a thread gathers
//     share an image row: one pixel decode per slice instead of four.
//
// KG = 2: split-K inside the workgroup.  The workgroup has two groups of four waves; both work on the SAME
// output tile, group g on the g-th half of the workgroup's k range with staging buffers of its own, and the
// second group hands its accumulators to the first through LDS at the end.  A launch then needs half as many
// split-K slabs (written once, read once by the reduction) for the same number of waves on the chip.
//
// ROW3 (FPROP / WGRAD of a big image whose channel count cb is no multiple of 4: the 3-channel layers that read or
// write RGB pixels, the 1 / 11 / 21-channel layers of the y-conditioned MNIST model).  The gather axis
// (tap, channel) is re-indexed per FILTER ROW: the kw * cb values one filter row touches are one contiguous run of
// floats in memory, padded to p.row3_spr slices of 16 (the pad multiplies zeros).  Element e of the run is column
// x0 + e / cb, channel e % cb, loaded as one dword with its own range check: neither a per-element tap decode nor
// vector alignment is needed.  FPROP: the virtual K is kh * spr * 16; WGRAD: the virtual M.
//
// REFLECT (FPROP / WGRAD): taps outside the big image read its mirror image (the `reflect` flag of the fp32 kernel as
// a compile-time switch: a run-time branch inside the slice loop would split it into basic blocks, and the pinned
// MFMA / vector interleave and the counted waits on the loads in flight only hold inside one block).
template <int MODE, int BM, int BN, int WM, int WN, bool FAST, int KG = 1, bool ROW3 = false, bool REFLECT = false>
__global__ __launch_bounds__(256 * KG) void igemm_split_kernel(const IgemmArgs p) {
    static_assert(!REFLECT || MODE != MODE_TPROP, "mirrored taps: forward and weight gradient only");
    static_assert(!ROW3 || (MODE != MODE_TPROP && !FAST), "the filter-row gather serves FPROP and WGRAD, general arithmetic");
    static_assert(WM * WN == 4, "four waves per block");
    static_assert(BM % 64 == 0 && BN % 32 == 0, "tile sides are multiples of 64 rows and 32 columns");
    static_assert(BN % 64 == 0 || MODE != MODE_TPROP, "the 32-wide tile has no loader for a k-contiguous B");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr bool A_TRANS = (MODE != MODE_WGRAD);  // A source contiguous along k
    constexpr bool B_TRANS = (MODE == MODE_TPROP);  // B source contiguous along k
    constexpr int GA = (BM + 8) * 16, GB = (BN + 8) * 16;  // bytes of one 8-deep group of one part
    constexpr int PA = 2 * GA, PB = 2 * GB;                // one part of a 16-deep slice
    constexpr int STAGE = 3 * (PA + PB);
    // vector instructions of one slice (split, address arithmetic) per MFMA of that slice
#ifdef GAN_VALU_PER_MFMA   // (experiments: the interleave density of the 128 x 128 tiles)
    constexpr int kValuPerMfma = (TM * TN >= 4) ? GAN_VALU_PER_MFMA : (TM * TN == 2 ? (FAST ? 8 : 11) : (FAST ? 14 : 20));
#else
    constexpr int kValuPerMfma = (TM * TN >= 4) ? (FAST ? 5 : 7) : (TM * TN == 2 ? (FAST ? 8 : 11) : (FAST ? 14 : 20));
#endif
    static_assert(KG == 1 || KG == 2, "one or two k groups");
    static_assert(KG == 1 || TM * TN * 16 * 256 * 4 <= KG * 2 * STAGE, "the accumulator hand-over must fit the staging buffers");
    __shared__ __attribute__((aligned(16))) unsigned char smem_all[KG * 2 * STAGE];

    if ((int)blockIdx.x >= p.own_blocks) {    // (uniform per workgroup) carried job: last stage of a bias-gradient column sum
        if (threadIdx.x < 256) colsum_final_block(p.tail, (int)blockIdx.x - p.own_blocks);
        return;
    }
    GAN_STAMP(0);
    const int kgrp = KG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
    unsigned char* const smem_raw = smem_all + kgrp * (2 * STAGE);
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const TileId tile = tile_of_workgroup<MODE>(p);
    const int split = tile.z % p.splits;
    const int ci = tile.z / p.splits;
    const gan_conv_geom& g = p.g;

    int M, N, K;
    ClassInfo cl = p.cls[0];
    if constexpr (MODE == MODE_TPROP) {
        if (ci == 1) cl = p.cls[1];
        else if (ci == 2) cl = p.cls[2];
        else if (ci == 3) cl = p.cls[3];
        M = cl.M; K = cl.K; N = g.cb;
    } else {
        M = p.M; N = p.N; K = p.K;
    }
    const int m0 = tile.mt * BM;
    if (m0 >= M) return;  // block-uniform
    const int n0 = tile.nt * BN;
    const int nkt = (K + BK - 1) / BK;
    // the workgroup's k range, divided over its k groups; every group runs `per_group` slices (a group whose
    // share is shorter multiplies zeros at the end: the barriers must match)
    const int bk0 = split * p.kt_per_split;
    const int bk1 = min(nkt, bk0 + p.kt_per_split);
    const int per_group = (max(bk1 - bk0, 0) + KG - 1) / KG;
    const int kt0 = bk0 + kgrp * per_group;
    const int kt1 = min(bk1, kt0 + per_group);
    const int kend = min(K, max(kt1, kt0) * BK);

    // ---------------------------------------------------------------- A loader set-up
    constexpr int A_PASS = A_TRANS ? BM / 64 : 4;   // k-contiguous: rows per thread; else: k rows per thread
    constexpr int A_W = A_TRANS ? 4 : 2;
    const int SH = (MODE == MODE_TPROP) ? g.hs : g.hb;
    const int SW = (MODE == MODE_TPROP) ? g.ws : g.wb;
    const int SC = (MODE == MODE_TPROP) ? g.cs : g.cb;
    const FastDiv dSC = (MODE == MODE_TPROP) ? p.d_cs : p.d_cb;
    const FastDiv dTW = (MODE == MODE_TPROP) ? cl.d_kwc : p.d_kw;
    // k-contiguous loader: thread -> (4-deep k group, row)
    const int t_kq = tid & 3, t_r = tid >> 2;
    int a_base[A_TRANS ? A_PASS : 1], a_y0[A_TRANS ? A_PASS : 1], a_x0[A_TRANS ? A_PASS : 1];
    // row-pair loader of A (WGRAD): thread -> (k group a_kq, row pair a_rc); 2 * BM threads take part
    const bool a_on = A_TRANS || tid < 2 * BM;
    const int a_rc = (tid >> 1) % (BM / 2);
    const int a_kq = ((tid / BM) << 1) | (tid & 1);
    int a_kh = 0, a_kw = 0, a_c = 0;
    bool a_mvalid = true;
    if constexpr (A_TRANS) {
        const int row_w = (MODE == MODE_TPROP) ? cl.wq : g.ws;
        const int row_hw = (MODE == MODE_TPROP) ? cl.hq * cl.wq : g.hs * g.ws;
#pragma unroll
        for (int ps = 0; ps < A_PASS; ++ps) {
            const int m = m0 + t_r + ps * 64;
            const bool ok = m < M;
            const unsigned um = ok ? (unsigned)m : 0u;
            const int img = um / (unsigned)row_hw;
            const int rem = um - img * row_hw;
            const int y = rem / row_w, x = rem - y * row_w;
            a_base[ps] = ok ? img * SH * SW : -1;
            if constexpr (MODE == MODE_TPROP) {
                a_y0[ps] = cl.hq0 + y;
                a_x0[ps] = cl.wq0 + x;
            } else {
                a_y0[ps] = y * g.sh - g.pt;
                a_x0[ps] = x * g.sw - g.pl;
            }
        }
    } else {
        const int m = m0 + 2 * a_rc;
        a_mvalid = a_on && m < M;
        const unsigned um = a_mvalid ? (unsigned)m : 0u;
        const int tap = um / (unsigned)g.cb;
        a_c = um - tap * g.cb;
        a_kh = tap / g.kw;
        a_kw = tap - a_kh * g.kw;
    }

    // ---------------------------------------------------------------- B loader set-up
    constexpr int B_PASS = B_TRANS ? BN / 64 : 4;
    constexpr int B_W = B_TRANS ? 4 : 2;
    // row-pair loader of B: the LAST 2 * BN threads take part (the first ones carry A's when BM is small)
    constexpr int B_T0 = 256 - (2 * BN < 256 ? 2 * BN : 256);
    const int bt = tid - B_T0;
    const bool b_on = B_TRANS || bt >= 0;
    const int btc = bt >= 0 ? bt : 0;
    const int b_rc = (btc >> 1) % (BN / 2);
    const int b_kq = ((btc / BN) << 1) | (btc & 1);

    // filter-row gather: the run elements of this thread (FPROP: 4 consecutive ones per slice half; WGRAD: its row pair)
    const int r3_len = ROW3 ? g.kw * g.cb : 0;       // floats of a filter row
    const int r3_rows = ROW3 ? p.row3_spr * 16 : 1;  // virtual rows (WGRAD) / k (FPROP) per filter row
    int w_ta = 0, w_e = 0;                           // WGRAD: filter row and run element of the thread's first row
    if constexpr (ROW3 && MODE == MODE_WGRAD) {
        const int mv = m0 + 2 * a_rc;
        w_ta = mv / r3_rows;
        w_e = mv - w_ta * r3_rows;
        a_mvalid = a_on && w_ta < g.kh;
    }
    // thread-constant byte offsets of the fast paths (a per-slice scalar is added to them)
    unsigned a_rb[A_TRANS ? A_PASS : 1], b_rb[B_TRANS ? B_PASS : 1], b_cj[4];
    bool b_nvalid[B_TRANS ? B_PASS : 1];
    if constexpr (A_TRANS) {
#pragma unroll
        for (int ps = 0; ps < A_PASS; ++ps)
            a_rb[ps] = (unsigned)((a_base[ps] + a_y0[ps] * SW + a_x0[ps]) * SC + t_kq * 4) * 4u;
    }
    if constexpr (B_TRANS) {
#pragma unroll
        for (int ps = 0; ps < B_PASS; ++ps) {
            const int n = n0 + t_r + ps * 64;
            b_nvalid[ps] = n < N;
            b_rb[ps] = (unsigned)(n * g.cs + t_kq * 4) * 4u;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) b_cj[j] = (unsigned)((b_kq * 4 + j) * N + n0 + 2 * b_rc) * 4u;
    }
    const bool b_pair_valid = b_on & (n0 + 2 * b_rc < N);
    // an odd N: the last column has no partner.  The pair load then fetches the first value of the next k row (or, at
    // the very end, what the buffer range check makes of it) beside it; that value is replaced by zero when the slice
    // is written to LDS -- a select there, not a branch around the load here, which would cut the slice loop into
    // basic blocks and force a wait on every load behind it.
    const bool b_single = !B_TRANS && (n0 + 2 * b_rc + 1 == N);

    typedef float AReg[A_PASS][A_W];
    typedef float BReg[B_PASS][B_W];
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, (int)p.b_bytes, 0x00020000);
    auto ld4 = [](const __amdgpu_buffer_rsrc_t& r, unsigned byte_off, float (&dst)[4]) {
#ifdef GAN_ABLATE_GLOAD   // probe builds: no memory operation, the offset stays live (wrong results)
        const u32x4 v = {byte_off, byte_off + 1u, byte_off + 2u, byte_off + 3u};
#else
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
#endif
        dst[0] = __uint_as_float(v[0]); dst[1] = __uint_as_float(v[1]);
        dst[2] = __uint_as_float(v[2]); dst[3] = __uint_as_float(v[3]);
    };
    auto ld2 = [](const __amdgpu_buffer_rsrc_t& r, unsigned byte_off, float (&dst)[2]) {
#ifdef GAN_ABLATE_GLOAD
        const u32x2 v = {byte_off, byte_off + 1u};
#else
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
#endif
        dst[0] = __uint_as_float(v[0]); dst[1] = __uint_as_float(v[1]);
    };

    auto ld1 = [](const __amdgpu_buffer_rsrc_t& r, unsigned byte_off, float& dst) {
        dst = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
    };
    // byte offset of run element e (column x0 + e / cb, channel e % cb) of image row `rowbase` (in pixels), or out of range
    auto r3_off = [&](int rowbase, int x0, int e, bool ok) -> unsigned {
        const int tc = (int)(((unsigned)e * (unsigned)p.row3_mul) >> 16);      // e / cb (e < 144)
        int col = x0 + tc;
        if constexpr (REFLECT) col = reflect_at(col, g.wb);
        ok = ok & (e < r3_len) & ((unsigned)col < (unsigned)g.wb);
        // (the offset is worked out unconditionally: with the arithmetic inside the conditional expression the compiler
        // keeps a branch around it, and a branch splits the slice loop's basic block)
        const unsigned off = (unsigned)((rowbase + col) * g.cb + (e - g.cb * tc)) * 4u;
        return ok ? off : kOutOfRange;
    };

    auto load_a = [&](int kt, AReg& a_reg) {
        if constexpr (ROW3 && MODE == MODE_FPROP) {
            const int ta = kt / p.row3_spr;
            const int e0 = (kt - ta * p.row3_spr) * 16 + t_kq * 4;
            const bool kv = kt * BK < kend;
#pragma unroll
            for (int ps = 0; ps < A_PASS; ++ps) {
                int iy = a_y0[ps] + ta;
                if constexpr (REFLECT) iy = reflect_at(iy, SH);
                const bool rowok = kv & (a_base[ps] >= 0) & ((unsigned)iy < (unsigned)SH);
                const int rowbase = a_base[ps] + iy * SW;
#pragma unroll
                for (int j = 0; j < 4; ++j) ld1(rsrc_a, r3_off(rowbase, a_x0[ps], e0 + j, rowok), a_reg[ps][j]);
            }
        } else if constexpr (ROW3) {
            // WGRAD: rows (w_ta, w_e) and (w_ta, w_e + 1) of 4 consecutive pixels
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kt * BK + a_kq * 4 + j;
                const bool kv = (k < kend) & a_mvalid;
                const unsigned uk = kv ? (unsigned)k : 0u;
                const int img = fdiv(uk, p.d_hsws);
                const int rem = uk - img * (int)p.d_hsws.d;
                const int oy = fdiv(rem, p.d_ws);
                const int ox = rem - oy * g.ws;
                int iy = oy * g.sh + w_ta - g.pt;
                if constexpr (REFLECT) iy = reflect_at(iy, g.hb);
                const bool rowok = kv & ((unsigned)iy < (unsigned)g.hb);
                const int rowbase = (img * g.hb + iy) * g.wb, x0 = ox * g.sw - g.pl;
                ld1(rsrc_a, r3_off(rowbase, x0, w_e, rowok), a_reg[j][0]);
                ld1(rsrc_a, r3_off(rowbase, x0, w_e + 1, rowok), a_reg[j][1]);
            }
        } else if constexpr (A_TRANS && FAST) {
            // scalar: the slice's tap and first channel
            const unsigned k0 = (unsigned)__builtin_amdgcn_readfirstlane(kt * BK);
            const int tapc = fdiv(k0, dSC);
            const int cc0 = k0 - tapc * SC;
            const int ta = fdiv(tapc, dTW);
            const int tb = tapc - ta * (int)dTW.d;
            const int sgn = (MODE == MODE_TPROP) ? -1 : 1;
            const unsigned u = (unsigned)__builtin_amdgcn_readfirstlane((sgn * (ta * SW + tb) * SC + cc0) * 4);
            const bool kv = (int)k0 + t_kq * 4 < kend;
            if constexpr (MODE == MODE_FPROP && REFLECT) {
                // mirrored taps: the offset is no longer base + per-slice constant
                const unsigned uc = (unsigned)__builtin_amdgcn_readfirstlane(cc0 * 4) + t_kq * 16;
#pragma unroll
                for (int ps = 0; ps < A_PASS; ++ps) {
                    const int iy = reflect_at(a_y0[ps] + ta, SH), ix = reflect_at(a_x0[ps] + tb, SW);
                    const bool ok = kv & (a_base[ps] >= 0);
                    const unsigned off = (unsigned)((a_base[ps] + iy * SW + ix) * SC) * 4u + uc;
                    ld4(rsrc_a, ok ? off : kOutOfRange, a_reg[ps]);
                }
            } else {
#pragma unroll
                for (int ps = 0; ps < A_PASS; ++ps) {
                    const int iy = a_y0[ps] + sgn * ta, ix = a_x0[ps] + sgn * tb;
                    const bool ok = kv & (a_base[ps] >= 0) & ((unsigned)iy < (unsigned)SH) & ((unsigned)ix < (unsigned)SW);
                    ld4(rsrc_a, ok ? a_rb[ps] + u : kOutOfRange, a_reg[ps]);
                }
            }
        } else if constexpr (A_TRANS) {
            if constexpr (MODE == MODE_TPROP) {
                // pairs: the gathered channel count may be even but no multiple of 4 -- four consecutive k can then
                // straddle two taps, two cannot -- so each half of the thread's quad is decoded on its own and loaded 8
                // bytes at a time (straight-line code: a trip count or a branch on `cs % 4` would split the slice loop)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int k = kt * BK + t_kq * 4 + 2 * hf;
                    const bool kv = k < kend;
                    const unsigned uk = kv ? (unsigned)k : 0u;
                    const int tapc = fdiv(uk, dSC);
                    const int cc = uk - tapc * SC;
                    const int ta = fdiv(tapc, dTW);
                    const int tb = tapc - ta * (int)dTW.d;
#pragma unroll
                    for (int ps = 0; ps < A_PASS; ++ps) {
                        const int iy = a_y0[ps] - ta, ix = a_x0[ps] - tb;
                        const bool ok = kv & (a_base[ps] >= 0) & ((unsigned)iy < (unsigned)SH) & ((unsigned)ix < (unsigned)SW);
                        const unsigned off = (unsigned)((a_base[ps] + iy * SW + ix) * SC + cc) * 4u;
                        float two[2];
                        ld2(rsrc_a, ok ? off : kOutOfRange, two);
                        a_reg[ps][2 * hf] = two[0];
                        a_reg[ps][2 * hf + 1] = two[1];
                    }
                }
            } else {
                const int k = kt * BK + t_kq * 4;
                const bool kv = k < kend;
                const unsigned uk = kv ? (unsigned)k : 0u;
                const int tapc = fdiv(uk, dSC);
                const int cc = uk - tapc * SC;
                const int ta = fdiv(tapc, dTW);
                const int tb = tapc - ta * (int)dTW.d;
#pragma unroll
                for (int ps = 0; ps < A_PASS; ++ps) {
                    int iy = a_y0[ps] + ta, ix = a_x0[ps] + tb;
                    if constexpr (REFLECT) { iy = reflect_at(iy, SH); ix = reflect_at(ix, SW); }
                    const bool ok = kv & (a_base[ps] >= 0) & ((unsigned)iy < (unsigned)SH) & ((unsigned)ix < (unsigned)SW);
                    const unsigned off = (unsigned)((a_base[ps] + iy * SW + ix) * SC + cc) * 4u;
                    ld4(rsrc_a, ok ? off : kOutOfRange, a_reg[ps]);
                }
            }
        } else if constexpr (FAST) {
            // the thread's 4 pixels are consecutive in one row of the small image
            const int k = kt * BK + a_kq * 4;
            const bool kv = (k < kend) & a_mvalid;
            const unsigned uk = kv ? (unsigned)k : 0u;
            const int img = fdiv(uk, p.d_hsws);
            const int rem = uk - img * (int)p.d_hsws.d;
            const int oy = fdiv(rem, p.d_ws);
            const int ox = rem - oy * g.ws;
            const int iy = oy * g.sh + a_kh - g.pt, ix0 = ox * g.sw + a_kw - g.pl;
            if constexpr (REFLECT) {
                const unsigned rowb = (unsigned)((img * g.hb + reflect_at(iy, g.hb)) * g.wb);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ix = reflect_at(ix0 + j * g.sw, g.wb);
                    const unsigned off = (unsigned)((rowb + ix) * g.cb + a_c) * 4u;
                    ld2(rsrc_a, kv ? off : kOutOfRange, a_reg[j]);
                }
            } else {
                const bool rowok = kv & ((unsigned)iy < (unsigned)g.hb);
                const unsigned off0 = (unsigned)(((img * g.hb + iy) * g.wb + ix0) * g.cb + a_c) * 4u;
                const unsigned step = (unsigned)(g.sw * g.cb) * 4u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = rowok & ((unsigned)(ix0 + j * g.sw) < (unsigned)g.wb);
                    ld2(rsrc_a, ok ? off0 + j * step : kOutOfRange, a_reg[j]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kt * BK + a_kq * 4 + j;
                const bool kv = (k < kend) & a_mvalid;
                const unsigned uk = kv ? (unsigned)k : 0u;
                const int img = fdiv(uk, p.d_hsws);
                const int rem = uk - img * (int)p.d_hsws.d;
                const int oy = fdiv(rem, p.d_ws);
                const int ox = rem - oy * g.ws;
                int iy = oy * g.sh + a_kh - g.pt, ix = ox * g.sw + a_kw - g.pl;
                if constexpr (REFLECT) { iy = reflect_at(iy, g.hb); ix = reflect_at(ix, g.wb); }
                const bool ok = kv & ((unsigned)iy < (unsigned)g.hb) & ((unsigned)ix < (unsigned)g.wb);
                const unsigned off = (unsigned)(((img * g.hb + iy) * g.wb + ix) * g.cb + a_c) * 4u;
                ld2(rsrc_a, ok ? off : kOutOfRange, a_reg[j]);
            }
        }
    };

    auto load_b = [&](int kt, BReg& b_reg) {
        if constexpr (B_TRANS && FAST) {
            const unsigned k0 = (unsigned)__builtin_amdgcn_readfirstlane(kt * BK);
            const int tapc = fdiv(k0, p.d_cs);
            const int cc0 = k0 - tapc * g.cs;
            const int ta = fdiv(tapc, cl.d_kwc);
            const int tb = tapc - ta * cl.kwc;
            const int wtap = (cl.ph + g.sh * ta) * g.kw + (cl.pw + g.sw * tb);
            const unsigned u = (unsigned)__builtin_amdgcn_readfirstlane((wtap * g.cb * g.cs + cc0) * 4);
            const bool kv = (int)k0 + t_kq * 4 < kend;
#pragma unroll
            for (int ps = 0; ps < B_PASS; ++ps) ld4(rsrc_b, (kv & b_nvalid[ps]) ? b_rb[ps] + u : kOutOfRange, b_reg[ps]);
        } else if constexpr (B_TRANS) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {     // (pairs, as for A)
                const int k = kt * BK + t_kq * 4 + 2 * hf;
                const bool kv = k < kend;
                const unsigned uk = kv ? (unsigned)k : 0u;
                const int tapc = fdiv(uk, p.d_cs);
                const int cc = uk - tapc * g.cs;
                const int ta = fdiv(tapc, cl.d_kwc);
                const int tb = tapc - ta * cl.kwc;
                const int wtap = (cl.ph + g.sh * ta) * g.kw + (cl.pw + g.sw * tb);
#pragma unroll
                for (int ps = 0; ps < B_PASS; ++ps) {
                    const int n = n0 + t_r + ps * 64;
                    const bool ok = kv & (n < N);
                    const unsigned off = (unsigned)((wtap * g.cb + n) * g.cs + cc) * 4u;
                    float two[2];
                    ld2(rsrc_b, ok ? off : kOutOfRange, two);
                    b_reg[ps][2 * hf] = two[0];
                    b_reg[ps][2 * hf + 1] = two[1];
                }
            }
        } else if constexpr (ROW3 && MODE == MODE_FPROP) {
            // virtual k row r of slice kt = run element e of filter row ta: weight row ta * kw * 3 + e, zeros beyond the run
            const int ta = kt / p.row3_spr;
            const int eb = (kt - ta * p.row3_spr) * 16;
            const unsigned u = (unsigned)__builtin_amdgcn_readfirstlane((ta * r3_len + eb) * N * 4);
            const bool kv = kt * BK < kend;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned off = (b_pair_valid & kv & (eb + b_kq * 4 + j < r3_len)) ? b_cj[j] + u : kOutOfRange;
                ld2(rsrc_b, off, b_reg[j]);
            }
        } else {
            // row-major [K][N]: a constant of the thread per k row plus the slice's scalar offset
            const unsigned u = (unsigned)__builtin_amdgcn_readfirstlane(kt * BK * N * 4);
            const int kb = kt * BK + b_kq * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned off = (b_pair_valid & (kb + j < kend)) ? b_cj[j] + u : kOutOfRange;
                ld2(rsrc_b, off, b_reg[j]);
            }
        }
    };

    // 4 consecutive k of one row -> 8 bytes of each part at (8-deep group kq >> 1, half kq & 1)
    auto put = [&](unsigned char* img, int part_bytes, int group_bytes, int kq, int pos, float x0, float x1, float x2,
                   float x3) {
        unsigned h01, m01, l01, h23, m23, l23;
        split3(x0, x1, h01, m01, l01);
        split3(x2, x3, h23, m23, l23);
        unsigned char* q = img + (kq >> 1) * group_bytes + pos * 16 + (kq & 1) * 8;
#ifdef GAN_ABLATE_LDSW    // probe builds: the parts stay live, nothing is written (wrong results)
        asm volatile("" :: "v"(h01), "v"(h23), "v"(m01), "v"(m23), "v"(l01), "v"(l23), "v"(q));
#else
        *reinterpret_cast<uint2*>(q) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(q + part_bytes) = make_uint2(m01, m23);
        *reinterpret_cast<uint2*>(q + 2 * part_bytes) = make_uint2(l01, l23);
#endif
    };

    auto store_a = [&](int buf, const AReg& a_reg) {
        unsigned char* As = smem_raw + buf * STAGE;
        if constexpr (A_TRANS) {
#pragma unroll
            for (int ps = 0; ps < A_PASS; ++ps)
                put(As, PA, GA, t_kq, t_r + ps * 64, a_reg[ps][0], a_reg[ps][1], a_reg[ps][2], a_reg[ps][3]);
        } else {
            if (a_on) {
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    put(As, PA, GA, a_kq, e * (BM / 2) + a_rc, a_reg[0][e], a_reg[1][e], a_reg[2][e], a_reg[3][e]);
            }
        }
    };
    auto store_b = [&](int buf, const BReg& b_reg) {
        unsigned char* Bs = smem_raw + buf * STAGE + 3 * PA;
        if constexpr (B_TRANS) {
#pragma unroll
            for (int ps = 0; ps < B_PASS; ++ps)
                put(Bs, PB, GB, t_kq, t_r + ps * 64, b_reg[ps][0], b_reg[ps][1], b_reg[ps][2], b_reg[ps][3]);
        } else {
            if (b_on) {
                put(Bs, PB, GB, b_kq, b_rc, b_reg[0][0], b_reg[1][0], b_reg[2][0], b_reg[3][0]);
                put(Bs, PB, GB, b_kq, BN / 2 + b_rc, b_single ? 0.f : b_reg[0][1], b_single ? 0.f : b_reg[1][1],
                    b_single ? 0.f : b_reg[2][1], b_single ? 0.f : b_reg[3][1]);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kh2 = lane >> 5;
    const int frag_a = kh2 * GA + (wm * (BM / WM) + (lane & 31)) * 16;
    const int frag_b = kh2 * GB + (wn * (BN / WN) + (lane & 31)) * 16;

    // One slice: multiply slice `kt` out of LDS buffer `buf`; meanwhile split the slice held in
    // (ra, rb) -- slice kt+1, fetched two slices ago -- into the other buffer and refill (ra, rb)
    // with slice `next` (= kt+3), which therefore has two slices of MFMAs to land.
    auto slice = [&](int buf, AReg& ra, BReg& rb, int next) {
        const unsigned char* As = smem_raw + buf * STAGE;
        const unsigned char* Bs = As + 3 * PA;
        bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#ifdef GAN_ABLATE_LDSR    // probe builds: fragments made up in registers (wrong results)
#pragma unroll
            for (int i = 0; i < TM; ++i) { unsigned u = (unsigned)(size_t)(As + s * PA + frag_a + i * 32 * 16); asm volatile("" : "+v"(u)); af[s][i] = __builtin_bit_cast(bf16x8, u32x4{u, u, u, u}); }
#pragma unroll
            for (int j = 0; j < TN; ++j) { unsigned u = (unsigned)(size_t)(Bs + s * PB + frag_b + j * 32 * 16); asm volatile("" : "+v"(u)); bf[s][j] = __builtin_bit_cast(bf16x8, u32x4{u, u, u, u}); }
#else
#pragma unroll
            for (int i = 0; i < TM; ++i) af[s][i] = *reinterpret_cast<const bf16x8*>(As + s * PA + frag_a + i * 32 * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[s][j] = *reinterpret_cast<const bf16x8*>(Bs + s * PB + frag_b + j * 32 * 16);
#endif
        }
        store_a(buf ^ 1, ra);
        store_b(buf ^ 1, rb);
        load_a(next, ra);
        load_b(next, rb);
        // six of the nine part products, smallest first: l*h, h*l, m*m, m*h, h*m, h*h
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 c = acc[i][j];
#ifdef GAN_ABLATE_MFMA    // probe builds: the fragments stay live, nothing is multiplied (wrong results)
                asm volatile("" :: "v"(af[0][i]), "v"(af[1][i]), "v"(af[2][i]), "v"(bf[0][j]), "v"(bf[1][j]), "v"(bf[2][j]));
#else
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], c, 0, 0, 0);
#endif
                acc[i][j] = c;
            }
        // Interleave: the matrix pipe needs 32 cycles per MFMA of which the issue port is busy for 8, so
        // about six vector instructions fit behind each one; left alone the scheduler clusters the
        // MFMAs behind the staging work.
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * (TM + TN), 0);   // the fragment reads
#pragma unroll
        for (int t = 0; t < 6 * TM * TN; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, kValuPerMfma, 0);  // vector ALU
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };

    GAN_STAMP(1);
    if (per_group > 0) {
        // Software pipeline with two register sets and two LDS buffers, one barrier per slice
        // (slices past the end load zeros and are never multiplied).
        AReg a0, a1;
        BReg b0, b1;
        load_a(kt0, a0);
        load_b(kt0, b0);
        load_a(kt0 + 1, a1);
        load_b(kt0 + 1, b1);
        store_a(0, a0);
        store_b(0, b0);
        load_a(kt0 + 2, a0);
        load_b(kt0 + 2, b0);
        __syncthreads();
        GAN_STAMP(2);
        // two slices per trip keep the register sets and LDS buffers compile-time; an odd slice count
        // multiplies one slice of zeros at the end
        for (int kt = kt0; kt < kt0 + per_group; kt += 2) {
            slice(0, a1, b1, kt + 3);
            slice(1, a0, b0, kt + 4);
        }
    }
    GAN_STAMP(3);
    if constexpr (KG > 1) {
        // the second group's accumulators -> LDS (register-major: conflict free), first group adds them and goes on
        // alone; a wave that has ended no longer counts at the barriers of the epilogue
        float* red = reinterpret_cast<float*>(smem_all);
        if (kgrp == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((i * TN + j) * 16 + r) * 256 + tid] = acc[i][j][r];
        }
        __syncthreads();
        if (kgrp == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((i * TN + j) * 16 + r) * 256 + tid];
    }

    GAN_STAMP(4);
    // ---------------------------------------------------------------- epilogue (as igemm_kernel, plus the
    // row / column permutation of operands filled by the row-pair loader)
    // (staging reuses the buffers, the hand-over area included: the barrier behind the row table below comes after
    // every read of it)
    float* smem = reinterpret_cast<float*>(smem_all);
    __shared__ int rowoff[BM];
    if (tid < BM) {
        const int m = m0 + tid;
        int off = -1;
        if (m < M) {
            if constexpr (MODE == MODE_TPROP) {
                const int hw = cl.hq * cl.wq;
                const int img = m / hw;
                const int rem = m - img * hw;
                const int yq = rem / cl.wq, xq = rem - yq * cl.wq;
                const int y = g.sh * (cl.hq0 + yq) + cl.ph - g.pt;
                const int x = g.sw * (cl.wq0 + xq) + cl.pw - g.pl;
                off = ((img * g.hb + y) * g.wb + x) * g.cb;
            } else if constexpr (ROW3 && MODE == MODE_WGRAD) {
                const int ta = m / r3_rows, e = m - ta * r3_rows;      // virtual row -> filter row, run element
                off = e < r3_len ? (ta * r3_len + e) * N : -1;
            } else {
                off = m * N;
            }
        }
        rowoff[tid] = off;
    }
    __syncthreads();

    float* dst = p.out + (p.splits > 1 ? (long)split * p.slab_elems : 0L);
    const bool direct = p.splits == 1;
    const bool accum = direct && p.accumulate;
    const bool epi_on = direct && epi_active(p.epi);

    int tcol[TN];  // tile column of this lane's accumulator column j
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pos = wn * (BN / WN) + j * 32 + (lane & 31);
        tcol[j] = B_TRANS ? pos : unpermute<BN>(pos);
    }

    if (p.vec_epilogue) {
        constexpr int R = ((2 * STAGE / 4) / BN / 32) * 32 < BM ? ((2 * STAGE / 4) / BN / 32) * 32 : BM;
        static_assert(R >= 32, "LDS staging must hold a 32-row group of the tile");
        constexpr int F4_ROW = BN / 4;
        const bool has_bias = direct && p.bias != nullptr;
        // staging rows are tile ROWS (after the permutation), so a pass covers rows [r0, r0 + R)
#pragma unroll
        for (int r0 = 0; r0 < BM; r0 += R) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pos = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh2;
                    const int row = A_TRANS ? pos : unpermute<BM>(pos);
                    if (row >= r0 && row < r0 + R) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) smem[(row - r0) * BN + tcol[j]] = acc[i][j][r];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int f = tid; f < R * F4_ROW; f += 256) {
                const int row = f / F4_ROW, c4 = f - row * F4_ROW;
                const int off = (r0 + row < BM) ? rowoff[r0 + row] : -1;
                const int col = n0 + c4 * 4;
                if (off >= 0 && col < N) {
                    float4 v = *reinterpret_cast<const float4*>(&smem[row * BN + c4 * 4]);
                    if (has_bias) {
                        const float4 bb = *reinterpret_cast<const float4*>(p.bias + col);
                        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                    }
                    float4* o = reinterpret_cast<float4*>(dst + off + col);
                    if (accum) {
                        const float4 old = *o;
                        v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
                    }
                    if (epi_on) v = epi_apply4(p.epi, v, (long)off + col);
                    *o = v;
                }
            }
            if (r0 + R < BM) __syncthreads();
        }
        GAN_STAMP(5);
        return;
    }

    float bias_v[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
        bias_v[j] = (direct && p.bias != nullptr && n0 + tcol[j] < N) ? p.bias[n0 + tcol[j]] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pos = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh2;
            const int off = rowoff[A_TRANS ? pos : unpermute<BM>(pos)];
            if (off < 0) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + tcol[j];
                if (col < N) {
                    float v = acc[i][j][r] + bias_v[j];
                    if (accum) v += dst[off + col];
                    if (epi_on) v = epi_apply(p.epi, v, (long)off + col);
                    dst[off + col] = v;
                }
            }
        }
    }
}
