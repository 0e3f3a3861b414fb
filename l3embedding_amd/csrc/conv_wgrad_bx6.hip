// conv_wgrad_bx6.hip -- weight gradient of a 3x3 'same' convolution as Winograd F(3x3, 2x2) with split-bf16 operands on the bf16
// matrix pipe: the partials conv_wgrad_wino.hip computes on v_mfma_f32_32x32x2_f32, from the same raw tensors, in the same layout.
//
// STATUS (round 5): an EXPERIMENT behind the debug knob L3_WG_BX6=1, parity-clean (tests/test_layer_parity_gpu.py::
// test_weight_gradient_split_bf16_experiment: 4-6e-7 of the range at all 14 layers, the fp32 kernel's own distance) and SLOWER than
// the fp32 kernel it would replace: 13.8 ms per step against 10.5 (profiles/r05_bx6_ablations.txt).  Both operands must be formed
// and split in registers (244 VALU per 24 MFMAs), 128 accumulator registers per wave leave room for ONE tile pair's reads in
// flight, so the operand phases run at LDS latency, and the per-stage LDS-DMA requests (seven per wave, their offsets re-derived:
// there are no registers to keep them) sit in front of a wave's MFMAs.  Kept as the measured answer to "convert the weight
// gradient first" (VERDICT r04 #2b), not as a product path.
//
// dW of the Conv2D layers of l3embedding/audio_model.py:372-445, vision_model.py:126-205 under train.py:282-284:
//     dU_p[c][k] = sum_t V_p[t][c] Z_p[t][k],   V = B^T d B (input tile),  Z = A dY A^T (2x2 tile of dY),   p = 16 positions
// Both operands are formed from raw fp32 pixels in registers (wave-uniform +-1 combinations, exact) and then split EXACTLY into
// three bfloat16 terms each; six of the nine cross products (the three dropped are <= 2^-24 of the product) run on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- fp32-grade results (scripts/wino_split_error_model.py: 2e-7 of the range
// against 4e-7 for the fp32 chain), 6 bf16 MFMAs per 16 tiles where the fp32 kernel issues 8 fp32 MFMAs at 1/16 of the rate.
//
// Mapping (gfx950) -- what round 5's forward kernel taught (profiles/r05_bx6_ablations.txt):
//   block  = 64 input channels x 64 output channels x 16 positions, 8 waves, wave w = positions 2 w, 2 w + 1 (128 accumulator
//            registers, 2 waves per SIMD: both operands live in registers, 48 of them per position); split-K over the units;
//   stage  = 16 tiles = TWO units of 8 (conv_wgrad_wino.hip's unit: 1x8, 2x4 or 4x2 tiles): the MFMA's k index is the tile, lane
//            group g = lane >> 5 owns unit g of the stage and walks its 8 tiles; a lane owns the channel PAIR (2 l, 2 l + 1) of both
//            operands (element i of the pair = row / column l of MFMA tile i) and reads it as one ds_read_b64 per raw pixel;
//   phases = per position an operand phase (50 LDS reads, ~244 VALU) and a matrix phase (24 MFMAs).  One wave does not overlap its
//            own VALU with its own MFMAs, another wave's do: waves 0-3 (E) and 4-7 (O) -- one of each per SIMD -- run one phase
//            apart, one s_barrier per phase;
//   DMA    = raw pixels HBM -> LDS (buffer_load ... lds, NHWC order: 256 contiguous bytes per pixel), double buffered by stage,
//            requested by the waves in a matrix phase a full stage before the first read; ~50 KiB per 384 MFMAs (the forward
//            kernel: 128) -- this kernel is not bound by the DMA stream.
#include "kernels.h"
#include "device_common.h"
#include "wgw_common.h"

#include <stdlib.h>

#include <mutex>
#include <type_traits>

namespace l3 {

namespace {

template <int V>
using IC = std::integral_constant<int, V>;

__device__ __forceinline__ unsigned hi16_pair(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

struct Split3 {
    unsigned h, m, l;
};
// a, b = h + m + l exactly, each term the top 16 bits of an fp32 (a bfloat16), packed (low half = a)
__device__ __forceinline__ Split3 split_pair(float a, float b) {
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    Split3 s;
    s.h = hi16_pair(ua, ub);
    const float ra = a - __builtin_bit_cast(float, ua & 0xffff0000u), rb = b - __builtin_bit_cast(float, ub & 0xffff0000u);
    const unsigned va = __builtin_bit_cast(unsigned, ra), vb = __builtin_bit_cast(unsigned, rb);
    s.m = hi16_pair(va, vb);
    const float qa = ra - __builtin_bit_cast(float, va & 0xffff0000u), qb = rb - __builtin_bit_cast(float, vb & 0xffff0000u);
    s.l = hi16_pair(__builtin_bit_cast(unsigned, qa), __builtin_bit_cast(unsigned, qb));
    return s;
}

__device__ __forceinline__ f32x2 lds_pair(unsigned addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>((uintptr_t)addr);
}

template <int UC>
struct WgbGeom : WgwGeom<UC> {
    using G = WgwGeom<UC>;
    static constexpr int UNIT_BYTES = G::STAGE;               // one unit: input strip + dY strip, in 1-KiB pieces
    static constexpr int STAGE2 = 2 * UNIT_BYTES;             // a stage = two units
    static constexpr unsigned BUF1 = 0x10000;                 // stage buffer 1 sits 64 KiB up: the lane's address registers (all below
    static constexpr size_t LDS_BYTES = BUF1 + (size_t)STAGE2; // 64 KiB in buffer 0) toggle between the buffers with one XOR each
    static_assert(STAGE2 <= 0x10000, "stage buffer");
    static constexpr int PIECES2 = 2 * G::PIECES;
    static constexpr int PER_WAVE = (PIECES2 + 7) / 8;        // LDS-DMA pieces per wave and stage (7 / 6)
    __host__ __device__ static constexpr int tr(int j) { return UC == 8 ? 0 : UC == 4 ? j >> 2 : j >> 1; }
    __host__ __device__ static constexpr int tc(int j) { return UC == 8 ? j : UC == 4 ? j & 3 : j & 1; }
};

template <int UC>
__global__ __launch_bounds__(512) void conv_wgrad_bx6_kernel(WgwArgs a) {
    using G = WgbGeom<UC>;
    constexpr int UR = G::UR, XPITCH = G::XPITCH, YPITCH = G::YPITCH, PER_WAVE = G::PER_WAVE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane_id = t & 63;
    const int tiles = a.ctiles * a.ktiles;
    const int logical = xcd_remap(blockIdx.x, tiles * a.splits);         // split-major: a split's tiles share an L2
    const int sp = logical / tiles, tile = logical - sp * tiles;
    const int ct = tile / a.ktiles, kt = tile - ct * a.ktiles;
    const int c0 = ct * 64, k0 = kt * 64;
    const int u_begin = sp * a.per_split, u_end = min(a.units, u_begin + a.per_split);
    const int nunits = max(0, u_end - u_begin), nstage = (nunits + 1) >> 1;

    // ---- LDS-DMA pieces of this wave: pieces wave + 8 q of the stage's 2 x PIECES (unit 0's pieces, then unit 1's).  Per lane the
    // pixel's byte offset from the strip's origin and whether it lies outside the image are RE-DERIVED at every request (a dozen VALU
    // per piece, in a matrix phase, where the vector ALUs idle): seven pieces' worth of precomputed offsets and border masks -- what
    // conv_wgrad_wino.hip keeps -- are 14 registers this kernel does not have (128 accumulators + 48 operand registers per wave)
    const int y_last = 2 * UR * (a.uy - 1), x_last = 2 * UC * (a.ux - 1);
    const size_t xshift = (size_t)(a.W + 1) * a.Cin * 4;
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const char*>(a.x) - xshift), 0, (int)((size_t)a.N * a.H * a.W * a.Cin * 4 + xshift), 0x00020000);
    const __amdgpu_buffer_rsrc_t ysrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((size_t)a.N * a.H * a.W * a.Cout * 4), 0x00020000);
    const int per_img = a.uy * a.ux;

    // Requests the next stage (two units) into buffer `s & 1`.  Every wave calls it with s = 0, 1, 2, ... in order, so the unit
    // counters (sample, unit row, unit column of the next unit to request) advance instead of being divided out each time.
    int q_u = u_begin;
    int q_n = u_begin / per_img, q_y = (u_begin - q_n * per_img) / a.ux, q_x = u_begin - q_n * per_img - q_y * a.ux;
    auto issue = [&](int s) __attribute__((always_inline)) {
        char* S = smem + (s & 1) * G::BUF1;
        int xbv[2], ybv[2];
        unsigned flg[2], dead[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int Y0 = 2 * UR * q_y, X0 = 2 * UC * q_x;
            xbv[h] = ((q_n * a.H + Y0) * a.W + X0) * a.Cin * 4;
            ybv[h] = ((q_n * a.H + Y0) * a.W + X0) * a.Cout * 4;
            // bit 0: first unit row, 1: last unit row, 2: first unit column, 3: last unit column
            flg[h] = (unsigned)(1 - min(q_y, 1)) | (unsigned)(1 - min(a.uy - 1 - q_y, 1)) << 1 | (unsigned)(1 - min(q_x, 1)) << 2 |
                     (unsigned)(1 - min(a.ux - 1 - q_x, 1)) << 3;
            // a unit beyond the range: every lane out of range (bit 31 of the vector offset), nothing is moved, zeros are written
            dead[h] = q_u < u_end ? 0u : 0x80000000u;
            ++q_u;
            if (++q_x == a.ux) {
                q_x = 0;
                if (++q_y == a.uy) {
                    q_y = 0;
                    ++q_n;
                }
            }
        }
        // (an opaque copy of the lane index per call: otherwise everything below that does not depend on the stage is hoisted out of
        //  the stage loop and lives in registers across it -- the registers this re-derivation exists to save)
        int lane = lane_id;
        asm volatile("" : "+v"(lane));
#pragma unroll
        for (int q = 0; q < PER_WAVE; ++q) {
            const int pi = wave + 8 * q;
            if (pi < G::PIECES2) {                                 // (wave-uniform)
                const int h = pi >= G::PIECES ? 1 : 0;
                const int piece = pi - h * G::PIECES;
                const bool isx = piece < G::XPIECES;
                const int pix = (isx ? piece : piece - G::XPIECES) * 4 + (lane >> 4);
                const int py = isx ? pix / XPITCH : pix / YPITCH, px = isx ? pix - py * XPITCH : pix - py * YPITCH;
                const int ry = py - (isx ? 1 : 0), rx = px - (isx ? 1 : 0);
                const unsigned f = h ? flg[1] : flg[0];
                // (bitwise, not short-circuit: the conditions are per lane, a short-circuit would branch on EXEC per piece)
                const unsigned out = ((f & 1u) & (unsigned)(ry < 0)) | ((f >> 1 & 1u) & (unsigned)(y_last + ry >= a.H)) |
                                     ((f >> 2 & 1u) & (unsigned)(rx < 0)) | ((f >> 3 & 1u) & (unsigned)(x_last + rx >= a.W)) |
                                     (unsigned)(isx && pix >= G::XPIX);
                const unsigned vo = (unsigned)((py * a.W + px) * (isx ? a.Cin : a.Cout) * 4 + (isx ? c0 : k0) * 4 + (lane & 15) * 16) |
                                    (out << 31) | (h ? dead[1] : dead[0]);
                auto* dst = (__attribute__((address_space(3))) void*)(S + h * G::UNIT_BYTES + piece * 1024);
                // (two calls under a uniform branch: a SELECT between buffer descriptors goes through scratch memory)
                if (isx)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, dst, 16, (int)vo, h ? xbv[1] : xbv[0], 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, dst, 16, (int)vo, h ? ybv[1] : ybv[0], 0, 0);
            }
        }
    };

    const int l31 = lane_id & 31, grp = lane_id >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned lane_base = lds0 + (unsigned)(grp * G::UNIT_BYTES + l31 * 8);      // buffer 0; buffer 1 = ^ BUF1 (see WgbGeom)

    // ---- this wave's two positions, as RUN-TIME parameters (eight compile-time variants behind a switch cost the register allocator
    // ~40 spills: 222 registers for one variant, 290 for eight).  xi = wave >> 1; the pair is ordered so that the first position
    // touches ONE column of dY (nu = 0 or 3) and the second TWO (nu = 1 or 2): what remains compile-time is how many rows of dY the
    // pair touches (xi = 0, 3: one; xi = 1, 2: two) -- NR -- and the wave's group.
    const int xi = wave >> 1;
    const int nuA = (wave & 1) ? 3 : 0, nuB = (wave & 1) ? 2 : 1;
    auto sgpr = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    auto bt_a = [](int x) { return x == 0 ? 0 : x == 2 ? 2 : 1; };
    auto bt_b = [](int x) { return x == 0 ? 2 : x == 1 ? 2 : x == 2 ? 1 : 3; };
    const int ra = bt_a(xi), rb = bt_b(xi);
    const float sa = sgpr(xi == 1 ? 1.f : -1.f);                    // V = (x[ra][ca] + sa x[rb][ca]) + sb (x[ra][cb] + sa x[rb][cb])
    const float sbA = sgpr(nuA == 1 ? 1.f : -1.f), sbB = sgpr(nuB == 1 ? 1.f : -1.f);
    // Z relative to the first row / column of A the position touches (rows of A: y0, y0 + y1, y0 - y1, -y1); that one's own sign
    // (w00) goes onto the accumulators at the store
    const float r1 = sgpr(xi == 1 ? 1.f : -1.f), c1B = sgpr(nuB == 1 ? 1.f : -1.f);
    const int ry0 = xi == 3 ? 1 : 0, cy0A = nuA == 3 ? 1 : 0;
    // LDS addresses of the lane's pixels for tile 0 of its unit (the other tiles: immediate offsets), per position
    auto xadr = [&](int r, int c) { return lane_base + (unsigned)((r * XPITCH + c) * 256); };
    auto yadr = [&](int r, int c) { return lane_base + (unsigned)(G::XBYTES + (r * YPITCH + c) * 256); };
    unsigned xA[4] = {xadr(ra, bt_a(nuA)), xadr(rb, bt_a(nuA)), xadr(ra, bt_b(nuA)), xadr(rb, bt_b(nuA))};
    unsigned xB[4] = {xadr(ra, bt_a(nuB)), xadr(rb, bt_a(nuB)), xadr(ra, bt_b(nuB)), xadr(rb, bt_b(nuB))};
    unsigned yA[2] = {yadr(ry0, cy0A), yadr(1, cy0A)};                                  // [second row]
    unsigned yB[4] = {yadr(ry0, 0), yadr(ry0, 1), yadr(1, 0), yadr(1, 1)};              // [row][column]
    asm volatile("" : "+v"(xA[0]), "+v"(xA[1]), "+v"(xA[2]), "+v"(xA[3]), "+v"(xB[0]), "+v"(xB[1]), "+v"(xB[2]), "+v"(xB[3]));
    asm volatile("" : "+v"(yA[0]), "+v"(yA[1]), "+v"(yB[0]), "+v"(yB[1]), "+v"(yB[2]), "+v"(yB[3]));

    f32x16 acc[2][2][2];        // [position of the pair][ci tile i][co tile jn]
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pp][i][jn][r] = 0.f;

    auto run = [&](auto NRT, auto GRP) __attribute__((always_inline)) {
        constexpr int NR = decltype(NRT)::value;
        // operands of the position whose matrix phase comes next: A = V, both channels of the lane's pair x {h, m, l}, 8 tiles = 4
        // dwords each; B = Z of ONE channel of the pair at a time (co tile 0 is built in the operand phase, co tile 1 between the two
        // halves of the matrix phase, into the same registers: 36 operand registers instead of 48, and the operand phase -- the longer
        // one -- gets shorter by what the matrix phase gets longer)
        u32x4 Ah[2], Am[2], Al[2], Bh, Bm, Bl;
        auto lds_f = [](unsigned addr) { return *reinterpret_cast<const __attribute__((address_space(3))) float*>((uintptr_t)addr); };

        // Z of channel 2 l + JN of the lane's unit, position pp, split into Bh / Bm / Bl
        auto build_b = [&](auto PP, auto JN, unsigned buf) __attribute__((always_inline)) {
            constexpr int pp = decltype(PP)::value, jn = decltype(JN)::value;
            float z[2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (!(j & 3)) __builtin_amdgcn_sched_barrier(0);
                const int tt = j & 1, tp = j >> 1;
                const unsigned yo = (unsigned)(((2 * G::tr(j)) * YPITCH + 2 * G::tc(j)) * 256 + 4 * jn);
                float zz;
                if constexpr (pp == 0) {
                    zz = lds_f(yA[0] + yo);
                    if constexpr (NR == 2) zz = fmaf(r1, lds_f(yA[1] + yo), zz);
                } else {
                    zz = fmaf(c1B, lds_f(yB[1] + yo), lds_f(yB[0] + yo));
                    if constexpr (NR == 2) zz = fmaf(r1, fmaf(c1B, lds_f(yB[3] + yo), lds_f(yB[2] + yo)), zz);
                }
                z[tt] = zz;
                if (tt == 1) {
                    const Split3 sz = split_pair(z[0], z[1]);
                    Bh[tp] = sz.h; Bm[tp] = sz.m; Bl[tp] = sz.l;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // ---- operand phase of the pair's position pp: V of the lane's unit, one tile's reads in flight at a time (128 accumulator
        //      + 36 operand registers are live), every two tiles split into bf16 triples; then Z of the first channel ----
        auto operands = [&](auto PP, unsigned buf) __attribute__((always_inline)) {
            constexpr int pp = decltype(PP)::value;             // 0: one dY column, 1: two
            const float sb = pp ? sbB : sbA;
            float v[2][2];                   // [tile of the pair][channel of the lane's pair]
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (!(j & 1)) __builtin_amdgcn_sched_barrier(0);
                const int tt = j & 1, tp = j >> 1;
                const unsigned xo = (unsigned)(((2 * G::tr(j)) * XPITCH + 2 * G::tc(j)) * 256);
                const f32x2 xa = lds_pair((pp ? xB[0] : xA[0]) + xo), xb = lds_pair((pp ? xB[1] : xA[1]) + xo);
                const f32x2 xc = lds_pair((pp ? xB[2] : xA[2]) + xo), xd = lds_pair((pp ? xB[3] : xA[3]) + xo);
#pragma unroll
                for (int e = 0; e < 2; ++e) v[tt][e] = fmaf(sb, fmaf(sa, xd[e], xc[e]), fmaf(sa, xb[e], xa[e]));
                if (tt == 1) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const Split3 sv = split_pair(v[0][e], v[1][e]);
                        Ah[e][tp] = sv.h; Am[e][tp] = sv.m; Al[e][tp] = sv.l;
                    }
                }
            }
            build_b(PP, IC<0>{}, buf);
        };
        // ---- matrix phase: the six products per (ci tile, co tile), smallest terms first, the ci tiles alternating; co tile 1's
        //      operand is built between the two halves ----
        auto multiply = [&](auto PP, unsigned buf) __attribute__((always_inline)) {
            constexpr int pp = decltype(PP)::value;
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                if (jn == 1) build_b(PP, IC<1>{}, buf);
                const bf16x8 bh = __builtin_bit_cast(bf16x8, Bh), bm = __builtin_bit_cast(bf16x8, Bm), bl = __builtin_bit_cast(bf16x8, Bl);
                const bf16x8 ah0 = __builtin_bit_cast(bf16x8, Ah[0]), am0 = __builtin_bit_cast(bf16x8, Am[0]), al0 = __builtin_bit_cast(bf16x8, Al[0]);
                const bf16x8 ah1 = __builtin_bit_cast(bf16x8, Ah[1]), am1 = __builtin_bit_cast(bf16x8, Am[1]), al1 = __builtin_bit_cast(bf16x8, Al[1]);
                f32x16 d0 = acc[pp][0][jn], d1 = acc[pp][1][jn];
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am0, bm, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am1, bm, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am0, bh, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am1, bh, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bm, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bm, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh, d1, 0, 0, 0);
                acc[pp][0][jn] = d0;
                acc[pp][1][jn] = d1;
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto barrier = [] {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto wait_all = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        // Phase slots (one s_barrier each; both groups pass 4 nstage + 1 of them).  Both groups run the SAME loop, group O one slot
        // later (a barrier in front instead of behind):
        //   slot 4 c      E: operands P0(c)                   O: multiplies P1(c - 1)
        //   slot 4 c + 1  E: requests stage c + 1, mult. P0   O: operands P0(c)
        //   slot 4 c + 2  E: operands P1(c)                   O: requests stage c + 1, multiplies P0(c)
        //   slot 4 c + 3  E: multiplies P1(c), waits          O: operands P1(c), waits
        // Stage c is read in slots 4 c .. 4 c + 3; its buffer takes stage c + 2 from slot 4 c + 4 on; every wave has waited for its
        // own pieces of stage c + 1 before the barrier in front of slot 4 c + 4.
        constexpr bool grp_o = decltype(GRP)::value != 0;
        if constexpr (grp_o) barrier();
        for (int c = 0; c < nstage; ++c) {
            const unsigned buf = 0;          // (the lane's addresses point into the stage's buffer: toggled below)
            operands(IC<0>{}, buf);
            barrier();
            issue(c + 1);
            multiply(IC<0>{}, buf);
            barrier();
            operands(IC<1>{}, buf);
            if constexpr (grp_o) wait_all();
            barrier();
            multiply(IC<1>{}, buf);
            if constexpr (!grp_o) wait_all();
            barrier();
            // the next stage lives in the other buffer, 64 KiB away: one XOR per address register (the addresses are < 64 KiB)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                xA[k] ^= G::BUF1;
                xB[k] ^= G::BUF1;
                yB[k] ^= G::BUF1;
            }
            yA[0] ^= G::BUF1;
            yA[1] ^= G::BUF1;
        }
        if constexpr (!grp_o) barrier();
    };
    {
        issue(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool two_rows = xi == 1 || xi == 2;
        if (wave < 4) {
            if (two_rows) run(IC<2>{}, IC<0>{}); else run(IC<1>{}, IC<0>{});
        } else {
            if (two_rows) run(IC<2>{}, IC<1>{}); else run(IC<1>{}, IC<1>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- dU partials of this (split, position pair): rows = input channels, lanes = output channels (as conv_wgrad_wino.hip).
    // The lane part of the addresses comes from opaque copies made HERE: computed from l31 / grp directly, the 32 row offsets (64
    // registers) are hoisted to the top of the kernel and live across the stage loops.
    int l31s = l31, grps = grp;
    asm volatile("" : "+v"(l31s), "+v"(grps));
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int nu = pp ? nuB : nuA;
        // the sign of the first row / column of A this position touches: rows of A are y0, y0 + y1, y0 - y1, -y1
        const float w00 = (xi == 3 ? -1.f : 1.f) * (nu == 3 ? -1.f : 1.f);
        float* out = a.part + ((size_t)(sp * 16 + xi * 4 + nu) * a.Cin + c0 + 8 * grps) * a.Cout + k0 + 2 * l31s;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 2 * ((r & 3) + 8 * (r >> 2)) + i;                     // (+ 8 grp: in `out`)
                __builtin_nontemporal_store(f32x2{w00 * acc[pp][i][0][r], w00 * acc[pp][i][1][r]},
                                            reinterpret_cast<f32x2*>(out + (size_t)c * a.Cout));
            }
    }
}

template <int UC>
void launch_wgb(const WgwArgs& a, hipStream_t s) {
    using G = WgbGeom<UC>;
    static std::once_flag once[L3_MAX_DEVICES];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & (L3_MAX_DEVICES - 1)], [] {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_bx6_kernel<UC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    });
    hipLaunchKernelGGL((conv_wgrad_bx6_kernel<UC>), dim3(a.ctiles * a.ktiles * a.splits), dim3(512), G::LDS_BYTES, s, a);
}

}  // namespace

void conv_wgrad_bx6_launch(const WgwArgs& a, int uc, hipStream_t s) {
    if (uc == 8)
        launch_wgb<8>(a, s);
    else if (uc == 4)
        launch_wgb<4>(a, s);
    else
        launch_wgb<2>(a, s);
}

}  // namespace l3
