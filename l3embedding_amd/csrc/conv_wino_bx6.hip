// conv_wino_bx6.hip -- 3x3 / stride 1 / pad 1 convolution as Winograd F(2x2, 3x3) with fp32 operands split into three
// bfloat16 terms and multiplied on the bf16 matrix pipe ("bf16 x 6", fp32 accumulate): l3_config.fp32_conv =
// L3_FP32_CONV_F2X2_BF16X6.
//
// Same operator as conv_wino.hip (the Conv2D forward / data-gradient launches of the VGG blocks,
// l3embedding/audio_model.py:372-445, vision_model.py:126-205), same decomposition
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A,     M_p[tile][k] = sum_c V_p[tile][c] U_p[c][k]  for the 16 positions p
//
// but the 16 GEMMs do not run on v_mfma_f32_32x32x2_f32.  On gfx950 the fp32 MFMA executes on the vector ALUs (64 flop per
// clock and SIMD, and every VALU instruction beside it costs ~4.3 cycles of matrix time); the bf16 MFMA is a
// pipe of its own, 16x faster, and VALU work beside it is nearly free.  An fp32 number is EXACTLY the sum of three bfloat16
// numbers (x = h + m + l: 8 + 8 + 8 significand bits), so
//
//     v * u = (vh + vm + vl)(uh + um + ul) = vh uh + vh um + vm uh + vm um + vh ul + vl uh  +  (vm ul + vl um + vl ul)
//
// where every bf16 x bf16 product is exact in fp32 and the three dropped terms are <= 2^-24 |v u| -- below the rounding of
// the fp32 product itself.  Six v_mfma_f32_32x32x16_bf16 (K = 16, 32 cycles each) replace eight v_mfma_f32_32x32x2_f32
// (K = 2, 64 cycles each): 192 against 512 matrix cycles per 16 input channels, with fp32-grade results (measured against
// float64: NOT worse than the fp32 MFMA chain -- scripts/wino_split_error_model.py, tests/test_layer_parity_gpu.py with the
// F(2x2,3x3) bound unchanged).  U is split once per step by the filter transform; V is split in registers by the wave that
// forms it (and/sub/perm: 5.5 VALU per value, on the vector ALUs the bf16 MFMA leaves alone).
//
// Mapping (gfx950):
//   block  = 64 tiles (BTY flat tile rows x BTX tile columns) x 64 output channels, 16 waves, wave p = position (xi, nu):
//            2 x 2 MFMA tiles of 32 x 32, 64 accumulator registers, 4 waves per SIMD; persistent grid;
//   phases = a wave's stage is an operand phase (16 raw reads, 136 VALU: transform + split) followed by a matrix phase (24 MFMAs).
//            Measured (profiles/r05_bx6_ablations.txt): ONE wave does not overlap its own VALU with its own MFMAs -- a version that
//            threaded the next operands between the MFMAs of a wave (8 waves x 2 positions) took exactly the sum of its parts --
//            but the VALU of one wave runs beside the MFMAs of ANOTHER on the same SIMD nearly for free at 4 waves per SIMD
//            (scripts/probes/valu_beside_bf16_mfma.hip).  So the waves of every SIMD form two groups half a stage apart: while
//            group E (waves 0-3, 8-11) builds operands, group O (4-7, 12-15) multiplies, and vice versa; one s_barrier per half stage;
//   stage  = 16 input channels = one MFMA k-step; lane (tile l & 31, k-group g = l >> 5) owns channels 8 g .. 8 g + 7;
//   A      = the raw input pixels of the block's tiles, HBM -> LDS with buffer_load ... lds (requested by the O waves at the start
//            of their matrix phase, a full stage before the first read), double buffered, one image of
//            every input row (adjacent tile rows share two of their four): [channel quad 4][input row][column parity][column / 2]
//            x 16 B, row pitch padded so that the 16 lanes of a ds_read_b128 group hit 16 different 16-B columns.  A block of
//            flat tile rows may straddle images: every image boundary inside it inserts two rows (the lower halo of one image
//            and the upper halo of the next are different rows);
//   B      = U split into bf16 triples, [pos][Cin/16][Cout/32][term 3][k-group 2][cout 32][8 channels]: the 6 KiB a wave needs
//            per stage are contiguous, every (cout half, term) piece is 1 KiB = one LDS-DMA instruction and is read back by
//            the SAME wave with one ds_read_b128 per lane.  The pieces are private to their wave, so B is SINGLE buffered
//            (96 KiB; a second copy would not fit): as soon as a piece is in registers the wave requests the next stage's
//            piece into the same place, and waits for it with a counted vmcnt just before it reads it again;
//   output = wino_common.h (shared with the fp32 kernel: same accumulator layout).
#include "kernels.h"
#include "device_common.h"
#include "wino_common.h"

#include <stdlib.h>

#include <mutex>
#include <type_traits>

namespace l3 {

namespace {

template <int BTX>
struct Bx6Geom {
    static constexpr int BTY = 64 / BTX;
    static constexpr int PXH = BTX + 1;                         // pixels per (input row, column parity)
    // A image: [input row][column parity][column / 2] x 5 slots of 16 B: the pixel's four channel quads of the stage (64 contiguous
    // bytes in HBM -- the four lanes of a quintet fetch one pixel, so an LDS-DMA piece touches 13 cache lines; with one quad PLANE per
    // 16 B, round 5's first layout, every lane of a piece touched its own line and the A requests, a quarter of the bytes, cost more
    // address time than the three quarters of B: profiles/r05_bx6_ablations.txt) + one unused slot.  The odd pixel stride makes the
    // 16 lanes of a ds_read_b128 group (consecutive tiles = consecutive pixels of one parity) hit 16 different 16-B columns; the row
    // pitch RP keeps that true across the tile rows of a group: RP = 0 (BTX 16), 4 (BTX 8), 2 (BTX 4) mod 8 slots.
    static constexpr int PSLOTS = 5;
    static constexpr int RP = BTX == 16 ? 176 : BTX == 8 ? 92 : 50;
    static constexpr int MAXB = BTX == 16 ? 0 : BTX == 8 ? 1 : 2; // image boundaries a block of BTY flat tile rows may cross
    static constexpr int IR = 2 * BTY + 2 + 2 * MAXB;           // input rows held
    static constexpr int A_SLOTS = IR * RP;
    static constexpr int A_PIECES = (A_SLOTS + 63) / 64;        // 1-KiB pieces: 30 / 25 / 24 ...
    static constexpr int A_BYTES = 32 * 1024;                   // ... of the 32 a buffer holds: every wave issues exactly two per stage,
                                                                // so that the stage loop's vmcnt counts are the same for every wave
    static constexpr int B_BYTES = 16 * 6 * 1024;               // 16 waves x (2 cout halves x 3 terms) pieces
    static constexpr int E_BYTES = 16 * 32 * 64 * 4;            // the output transform's exchange area (wino_common.h)
    static constexpr int LOOP_BYTES = 2 * A_BYTES + B_BYTES;
    static constexpr size_t LDS_BYTES = LOOP_BYTES > E_BYTES ? LOOP_BYTES : E_BYTES;
    static_assert(A_PIECES <= 32, "two A pieces per wave at most");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(RP >= 2 * PXH * PSLOTS, "row pitch");
};

// the high halves of two fp32 bit patterns as one dword of two bfloat16 (low half = a): v_perm_b32
__device__ __forceinline__ unsigned hi16_pair(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// a = h + m + l exactly, each the top 16 bits of an fp32 (a bfloat16): truncate, subtract (exact), twice
struct Split2 {
    unsigned h, m, l;
};
__device__ __forceinline__ Split2 split_pair(float a, float b) {
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    Split2 s;
    s.h = hi16_pair(ua, ub);
    const float ra = a - __builtin_bit_cast(float, ua & 0xffff0000u), rb = b - __builtin_bit_cast(float, ub & 0xffff0000u);
    const unsigned va = __builtin_bit_cast(unsigned, ra), vb = __builtin_bit_cast(unsigned, rb);
    s.m = hi16_pair(va, vb);
    const float qa = ra - __builtin_bit_cast(float, va & 0xffff0000u), qb = rb - __builtin_bit_cast(float, vb & 0xffff0000u);
    s.l = hi16_pair(__builtin_bit_cast(unsigned, qa), __builtin_bit_cast(unsigned, qb));
    return s;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BTX, int SM>
__global__ __launch_bounds__(1024) void conv_wino_bx6_kernel(WinoArgs a) {
    using G = Bx6Geom<BTX>;
    constexpr int BTY = G::BTY, PXH = G::PXH, RP = G::RP, PSLOTS = G::PSLOTS, A_SLOTS = G::A_SLOTS;
    constexpr int A_BYTES = G::A_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Bs = smem + 2 * A_BYTES;

    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane0 = t & 63;   // wave == position
    const int total_tiles = a.mblocks * a.nblocks;
    const __amdgpu_buffer_rsrc_t xsrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)a.N * a.H * a.W * a.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t usrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, (int)((size_t)16 * a.Cin * a.Cout * 6), 0x00020000);
    const __amdgpu_buffer_rsrc_t nullsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0, 0x00020000);
#ifdef BX6_ABL_NOLOOP
    const int nstage = 0;
#else
    const int nstage = a.nchunks;                          // Cin / 16
#endif
    const int u_stage_bytes = (a.Cout >> 5) * 3072;        // one stage of one position: Cout/32 cout halves x 3 KiB
    // ---- this wave's position: V = (d[ra][ca] + sa d[rb][ca]) + sb (d[ra][cb] + sa d[rb][cb]),  rows of B^T:
    //      0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3
    const int xi = wave >> 2, nu = wave & 3;
    // waves w, w + 4, w + 8, w + 12 share a SIMD (MI355X_MICROARCH.md, LDS): (w >> 2) & 1 puts two waves of each group on every SIMD
    const int grp_o = (wave >> 2) & 1;
    const int orank = (wave & 3) + 4 * (wave >> 3);
    const int ra = xi == 0 ? 0 : xi == 2 ? 2 : 1, rbw = xi == 0 ? 2 : xi == 1 ? 2 : xi == 2 ? 1 : 3;
    const int ca = nu == 0 ? 0 : nu == 2 ? 2 : 1, cbw = nu == 0 ? 2 : nu == 1 ? 2 : nu == 2 ? 1 : 3;

    for (int lt = blockIdx.x; lt < total_tiles; lt += (int)gridDim.x) {
    if (lt != (int)blockIdx.x) __syncthreads();          // the previous tile block's last LDS reads are done
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int logical = xcd_remap(lt, total_tiles);
    // cout-block major: an XCD's contiguous range of tile blocks shares ONE 64-channel slice of U (6 KiB x Cin: <= 3 MiB, resident in
    // its 4-MiB L2 for the whole launch) -- B is three quarters of what the block moves into LDS
    const int nb = logical / a.mblocks, mb = logical - nb * a.mblocks;
    const int rb = mb / a.txb, cb = mb - rb * a.txb;
    const int R0 = rb * BTY, tx0 = cb * BTX, n0 = nb * 64;
    // the block's first image and how many of its tile rows lie in it
    const int img0 = R0 / a.TY, ty0 = R0 - img0 * a.TY;
    const int cnt0 = min(BTY, a.TY - ty0);
    const int seg0_rows = 2 * cnt0 + 2, seg_rows = 2 * a.TY + 2;

    // ---- A staging (the eight O waves, rank orank): pieces orank, 8 + orank, 16 + orank, 24 + orank;
    //      slot -> (quad, input row, parity, column / 2) -> pixel --------
    unsigned avoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int s = (orank + 8 * q) * 64 + lane;
        unsigned vo = 0x80000000u;
        if (s < A_SLOTS) {
            const int lr = s / RP, rem = s - lr * RP;
            const int pix = rem / PSLOTS, quad = rem - pix * PSLOTS;
            const int par = pix / PXH, pxh = pix - par * PXH;
            int img, yy;
            if (lr < seg0_rows) {
                img = img0;
                yy = 2 * ty0 - 1 + lr;
            } else {
                const int l2 = lr - seg0_rows, k = l2 / seg_rows;
                img = img0 + 1 + k;
                yy = l2 - k * seg_rows - 1;
            }
            const int xx = 2 * tx0 - 1 + 2 * pxh + par;
            if (par < 2 && quad < 4 && img < a.N && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W)
                vo = (unsigned)(((img * a.H + yy) * a.W + xx) * a.Cin * 4 + quad * 16);
        }
        avoff[q] = vo;
    }
    const unsigned bvoff = (unsigned)(lane * 16);
    const int bsbase = (wave * nstage * (a.Cout >> 5) + 2 * nb) * 3072;       // this wave's 6 KiB of stage 0

    // (stage == nstage: the requests behind the last stage.  They keep the loop body free of branches and the vmcnt counts the same
    //  in every stage; they go through a descriptor of zero records, so they move no data -- every lane is out of range and the
    //  buffer unit writes zeros, into a place nobody reads before it is requested again)
    auto issue_a = [&](int buf, int stage) {          // (O waves only)
        char* As = smem + buf * A_BYTES;
        const __amdgpu_buffer_rsrc_t srd = stage < nstage ? xsrd : nullsrd;
        const int asoff = stage * 64;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(As + (orank + 8 * q) * 1024), 16,
                                                     (int)avoff[q], asoff, 0, 0);
    };
    auto issue_b = [&](int stage, int jn) {          // the three terms of cout half jn
        const __amdgpu_buffer_rsrc_t srd = stage < nstage ? usrd : nullsrd;
        const int bsoff = bsbase + stage * u_stage_bytes + jn * 3072;
#pragma unroll
        for (int s = 0; s < 3; ++s)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(Bs + wave * 6144 + (jn * 3 + s) * 1024),
                                                     16, (int)bvoff + s * 1024, bsoff, 0, 0);
    };

    // ---- the lane's tiles: l31 (and 32 + l31) of the block; tile row r sits 2 r + 2 (images crossed) rows down --------
    const int l31 = lane & 31, grp = lane >> 5;
    int a_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int tr = (l31 + 32 * i) / BTX, tcol = (l31 + 32 * i) - tr * BTX;
        const int crossed = (ty0 + tr) / a.TY;                       // images crossed before this tile row
        a_off[i] = ((2 * tr + 2 * crossed) * RP + tcol * PSLOTS + 2 * grp) * 16;
    }
    auto px = [&](int r, int c) { return (r * RP + ((c & 1) * PXH + (c >> 1)) * PSLOTS) * 16; };
    const int o_aa = px(ra, ca), o_ba = px(rbw, ca), o_ab = px(ra, cbw), o_bb = px(rbw, cbw);
    const char* const b_rd = Bs + wave * 6144 + lane * 16;

    f32x16 acc1[1][2][2];
    f32x16 (&acc)[2][2] = acc1[0];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

    auto stage_loop = [&](auto SA, auto SB) {
        constexpr bool PA = decltype(SA)::value, PB = decltype(SB)::value;
        u32x4 ah[2], am[2], al[2];
        // ---- operand phase: V of the lane's 2 x 8 values of stage c, split into bf16 triples ----
        auto operands = [&](int c) {
#ifdef BX6_ABL_NOVALU
            return;
#endif
            const char* As = smem + (c & 1) * A_BYTES;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int qh = 0; qh < 2; ++qh) {
                    // one group = the four raw pixels of four channels: 16 registers in flight, not 64 (the accumulators leave ~60)
                    __builtin_amdgcn_sched_barrier(0);
                    const char* p = As + a_off[i] + qh * 16;
                    const f32x4 daa = *reinterpret_cast<const f32x4*>(p + o_aa);
                    const f32x4 dba = *reinterpret_cast<const f32x4*>(p + o_ba);
                    const f32x4 dab = *reinterpret_cast<const f32x4*>(p + o_ab);
                    const f32x4 dbb = *reinterpret_cast<const f32x4*>(p + o_bb);
                    f32x4 v;          // element by element: a vector add becomes v_pk_add_f32, which costs MFMA issue time beside MFMAs
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float t0 = PA ? daa[k] + dba[k] : daa[k] - dba[k], t1 = PA ? dab[k] + dbb[k] : dab[k] - dbb[k];
                        v[k] = PB ? t0 + t1 : t0 - t1;
                    }
                    const Split2 s0 = split_pair(v[0], v[1]), s1 = split_pair(v[2], v[3]);
                    ah[i][2 * qh] = s0.h; ah[i][2 * qh + 1] = s1.h;
                    am[i][2 * qh] = s0.m; am[i][2 * qh + 1] = s1.m;
                    al[i][2 * qh] = s0.l; al[i][2 * qh + 1] = s1.l;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // ---- matrix phase of stage c: the six products per (tile half, cout half); behind each cout half's reads the next stage's
        //      pieces are requested into the same place.  KEEP = how many younger requests may stay in flight at the reads ----
        // One LDS-DMA instruction holds the issuing wave for 60-190 cycles (MI355X_MICROARCH.md); a wave issues in order, so requests
        // at the head of the matrix phase delay its first MFMA by that much each.  They go BETWEEN the MFMAs instead (whose 32 pipe
        // cycles each cover the hold): every fourth MFMA is followed by one piece of the cout half's refill, and in an O wave every
        // second group of four also by one of its four A pieces.  WITH_A = the wave requests A(a_stage) into buffer a_stage & 1.
        auto multiply = [&](int c, auto KEEP0, auto KEEP1, auto WITH_A, int a_stage) {
            constexpr bool with_a = decltype(WITH_A)::value;
            const __amdgpu_buffer_rsrc_t asrd = a_stage < nstage ? xsrd : nullsrd;
            const __amdgpu_buffer_rsrc_t bsrd = c + 1 < nstage ? usrd : nullsrd;
            char* const Anext = smem + (a_stage & 1) * A_BYTES;
            auto req_a = [&](int q) {
#ifdef BX6_ABL_NODMA
                return;
#endif
                if constexpr (with_a)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(asrd, (__attribute__((address_space(3))) void*)(Anext + (orank + 8 * q) * 1024), 16,
                                                             (int)avoff[q], a_stage * 64, 0, 0);
            };
            auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                if (jn == 0) wait_vm<decltype(KEEP0)::value>(); else wait_vm<decltype(KEEP1)::value>();
                const u32x4 bh = *reinterpret_cast<const u32x4*>(b_rd + (jn * 3 + 0) * 1024);
                const u32x4 bm = *reinterpret_cast<const u32x4*>(b_rd + (jn * 3 + 1) * 1024);
                const u32x4 bl = *reinterpret_cast<const u32x4*>(b_rd + (jn * 3 + 2) * 1024);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the pieces are in registers: their place is free
                const int bsoff = bsbase + (c + 1) * u_stage_bytes + jn * 3072;
                auto req_b = [&](int t) {
#ifdef BX6_ABL_NODMA
                    return;
#endif
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(bsrd, (__attribute__((address_space(3))) void*)(Bs + wave * 6144 + (jn * 3 + t) * 1024),
                                                             16, (int)bvoff + t * 1024, bsoff, 0, 0);
                };
                const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bm = __builtin_bit_cast(bf16x8, bm), Bl = __builtin_bit_cast(bf16x8, bl);
                const bf16x8 Ah0 = __builtin_bit_cast(bf16x8, ah[0]), Am0 = __builtin_bit_cast(bf16x8, am[0]), Al0 = __builtin_bit_cast(bf16x8, al[0]);
                const bf16x8 Ah1 = __builtin_bit_cast(bf16x8, ah[1]), Am1 = __builtin_bit_cast(bf16x8, am[1]), Al1 = __builtin_bit_cast(bf16x8, al[1]);
                // smallest terms first; the two tile halves alternate, so an accumulator is touched every other MFMA
                f32x16 d0 = acc[0][jn], d1 = acc[1][jn];
                fence();
#ifdef BX6_ABL_NOMFMA
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, x_, y_, z_) (c_)
#endif
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al0, Bh, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al1, Bh, d1, 0, 0, 0);
                fence();
                req_a(2 * jn);
                fence();
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bl, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bl, d1, 0, 0, 0);
                fence();
                req_b(0);
                fence();
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am0, Bm, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am1, Bm, d1, 0, 0, 0);
                fence();
                req_a(2 * jn + 1);
                fence();
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am0, Bh, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am1, Bh, d1, 0, 0, 0);
                fence();
                req_b(1);
                fence();
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bm, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bm, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bh, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bh, d1, 0, 0, 0);
                fence();
                req_b(2);
                fence();
                acc[0][jn] = d0;
                acc[1][jn] = d1;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto barrier = [] {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        // Half stages ("slots"); both groups pass 2 nstage + 1 barriers:
        //   slot 2 c      E: operands(c)      O: requests A(c + 1), multiply(c - 1)
        //   slot 2 c + 1  E: multiply(c)      O: operands(c), then waits for A(c + 1)
        // A(c) is read in slots 2 c (E) and 2 c + 1 (O); its buffer takes A(c + 2) from slot 2 c + 2 on.
        using K2 = std::integral_constant<int, 2>;
        using K3 = std::integral_constant<int, 3>;
        using K5 = std::integral_constant<int, 5>;
        if (!grp_o) {
            // an E wave requests B0 x3, B1 x3 per stage: at the reads of a cout half the other half's 3 may be in flight
            for (int c = 0; c < nstage; ++c) {
                operands(c);
                barrier();
                multiply(c, K3{}, K3{}, FalseT{}, 0);
                barrier();
            }
            barrier();
        } else {
            // an O wave requests, per matrix phase and in this order, A.0 B0.0 A.1 B0.1 B0.2 | A.2 B1.0 A.3 B1.1 B1.2: at the reads of
            // a cout half the five requests behind its own three may be in flight; behind operands(c) only the last two (B1.1, B1.2)
            issue_a(1, 1);
            barrier();
            operands(0);
            wait_vm<0>();                       // (nothing but A(1) is in flight here)
            barrier();
            for (int c = 1; c < nstage; ++c) {
                multiply(c - 1, K5{}, K5{}, TrueT{}, c + 1);
                barrier();
                operands(c);
                wait_vm<2>();
                barrier();
            }
            multiply(nstage - 1, K5{}, K5{}, TrueT{}, nstage + 1);      // (A(nstage + 1): nothing is moved, nobody reads the buffer)
            barrier();
        }
    };
    issue_b(0, 0);
    issue_b(0, 1);
    if (grp_o) issue_a(0, 0);
    // (only A must be visible to the other waves; the B pieces are waited for by their own wave in the loop)
    wait_vm<0>();
    __syncthreads();
    if (xi == 1) {
        if (nu == 1) stage_loop(TrueT{}, TrueT{}); else stage_loop(TrueT{}, FalseT{});
    } else {
        if (nu == 1) stage_loop(FalseT{}, TrueT{}); else stage_loop(FalseT{}, FalseT{});
    }
    wait_vm<0>();
    __syncthreads();
#ifndef BX6_ABL_NOEPI
    wino_output<BTX, SM, 16>(a, acc1, reinterpret_cast<float*>(smem), t, wave, lane, R0, tx0, n0, mb);
#else
    if (acc[0][0][0] == 123.f && acc[1][1][3] == 5.f && acc[0][1][7] == 1.f && acc[1][0][9] == 3.f) a.y[t] = acc[0][1][1] + acc[1][0][2];
#endif
    }   // tile-block loop
}

// U = G g G^T per (input channel c, output channel k), split into bf16 triples (round to nearest even at every level: the sum
// of the three terms is the fp32 value exactly), in the layout the kernel's B pieces want:
// [pos 16][Cin/16][Cout/32][term 3][k-group 2][cout 32][8 channels].  One thread = 8 consecutive input channels of one k.
// from_fwd_for_dgrad: as conv_wino.hip (the data gradient's filter from the forward filter, flip + transpose folded in).
__device__ __forceinline__ unsigned bf16_rne(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__global__ __launch_bounds__(256) void wino_weights_bx6_kernel(const float* __restrict__ w, unsigned short* __restrict__ u3, int Cin,
                                                               int Cout, int dgrad) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= (Cin >> 3) * Cout) return;
    const int k = idx % Cout, c8 = idx / Cout;
    float gg[8][4][3];                     // G g per channel: the column transform is applied per position below
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c8 * 8 + e;
        float g[3][3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                g[kh][kw] = dgrad ? w[((size_t)((2 - kh) * 3 + (2 - kw)) * Cout + k) * Cin + c]
                                  : w[((size_t)(kh * 3 + kw) * Cin + c) * Cout + k];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            gg[e][0][kw] = g[0][kw];
            gg[e][1][kw] = 0.5f * (g[0][kw] + g[1][kw] + g[2][kw]);
            gg[e][2][kw] = 0.5f * (g[0][kw] - g[1][kw] + g[2][kw]);
            gg[e][3][kw] = g[2][kw];
        }
    }
    const int c16 = c8 >> 1, g2 = c8 & 1, cb32 = k >> 5, n = k & 31;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            unsigned h[8], m[8], l[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float g0 = gg[e][xi][0], g1 = gg[e][xi][1], g2v = gg[e][xi][2];
                const float x = nu == 0 ? g0 : nu == 1 ? 0.5f * (g0 + g1 + g2v) : nu == 2 ? 0.5f * (g0 - g1 + g2v) : g2v;
                h[e] = bf16_rne(x);
                const float r1 = x - __builtin_bit_cast(float, h[e] << 16);
                m[e] = bf16_rne(r1);
                const float r2 = r1 - __builtin_bit_cast(float, m[e] << 16);
                l[e] = bf16_rne(r2);
            }
            const int p = xi * 4 + nu;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const unsigned* src = s == 0 ? h : s == 1 ? m : l;
                const size_t o = ((((((size_t)p * (Cin >> 4) + c16) * (Cout >> 5) + cb32) * 3 + s) * 2 + g2) * 32 + n) * 8;
                u32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = src[2 * q] | (src[2 * q + 1] << 16);
                *reinterpret_cast<u32x4*>(u3 + o) = v;
            }
        }
}

template <int BTX, int SM>
void launch_bx6_2(const WinoArgs& a, hipStream_t s) {
    using G = Bx6Geom<BTX>;
    static std::once_flag once[L3_MAX_DEVICES];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & (L3_MAX_DEVICES - 1)], [] {
        (void)hipFuncSetAttribute((const void*)conv_wino_bx6_kernel<BTX, SM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    });
    static const int persist_env = l3_knob("L3_WINO_PERSIST") ? atoi(l3_knob("L3_WINO_PERSIST")) : -1;
    const int persist = persist_env >= 0 ? persist_env : 1;
    static int cus[L3_MAX_DEVICES] = {0};
    int& ncu = cus[dev & (L3_MAX_DEVICES - 1)];
    if (ncu == 0) {
        hipDeviceProp_t prop;
        ncu = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8 ? prop.multiProcessorCount / 8 * 8 : 256;
    }
    const int total = a.mblocks * a.nblocks;
    hipLaunchKernelGGL((conv_wino_bx6_kernel<BTX, SM>), dim3(persist && total > ncu ? ncu : total), dim3(1024), G::LDS_BYTES, s, a);
}
template <int BTX>
void launch_bx6(const WinoArgs& a, hipStream_t s) {
    if (a.stat_part != nullptr && a.bb.x != nullptr)
        launch_bx6_2<BTX, 2>(a, s);
    else if (a.stat_part != nullptr)
        launch_bx6_2<BTX, 1>(a, s);
    else
        launch_bx6_2<BTX, 0>(a, s);
}

}  // namespace

// tile-column block width: the least padded tile area among the shapes whose block of BTY flat tile rows crosses no more image
// boundaries than the A image has rows for (TY >= BTY; BTX = 4: TY >= 8)
int conv_wino_bx6_btx(const ConvGeom& g) {
    const int TY = (g.H + 1) / 2, TX = (g.W + 1) / 2;
    int best = 0;
    size_t waste = ~(size_t)0;
    for (int btx : {16, 8, 4}) {
        const int bty = 64 / btx;
        if (btx == 16 ? TY % 4 != 0 : btx == 8 ? TY < 8 : TY < 8) continue;      // image crossings per block: none / one / two (Bx6Geom::MAXB)
        const size_t wst = (size_t)((TX + btx - 1) / btx) * btx;            // flat rows pad only once per launch
        if (wst < waste) {
            waste = wst;
            best = btx;
        }
    }
    static const int force = l3_knob("L3_BX6_BTX") ? atoi(l3_knob("L3_BX6_BTX")) : 0;
    if ((force == 4 || force == 8 || force == 16) && (force == 16 ? TY % 4 == 0 : TY >= 8)) best = force;
    return best;
}

bool conv_wino_bx6_ok(const ConvGeom& g) {
    return g.KH == 3 && g.KW == 3 && g.padT == 1 && g.padL == 1 && g.Ho == g.H && g.Wo == g.W && g.Cin % 16 == 0 && g.Cout % 64 == 0 &&
           (size_t)16 * g.Cin * g.Cout * 6 < (1ull << 31) && conv_wino_bx6_btx(g) != 0;
}

int conv_wino_bx6_blocks(const ConvGeom& g, int n) {
    const int btx = conv_wino_bx6_btx(g), bty = 64 / btx;
    const int TY = (g.H + 1) / 2, TX = (g.W + 1) / 2;
    return ((n * TY + bty - 1) / bty) * ((TX + btx - 1) / btx);
}

// bf16 MFMA flops the kernel issues: six products per (position, tile, c, k), padded tiles included
double conv_wino_bx6_executed_flops(const ConvGeom& g) {
    return 6.0 * 2.0 * 16.0 * 64.0 * (double)conv_wino_bx6_blocks(g, g.N) * (double)g.Cin * (double)g.Cout;
}

void conv_wino_bx6_transform_weights(const float* w, float* u, const ConvGeom& g, bool from_fwd_for_dgrad, hipStream_t s) {
    const int total = (g.Cin >> 3) * g.Cout;
    hipLaunchKernelGGL(wino_weights_bx6_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(u), g.Cin,
                       g.Cout, from_fwd_for_dgrad ? 1 : 0);
}

// n samples starting at x / y
void conv_wino_bx6_launch(const float* x, const float* u, const float* bias, float* y, const ConvGeom& g, int n, hipStream_t s,
                          float* stat_part, int stat_mode, const BnBwdFuse* bn_bwd) {
    WinoArgs a;
    a.x = x; a.u = u; a.bias = bias; a.y = y;
    a.N = n; a.H = g.H; a.W = g.W; a.Cin = g.Cin; a.Cout = g.Cout;
    a.TY = (g.H + 1) / 2;
    a.TX = (g.W + 1) / 2;
    a.rows = n * a.TY;
    a.nblocks = g.Cout / 64;
    a.nchunks = g.Cin / 16;
    a.inv_ty = 1.0f / (float)a.TY;
    a.stat_part = stat_part;
    a.stat_mode = stat_mode;
    a.bb = BnBwdFuse{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0};
    if (bn_bwd != nullptr && stat_part != nullptr) a.bb = *bn_bwd;
    const int btx = conv_wino_bx6_btx(g), bty = 64 / btx;
    a.txb = (a.TX + btx - 1) / btx;
    a.mblocks = ((a.rows + bty - 1) / bty) * a.txb;
    if (btx == 16)
        launch_bx6<16>(a, s);
    else if (btx == 8)
        launch_bx6<8>(a, s);
    else
        launch_bx6<4>(a, s);
}

}  // namespace l3
