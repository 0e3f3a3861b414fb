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
//   stage  = 16 input channels = one MFMA k-step; lane (tile l & 31, k-group g = l >> 5) owns channels 8 g .. 8 g + 7;
//   A      = the raw input pixels of the block's tiles, HBM -> LDS with buffer_load ... lds, double buffered, one image of
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

namespace l3 {

namespace {

template <int BTX>
struct Bx6Geom {
    static constexpr int BTY = 64 / BTX;
    static constexpr int PXH = BTX + 1;                         // 16-B slots per (input row, column parity)
    // row pitch in slots: lanes 0-15 of a ds_read_b128 group are 16 / BTX tile rows of BTX consecutive tiles, a tile row = two
    // input rows down; conflict free when 2 * PITCH = 0 (BTX 16), 8 (BTX 8), 4 (BTX 4) mod 16 (MI355X_MICROARCH.md, LDS)
    static constexpr int PITCH = BTX == 16 ? 40 : BTX == 8 ? 20 : 10;
    static constexpr int MAXB = BTX == 4 ? 2 : 1;               // image boundaries a block of BTY flat tile rows may cross
    static constexpr int IR = 2 * BTY + 2 + 2 * MAXB;           // input rows held
    static constexpr int PLANE = IR * PITCH;                    // slots per channel quad
    static constexpr int A_SLOTS = 4 * PLANE;
    static constexpr int A_PIECES = (A_SLOTS + 63) / 64;        // 1-KiB pieces: 30 / 25 / 24 ...
    static constexpr int A_BYTES = 32 * 1024;                   // ... of the 32 a buffer holds: every wave issues exactly two per stage,
                                                                // so that the stage loop's vmcnt counts are the same for every wave
    static constexpr int B_BYTES = 16 * 6 * 1024;               // 16 waves x (2 cout halves x 3 terms) pieces
    static constexpr int E_BYTES = 16 * 32 * 64 * 4;            // the output transform's exchange area (wino_common.h)
    static constexpr int LOOP_BYTES = 2 * A_BYTES + B_BYTES;
    static constexpr size_t LDS_BYTES = LOOP_BYTES > E_BYTES ? LOOP_BYTES : E_BYTES;
    static_assert(A_PIECES <= 32, "two A pieces per wave at most");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(2 * PITCH >= 2 * PXH, "row pitch");
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
    constexpr int BTY = G::BTY, PXH = G::PXH, PITCH = G::PITCH, PLANE = G::PLANE, A_SLOTS = G::A_SLOTS, A_PIECES = G::A_PIECES;
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
    const int nstage = a.nchunks;                          // Cin / 16
    const int u_stage_bytes = (a.Cout >> 5) * 3072;        // one stage of one position: Cout/32 cout halves x 3 KiB
    // ---- this wave's position: V = (d[ra][ca] + sa d[rb][ca]) + sb (d[ra][cb] + sa d[rb][cb]),  rows of B^T:
    //      0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3
    const int xi = wave >> 2, nu = wave & 3;
    const int ra = xi == 0 ? 0 : xi == 2 ? 2 : 1, rbw = xi == 0 ? 2 : xi == 1 ? 2 : xi == 2 ? 1 : 3;
    const int ca = nu == 0 ? 0 : nu == 2 ? 2 : 1, cbw = nu == 0 ? 2 : nu == 1 ? 2 : nu == 2 ? 1 : 3;

    for (int lt = blockIdx.x; lt < total_tiles; lt += (int)gridDim.x) {
    if (lt != (int)blockIdx.x) __syncthreads();          // the previous tile block's last LDS reads are done
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int logical = xcd_remap(lt, total_tiles);
    const int nb = logical % a.nblocks, mb = logical / a.nblocks;
    const int rb = mb / a.txb, cb = mb - rb * a.txb;
    const int R0 = rb * BTY, tx0 = cb * BTX, n0 = nb * 64;
    // the block's first image and how many of its tile rows lie in it
    const int img0 = R0 / a.TY, ty0 = R0 - img0 * a.TY;
    const int cnt0 = min(BTY, a.TY - ty0);
    const int seg0_rows = 2 * cnt0 + 2, seg_rows = 2 * a.TY + 2;

    // ---- A staging: pieces `wave` and `16 + wave`; slot -> (quad, input row, parity, column / 2) -> pixel --------
    unsigned avoff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int s = (wave + 16 * q) * 64 + lane;
        unsigned vo = 0x80000000u;
        if (s < A_SLOTS) {
            const int quad = s / PLANE, rem = s - quad * PLANE;
            const int lr = rem / PITCH, rem2 = rem - lr * PITCH;
            const int par = rem2 / PXH, pxh = rem2 - par * PXH;
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
            if (par < 2 && img < a.N && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W)
                vo = (unsigned)(((img * a.H + yy) * a.W + xx) * a.Cin * 4 + quad * 16);
        }
        avoff[q] = vo;
    }
    const unsigned bvoff = (unsigned)(lane * 16);
    const int bsbase = (wave * nstage * (a.Cout >> 5) + 2 * nb) * 3072;       // this wave's 6 KiB of stage 0

    // (stage == nstage: the requests behind the last stage.  They keep the loop body free of branches and the vmcnt counts the same
    //  in every stage; they go through a descriptor of zero records, so they move no data -- every lane is out of range and the
    //  buffer unit writes zeros, into a place nobody reads before it is requested again)
    auto issue_a = [&](int buf, int stage) {
        char* As = smem + buf * A_BYTES;
        const __amdgpu_buffer_rsrc_t srd = stage < nstage ? xsrd : nullsrd;
        const int asoff = stage * 64;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(As + wave * 1024), 16,
                                                 (int)avoff[0], asoff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(As + (wave + 16) * 1024), 16,
                                                 (int)avoff[1], asoff, 0, 0);
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
        a_off[i] = ((2 * grp * PLANE) + (2 * tr + 2 * crossed) * PITCH + tcol) * 16;
    }
    auto px = [&](int r, int c) { return (r * PITCH + (c & 1) * PXH + (c >> 1)) * 16; };
    const int o_aa = px(ra, ca), o_ba = px(rbw, ca), o_ab = px(ra, cbw), o_bb = px(rbw, cbw);
    const char* const b_rd = Bs + wave * 6144 + lane * 16;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

    auto stage_loop = [&](auto SA, auto SB) {
        constexpr bool PA = decltype(SA)::value, PB = decltype(SB)::value;
        for (int c = 0; c < nstage; ++c) {
            const char* As = smem + (c & 1) * A_BYTES;
            // A(c + 1): its buffer was last read in stage c - 1, which every wave left through the barrier below
            issue_a((c + 1) & 1, c + 1);
            // ---- V of the lane's 2 x 8 values, split into bf16 triples ----
            u32x4 ah[2], am[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int qh = 0; qh < 2; ++qh) {
                    // one group = the four raw pixels of four channels: 16 registers in flight, not 64 (the accumulators leave ~60)
                    __builtin_amdgcn_sched_barrier(0);
                    const char* p = As + a_off[i] + qh * (PLANE * 16);
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
            // ---- the six products per (tile half, cout half); B(c) jn = 0 was requested before B(c) jn = 1 before A(c + 1) ----
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                // outstanding, oldest first: [B0(c) x3] B1(c) x3, A(c + 1) x2, [B0(c + 1) x3 when jn = 1]: all but the youngest five
                wait_vm<5>();
                const u32x4 bh = *reinterpret_cast<const u32x4*>(b_rd + (jn * 3 + 0) * 1024);
                const u32x4 bm = *reinterpret_cast<const u32x4*>(b_rd + (jn * 3 + 1) * 1024);
                const u32x4 bl = *reinterpret_cast<const u32x4*>(b_rd + (jn * 3 + 2) * 1024);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the pieces are in registers: their place is free
                issue_b(c + 1, jn);
                const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bm = __builtin_bit_cast(bf16x8, bm), Bl = __builtin_bit_cast(bf16x8, bl);
                const bf16x8 Ah0 = __builtin_bit_cast(bf16x8, ah[0]), Am0 = __builtin_bit_cast(bf16x8, am[0]), Al0 = __builtin_bit_cast(bf16x8, al[0]);
                const bf16x8 Ah1 = __builtin_bit_cast(bf16x8, ah[1]), Am1 = __builtin_bit_cast(bf16x8, am[1]), Al1 = __builtin_bit_cast(bf16x8, al[1]);
                // smallest terms first; the two tile halves alternate, so an accumulator is touched every other MFMA
                f32x16 d0 = acc[0][jn], d1 = acc[1][jn];
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al0, Bh, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al1, Bh, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bl, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bl, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am0, Bm, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am1, Bm, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am0, Bh, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am1, Bh, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bm, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bm, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah0, Bh, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah1, Bh, d1, 0, 0, 0);
                acc[0][jn] = d0;
                acc[1][jn] = d1;
            }
            // A(c + 1) has landed (everything but the six B(c + 1) pieces), and every wave is done with A(c)
            __builtin_amdgcn_sched_barrier(0);
            wait_vm<6>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    issue_b(0, 0);
    issue_b(0, 1);
    issue_a(0, 0);
    // (only A must be visible to the other waves; the B pieces are waited for by their own wave in the loop)
    wait_vm<0>();
    __syncthreads();
    if (xi == 1) {
        if (nu == 1) stage_loop(TrueT{}, TrueT{}); else stage_loop(TrueT{}, FalseT{});
    } else {
        if (nu == 1) stage_loop(FalseT{}, TrueT{}); else stage_loop(FalseT{}, FalseT{});
    }
    __syncthreads();
    wino_output<BTX, SM>(a, acc, reinterpret_cast<float*>(smem), t, wave, lane, R0, tx0, n0, mb);
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
        if (btx == 4 ? TY < 8 : TY < bty) continue;
        const size_t wst = (size_t)((TX + btx - 1) / btx) * btx;            // flat rows pad only once per launch
        if (wst < waste) {
            waste = wst;
            best = btx;
        }
    }
    static const int force = l3_knob("L3_BX6_BTX") ? atoi(l3_knob("L3_BX6_BTX")) : 0;
    if ((force == 4 || force == 8 || force == 16) && (force == 4 ? TY >= 8 : TY >= 64 / force)) best = force;
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
