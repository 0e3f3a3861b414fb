// conv.hip -- fp32 convolution as implicit GEMM on the CDNA4 matrix cores.
//
// Replaces the TF/cuDNN Conv2D ops that keras instantiates for the 16 3x3 'same'
// convolutions of the L3 towers (reference: l3embedding/audio_model.py:376-432,
// l3embedding/vision_model.py:130-186), their data gradients and their weight
// gradients, plus the kapre DFT-as-conv (as a 1x1 conv over framed audio).
//
//   forward / dgrad:  Y[m][co] = sum_k A[m][k] * W[k][co]      m=(n,ho,wo)  k=(kh,kw,ci)
//   wgrad:           dW[k][co] = sum_m A[m][k] * dY[m][co]
// A is the im2col view of the NHWC input and is never materialised: tiles are
// gathered straight from HBM/L2 into LDS with zero fill for the halo.
//
// Matrix instruction: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/SIMD).  Operand
// fragments (MI355X guide section 3): A lane l holds A[i=l&31][k=l>>5], B lane l
// holds B[k=l>>5][j=l&31]; C/D reg r of lane l is row (r&3)+8*(r>>2)+4*(l>>5), col l&31.
// The reduction index of an MFMA may be any permutation as long as A and B agree,
// so the two half-waves read k and k+4: that lets each lane fetch 4 consecutive k
// of its A row with one ds_read_b128 (row stride padded 16->20 floats: conflict
// free for the b128 lane groups).
#include "kernels.h"
#include "device_common.h"
#include <cstdlib>
#include <mutex>

namespace l3 {



struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, padT, padL;
    int M, K, nkt, cpt;          // cpt: 16-channel chunks per tap (vector path)
    int mtiles, ntiles;
    int nvec;                    // Cout % 4 == 0
};


// Out-of-range elements (halo, M/N/K tails) are fetched from this zero page instead of
// being branched around or masked after the load: a branch around a load makes hipcc
// wait vmcnt(0) at every join, and a select after it pulls the wait in front of the
// MFMAs; either one serialises the prefetch (guide section 5, trap (c)).  With the
// pointer select the loaded registers go to LDS untouched, after the MFMA block.
__device__ __attribute__((aligned(16))) float g_zero_page[16];

template <int WAVES_M, int WAVES_N, int WT_M, int WT_N, int BKT, bool SMALLC, bool NVEC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    constexpr int BM = WAVES_M * WT_M, BN = WAVES_N * WT_N;
    constexpr int TM = WT_M / 32, TN = WT_N / 32;
    constexpr int ALD = BKT + 4;                   // padded A row: conflict-free ds_read_b128
    constexpr int TPR = BKT / 4;                   // threads (float4) per A row
    constexpr int RPP = 256 / TPR;                 // A rows per staging pass
    constexpr int A_ITERS = BM / RPP;
    constexpr int B_F4 = BKT * BN / 4;
    constexpr int B_ITERS = (B_F4 + 255) / 256;
    constexpr int A_TILE = BM * ALD, B_TILE = BKT * BN;
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_TILE + B_TILE)];
    float* As = smem;
    float* Bs = smem + 2 * A_TILE;

    const int t = threadIdx.x;
    const int logical = xcd_remap(blockIdx.x, a.mtiles * a.ntiles);
    const int nt = logical % a.ntiles, mt = logical / a.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // per-thread A rows (fixed across the k loop): element offset of tap (0,0) and a
    // validity bit per filter tap, so the k loop only adds a wave-uniform tap offset
    const int cv = t % TPR;
    const int arow = t / TPR;
    int64_t rowoff[A_ITERS];
    unsigned tapmask[A_ITERS];
    int hi0[A_ITERS], wi0[A_ITERS];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int m = m0 + arow + RPP * i;
        unsigned mask = 0;
        int64_t off = 0;
        hi0[i] = -100000;
        wi0[i] = 0;
        if (m < a.M) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            const int h0 = ho - a.padT, w0 = wo - a.padL;
            hi0[i] = h0;
            wi0[i] = w0;
            off = ((int64_t)(n * a.H + h0) * a.W + w0) * a.Cin + cv * 4;
            for (int tap = 0; tap < a.KH * a.KW; ++tap) {
                const int dh = tap / a.KW, dw = tap - dh * a.KW;
                if ((unsigned)(h0 + dh) < (unsigned)a.H && (unsigned)(w0 + dw) < (unsigned)a.W) mask |= 1u << tap;
            }
        }
        rowoff[i] = off;
        tapmask[i] = mask;
    }

    f32x4 areg[A_ITERS];
    f32x4 breg[B_ITERS];

    // running (wave-uniform) position of the NEXT k-tile to load: filter tap and channel chunk
    // advance incrementally, so the loop carries no integer division
    int ld_tap = 0, ld_c0 = 0, ld_dh = 0, ld_dw = 0;
    // per-thread B source pointers advance by BKT rows of the filter matrix per k-tile
    const float* bptr[B_ITERS];
    bool bok[B_ITERS];
#pragma unroll
    for (int j = 0; j < B_ITERS; ++j) {
        const int f = t + 256 * j;
        const int row = f / (BN / 4), c4 = f - row * (BN / 4);
        const int n = n0 + c4 * 4;
        bok[j] = (B_F4 % 256 == 0 || f < B_F4) && n < a.Cout;
        bptr[j] = a.w + ((size_t)row * a.Cout + (bok[j] ? n : 0));
    }
    const size_t bstep = (size_t)BKT * a.Cout;

    auto load_tiles = [&](int kt) {
        if constexpr (!SMALLC) {
            const int64_t tapoff = (int64_t)(ld_dh * a.W + ld_dw) * a.Cin + ld_c0;     // wave-uniform
            const unsigned tapbit = 1u << ld_tap;
#pragma unroll
            for (int i = 0; i < A_ITERS; ++i) {
                const bool ok = (tapmask[i] & tapbit) != 0;
                const float* p = ok ? a.x + (rowoff[i] + tapoff) : g_zero_page;
                areg[i] = *reinterpret_cast<const f32x4*>(p);
            }
            ld_c0 += BKT;
            if (ld_c0 >= a.Cin) {
                ld_c0 = 0;
                ++ld_tap;
                if (++ld_dw == a.KW) {
                    ld_dw = 0;
                    ++ld_dh;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_ITERS; ++i) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = kt * BKT + cv * 4 + e;
                    const int tap = k / a.Cin, ci = k - tap * a.Cin;
                    const int dh = tap / a.KW, dw = tap - dh * a.KW;
                    const bool ok = k < a.K && ((tapmask[i] >> tap) & 1u);
                    v[e] = *(ok ? a.x + (rowoff[i] - cv * 4 + (int64_t)(dh * a.W + dw) * a.Cin + ci) : g_zero_page);
                }
                areg[i] = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
#pragma unroll
        for (int j = 0; j < B_ITERS; ++j) {
            const int f = t + 256 * j;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (B_F4 % 256 == 0 || f < B_F4) {       // compile-time / wave-uniform
                if constexpr (NVEC && !SMALLC) {
                    v = *reinterpret_cast<const f32x4*>(bok[j] ? bptr[j] : g_zero_page);
                    bptr[j] += bstep;
                } else {
                    const int row = f / (BN / 4), c4 = f - row * (BN / 4);
                    const int k = kt * BKT + row, n = n0 + c4 * 4;
                    const bool kok = k < a.K;
                    const size_t base = kok ? (size_t)k * a.Cout : 0;
                    const bool o0 = kok && n + 0 < a.Cout, o1 = kok && n + 1 < a.Cout;
                    const bool o2 = kok && n + 2 < a.Cout, o3 = kok && n + 3 < a.Cout;
                    v = f32x4{*(o0 ? a.w + base + n + 0 : g_zero_page), *(o1 ? a.w + base + n + 1 : g_zero_page),
                              *(o2 ? a.w + base + n + 2 : g_zero_page), *(o3 ? a.w + base + n + 3 : g_zero_page)};
                }
            }
            breg[j] = v;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int r = arow + RPP * i;
            *reinterpret_cast<f32x4*>(&As[buf * A_TILE + r * ALD + cv * 4]) = areg[i];
        }
#pragma unroll
        for (int j = 0; j < B_ITERS; ++j) {
            const int f = t + 256 * j;
            if (B_F4 % 256 == 0 || f < B_F4)
                *reinterpret_cast<f32x4*>(&Bs[buf * B_TILE + f * 4]) = breg[j];
        }
    };

    const int wave = t >> 6, lane = t & 63;
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int l31 = lane & 31, hi32 = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    for (int kt = 0; kt < a.nkt; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < a.nkt;
        if (more) load_tiles(kt + 1);
        const float* Ab = As + buf * A_TILE + (wm * WT_M + l31) * ALD + hi32 * 4;
        const float* Bb = Bs + buf * B_TILE + hi32 * 4 * BN + wn * WT_N + l31;
#pragma unroll
        for (int q = 0; q < BKT / 8; ++q) {
            f32x4 av[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                av[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * ALD + q * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float bv[TN];
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) bv[jn] = Bb[(q * 8 + j) * BN + jn * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float ae = j == 0 ? av[i].x : j == 1 ? av[i].y : j == 2 ? av[i].z : av[i].w;
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, bv[jn], acc[i][jn], 0, 0, 0);
                }
            }
        }
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // epilogue: + bias, store (lanes 0..31 write 128 contiguous bytes per row)
    const bool interior = (m0 + BM <= a.M) && (n0 + BN <= a.Cout);   // block-uniform
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int n = n0 + wn * WT_N + jn * 32 + l31;
        const bool nok = n < a.Cout;
        const float bz = (a.bias != nullptr && nok) ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * WT_M + i * 32 + 4 * hi32;
            float* yp = a.y + (size_t)mb * a.Cout + n;
            if (interior) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    yp[(size_t)((r & 3) + 8 * (r >> 2)) * a.Cout] = acc[i][jn][r] + bz;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = (r & 3) + 8 * (r >> 2);
                    if (nok && mb + dm < a.M) yp[(size_t)dm * a.Cout] = acc[i][jn][r] + bz;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// same implicit GEMM with direct global->LDS staging (global_load_lds_dwordx4)
// ---------------------------------------------------------------------------------
// No staging VGPRs and no ds_write pass: tiles land in LDS asynchronously while the MFMAs
// of the current stage run, and the register budget drops below 128 so four blocks fit a
// CU.  The LDS image of a wave-instruction is lane-linear (base + lane*16 B), so A rows are
// unpadded (64 B) and bank conflicts are avoided by XOR-swizzling the 16-B chunk index with
// (row>>2)&3 on the SOURCE address and on the read address (guide 5.4 rule 21).
//
// SRD = true addresses both tiles through buffer resources (buffer_load_dwordx4 ... lds): the
// filter-tap / channel-chunk displacement rides in the SCALAR soffset operand, a lane's pixel
// offset is a loop-invariant 32-bit voffset, and an invalid (padding) tap is a voffset with
// bit 31 set -- out of range, which the buffer unit answers with zeros.  That removes the
// 64-bit pointer arithmetic and the zero-page selects from the loop: on gfx950 every VALU
// instruction issued next to fp32 MFMAs costs ~4.6 cycles of MFMA issue while SALU is free
// (scripts/mfma_mix.hip), so the loop's VALU count is what separates it from peak.  Needs
// the activation tensor (+ one padding row) below 2 GiB; larger ones take SRD = false.
template <int WAVES_M, int WAVES_N, int WT_M, int WT_N, int MINW, bool SRD>
__global__ __launch_bounds__(256, MINW) void conv_igemm_glds_kernel(ConvArgs a) {
    constexpr int BKT = 16;
    constexpr int BM = WAVES_M * WT_M, BN = WAVES_N * WT_N;
    constexpr int TM = WT_M / 32, TN = WT_N / 32;
    constexpr int A_TILE = BM * BKT, B_TILE = BKT * BN;
    constexpr int A_PW = BM / 64;          // 1-KiB A pieces per wave (16 rows each)
    constexpr int B_PW = BN / 64;          // 1-KiB B pieces per wave
    static_assert(BN % 64 == 0, "glds path needs BN multiple of 64");
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_TILE + B_TILE)];
    float* As = smem;
    float* Bs = smem + 2 * A_TILE;

    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int logical = xcd_remap(blockIdx.x, a.mtiles * a.ntiles);
    const int nt = logical % a.ntiles, mt = logical / a.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // A pieces: piece p = wave*A_PW + i covers tile rows 16p..16p+15; lane -> (row, physical chunk)
    int64_t rowoff[A_PW];
    unsigned tapmask[A_PW];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
        const int r = (wave * A_PW + i) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);          // logical 16-B chunk stored at this slot
        const int m = m0 + r;
        unsigned mask = 0;
        int64_t off = 0;
        if (m < a.M) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            const int h0 = ho - a.padT, w0 = wo - a.padL;
            off = ((int64_t)(n * a.H + h0) * a.W + w0) * a.Cin + c * 4;
            for (int tap = 0; tap < a.KH * a.KW; ++tap) {
                const int dh = tap / a.KW, dw = tap - dh * a.KW;
                if ((unsigned)(h0 + dh) < (unsigned)a.H && (unsigned)(w0 + dw) < (unsigned)a.W) mask |= 1u << tap;
            }
        }
        rowoff[i] = off;
        tapmask[i] = mask;
    }
    const float* bptr[B_PW];
    bool bok[B_PW];
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        const int f = (wave * B_PW + j) * 256 + lane * 4;    // float index inside the B tile
        const int row = f / BN, col = f - row * BN;
        bok[j] = n0 + col < a.Cout;
        bptr[j] = a.w + ((size_t)row * a.Cout + (bok[j] ? n0 + col : 0));
    }
    // SRD mode: byte offsets relative to (x - margin), margin = the most negative pixel offset
    const int margin = (a.padT * a.W + a.padL) * a.Cin * 4;
    unsigned avoff[A_PW], anot[A_PW], bvoff[B_PW];
    __amdgpu_buffer_rsrc_t xsrd, wsrd;
    if constexpr (SRD) {
#pragma unroll
        for (int i = 0; i < A_PW; ++i) {
            avoff[i] = (unsigned)((int)(rowoff[i] * 4) + margin);
            anot[i] = ~tapmask[i];
        }
#pragma unroll
        for (int j = 0; j < B_PW; ++j)
            bvoff[j] = bok[j] ? (unsigned)((bptr[j] - a.w) * 4) : 0x80000000u;
        xsrd = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.x - margin), 0,
                                                 (int)((size_t)a.N * a.H * a.W * a.Cin * 4 + margin), 0x00020000);
        wsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.K * a.Cout * 4, 0x00020000);
    }
    int ld_tap = 0, ld_c0 = 0, ld_dh = 0, ld_dw = 0;

    auto issue_tiles = [&](int buf) {
        // k order: channel chunk OUTER, filter tap INNER -- the nine taps of one 16-channel
        // chunk re-touch the same cache lines back to back (L1/L2 hits) instead of sweeping all
        // channels between two visits of a pixel
        const int64_t tapoff = (int64_t)(ld_dh * a.W + ld_dw) * a.Cin + ld_c0;
        const unsigned tapbit = 1u << ld_tap;
        const size_t brow = (size_t)(ld_tap * a.Cin + ld_c0) * a.Cout;
        if constexpr (SRD) {
            const int asoff = (int)tapoff * 4, bsoff = (int)brow * 4;
#pragma unroll
            for (int i = 0; i < A_PW; ++i) {
                // (~mask >> tap) << 31 | voffset : bit 31 set <=> padding tap <=> out of range
                const unsigned vo = ((anot[i] >> ld_tap) << 31) | avoff[i];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    xsrd, (__attribute__((address_space(3))) void*)(As + buf * A_TILE + (wave * A_PW + i) * 256), 16,
                    (int)vo, asoff, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < B_PW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    wsrd, (__attribute__((address_space(3))) void*)(Bs + buf * B_TILE + (wave * B_PW + j) * 256), 16,
                    (int)bvoff[j], bsoff, 0, 0);
        } else {
#pragma unroll
        for (int i = 0; i < A_PW; ++i) {
            const bool ok = (tapmask[i] & tapbit) != 0;
            const float* p = ok ? a.x + (rowoff[i] + tapoff) : g_zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)(As + buf * A_TILE + (wave * A_PW + i) * 256),
                                             16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_PW; ++j) {
            const float* p = bok[j] ? bptr[j] + brow : g_zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)(Bs + buf * B_TILE + (wave * B_PW + j) * 256),
                                             16, 0, 0);
        }
        }
        ++ld_tap;
        if (++ld_dw == a.KW) {
            ld_dw = 0;
            if (++ld_dh == a.KH) {
                ld_dh = 0;
                ld_tap = 0;
                ld_c0 += BKT;
            }
        }
    };

    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int l31 = lane & 31, hi32 = lane >> 5;
    const int swz = (l31 >> 2) & 3;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue_tiles(0);
    __syncthreads();

    // one k-tile out of stage buffer `buf` (a compile-time constant in both call sites, so every
    // ds_read address is lane base + immediate)
    auto compute = [&](int buf) {
        const float* Ab = As + buf * A_TILE + (wm * WT_M + l31) * BKT;
        const float* Bb = Bs + buf * B_TILE + hi32 * 4 * BN + wn * WT_N + l31;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x4 av[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                av[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * BKT + (((q * 2 + hi32) ^ swz) * 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float bv[TN];
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) bv[jn] = Bb[(q * 8 + j) * BN + jn * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float ae = j == 0 ? av[i].x : j == 1 ? av[i].y : j == 2 ? av[i].z : av[i].w;
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, bv[jn], acc[i][jn], 0, 0, 0);
                }
            }
        }
    };
    for (int kt = 0; kt < a.nkt; kt += 2) {
        if (kt + 1 < a.nkt) issue_tiles(1);
        compute(0);
        __syncthreads();      // hipcc drains the LDS-DMA queue (vmcnt(0)) in front of the barrier
        if (kt + 1 < a.nkt) {
            if (kt + 2 < a.nkt) issue_tiles(0);
            compute(1);
            __syncthreads();
        }
    }

    // epilogue: transpose each wave's 64 x WT_N accumulator block through LDS (32 rows at a
    // time; the stage buffers are free after the final barrier) so that rows leave as 16-B
    // stores -- 4x fewer VMEM instructions than per-accumulator dword stores
    static_assert(WT_M == 64 && (WT_N == 64), "epilogue assumes 64x64 wave tiles");
    static_assert(2 * (A_TILE + B_TILE) >= 4 * 32 * 64, "stage buffers too small for the epilogue");
    float* Es = smem + wave * (32 * 64);
    const int n_base = n0 + wn * WT_N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Es[((r & 3) + 8 * (r >> 2) + 4 * hi32) * 64 + jn * 32 + l31] = acc[i][jn][r];
        // wave-private region: only this wave's lanes exchange data
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = p * 4 + (lane >> 4), c4 = (lane & 15) * 4;
            const int m = m0 + wm * WT_M + i * 32 + row, n = n_base + c4;
            f32x4 v = *reinterpret_cast<const f32x4*>(Es + row * 64 + c4);
            if (m < a.M && n < a.Cout) {
                if (a.bias != nullptr) v += *reinterpret_cast<const f32x4*>(a.bias + n);
                *reinterpret_cast<f32x4*>(a.y + (size_t)m * a.Cout + n) = v;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

template <int WAVES_M, int WAVES_N, int WT_M, int WT_N>
static void launch_igemm(ConvArgs a, const ConvGeom& g, hipStream_t s) {
    constexpr int BM = WAVES_M * WT_M, BN = WAVES_N * WT_N;
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.Cout + BN - 1) / BN;
    dim3 grid(a.mtiles * a.ntiles), block(256);
    static const int force_bk = l3_knob("L3_IGEMM_BK") ? atoi(l3_knob("L3_IGEMM_BK")) : 0;
    const bool smallc = (g.Cin % 16) != 0;
    const int bk = smallc ? 16 : (force_bk ? force_bk : 16);   // BK=32 halves occupancy (LDS) and measured 6 % slower
    a.cpt = smallc ? 0 : g.Cin / bk;
    a.nkt = (a.K + bk - 1) / bk;
#define L3_IG(BK_, SC_, NV_) \
    hipLaunchKernelGGL((conv_igemm_kernel<WAVES_M, WAVES_N, WT_M, WT_N, BK_, SC_, NV_>), grid, block, 0, s, a)
    static const int use_glds = l3_knob("L3_IGEMM_GLDS") ? atoi(l3_knob("L3_IGEMM_GLDS")) : 1;
    if constexpr (BN % 64 == 0) {
        if (use_glds && !smallc && a.nvec && bk == 16) {
            static const int use_srd = l3_knob("L3_IGEMM_SRD") ? atoi(l3_knob("L3_IGEMM_SRD")) : 1;
            const size_t xbytes = (size_t)a.N * a.H * a.W * a.Cin * 4 + (size_t)(a.padT * a.W + a.padL) * a.Cin * 4;
            constexpr int MINW = WT_M * WT_N > 4096 ? 3 : 4;
            if (use_srd && xbytes < (1ull << 31) && (size_t)a.K * a.Cout * 4 < (1ull << 31))
                hipLaunchKernelGGL((conv_igemm_glds_kernel<WAVES_M, WAVES_N, WT_M, WT_N, MINW, true>), grid, block, 0, s, a);
            else
                hipLaunchKernelGGL((conv_igemm_glds_kernel<WAVES_M, WAVES_N, WT_M, WT_N, MINW, false>), grid, block, 0, s, a);
            return;
        }
    }
    if (smallc) {
        if (a.nvec) L3_IG(16, true, true); else L3_IG(16, true, false);
    } else if (bk == 32 && g.Cin % 32 == 0) {
        if (a.nvec) L3_IG(32, false, true); else L3_IG(32, false, false);
    } else {
        if (a.nvec) L3_IG(16, false, true); else L3_IG(16, false, false);
    }
#undef L3_IG
}

void conv_fwd(const float* x, const float* w, const float* bias, float* y, const ConvGeom& g,
              hipStream_t s, const float* wino_u, float* bn_part, int bn_mode, const BnBwdFuse* bn_bwd) {
    if (wino_u != nullptr && conv_wino_ok(g)) {
        conv_wino_fwd(x, wino_u, bias, y, g, s, bn_part, bn_mode, bn_bwd);
        return;
    }
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    a.N = g.N; a.H = g.H; a.W = g.W; a.Cin = g.Cin; a.Ho = g.Ho; a.Wo = g.Wo; a.Cout = g.Cout;
    a.KH = g.KH; a.KW = g.KW; a.padT = g.padT; a.padL = g.padL;
    a.M = g.N * g.Ho * g.Wo;
    a.K = g.KH * g.KW * g.Cin;
    a.cpt = 0;
    a.nkt = 0;
    a.nvec = (g.Cout % 4) == 0;
    a.mtiles = a.ntiles = 0;
    if (g.Cout > 64)
        launch_igemm<2, 2, 64, 64>(a, g, s);     // 128 x 128
    else if (g.Cout > 32)
        launch_igemm<4, 1, 64, 64>(a, g, s);     // 256 x 64
    else
        launch_igemm<4, 1, 64, 32>(a, g, s);     // 256 x 32
}

// ---------------------------------------------------------------------------------
// data gradient of the FIRST conv of a tower (input has 1 or 3 channels, 64 filters)
// ---------------------------------------------------------------------------------
// Needed only because the input BatchNorm's gamma/beta are trainable (audio_model.py:370,
// vision_model.py:124).  As an implicit GEMM it wastes a 32-wide MFMA tile on 1-3 output
// channels; here lanes map to the 64 dY channels, a 3x3 register window slides along the
// row (3 coalesced 256-B loads per pixel) and a wave reduction produces each output value.
template <int CIN>
__global__ __launch_bounds__(256) void conv_dgrad_small_kernel(const float* dy, const float* w, float* dx, int N,
                                                               int H, int W, int strip) {
    // w is the forward filter [3][3][CIN][64]; dx[n,h,x,ci] = sum_{kh,kw,co} dy[n,h+1-kh,x+1-kw,co]*w[kh][kw][ci][co]
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int strips = (W + strip - 1) / strip;
    const int total = N * H * strips;
    if (wave_global >= total) return;
    const int row = wave_global / strips, st = wave_global - row * strips;
    const int n = row / H, h = row - n * H;
    const int x0 = st * strip, x1 = min(W, x0 + strip);
    float wt[9][CIN];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) wt[t9][ci] = w[(t9 * CIN + ci) * 64 + lane];
    // window rows r=0..2 hold dy rows h+1, h, h-1 (kh = 0,1,2); columns c=0..2 hold x+1, x, x-1
    const float* rp[3];
    bool rok[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int hh = h + 1 - r;
        rok[r] = (unsigned)hh < (unsigned)H;
        rp[r] = dy + ((size_t)(n * H + (rok[r] ? hh : 0)) * W) * 64 + lane;
    }
    float win[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        win[r][2] = (rok[r] && x0 - 1 >= 0) ? rp[r][(size_t)(x0 - 1) * 64] : 0.f;
        win[r][1] = rok[r] ? rp[r][(size_t)x0 * 64] : 0.f;
    }
    for (int x = x0; x < x1; ++x) {
#pragma unroll
        for (int r = 0; r < 3; ++r) win[r][0] = (rok[r] && x + 1 < W) ? rp[r][(size_t)(x + 1) * 64] : 0.f;
        float acc[CIN];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) acc[ci] = fmaf(win[r][c], wt[r * 3 + c][ci], acc[ci]);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            float v = acc[ci];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            acc[ci] = v;
        }
        if (lane == 0) {
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) dx[((size_t)(n * H + h) * W + x) * CIN + ci] = acc[ci];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            win[r][2] = win[r][1];
            win[r][1] = win[r][0];
        }
    }
}

// true if handled (3x3 'same', stride 1, 64 filters, 1 or 3 input channels)
bool conv_dgrad_small(const float* dy, const float* w, float* dx, const ConvGeom& g, hipStream_t s) {
    if (!(g.KH == 3 && g.KW == 3 && g.padT == 1 && g.padL == 1 && g.Cout == 64 && g.Ho == g.H && g.Wo == g.W &&
          (g.Cin == 1 || g.Cin == 3)))
        return false;
    const int strip = 32;
    const int strips = (g.W + strip - 1) / strip;
    const int waves = g.N * g.H * strips;
    const dim3 grid((waves + 3) / 4), block(256);
    if (g.Cin == 1)
        hipLaunchKernelGGL((conv_dgrad_small_kernel<1>), grid, block, 0, s, dy, w, dx, g.N, g.H, g.W, strip);
    else
        hipLaunchKernelGGL((conv_dgrad_small_kernel<3>), grid, block, 0, s, dy, w, dx, g.N, g.H, g.W, strip);
    return true;
}

__global__ void flip_weights_kernel(const float* w, float* wt, int KH, int KW, int Cin, int Cout) {
    const int total = KH * KW * Cin * Cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        // i indexes wt[kh][kw][co][ci]
        const int ci = i % Cin;
        int r = i / Cin;
        const int co = r % Cout;
        r /= Cout;
        const int kw = r % KW, kh = r / KW;
        wt[i] = w[(((KH - 1 - kh) * KW + (KW - 1 - kw)) * Cin + ci) * Cout + co];
    }
}

void conv_flip_weights(const float* w, float* wt, int KH, int KW, int Cin, int Cout, hipStream_t s) {
    const int total = KH * KW * Cin * Cout;
    const int blocks = (total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048;
    hipLaunchKernelGGL(flip_weights_kernel, dim3(blocks), dim3(256), 0, s, w, wt, KH, KW, Cin, Cout);
}

// ---------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------
struct WgradArgs {
    const float* x;
    const float* dy;
    float* part;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, padT, padL;
    int M, K, ktiles, ntiles, splits, m_per_split;
    int nvec;
};

static constexpr int WG_MC = 16;   // m rows per LDS stage

template <int TK, int TN, bool SMALLC, bool NVEC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    constexpr int WK = TK / 2, WN = TN / 2;         // wave tile (2x2 waves)
    constexpr int TKm = WK / 32, TNn = WN / 32;
    constexpr int A_F4 = WG_MC * TK / 4, D_F4 = WG_MC * TN / 4;
    constexpr int A_ITERS = A_F4 / 256, D_ITERS = D_F4 / 256;
    static_assert(A_F4 % 256 == 0 && D_F4 % 256 == 0, "tile/thread mismatch");
    constexpr int A_TILE = WG_MC * TK, D_TILE = WG_MC * TN;
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_TILE + D_TILE)];
    float* As = smem;
    float* Ds = smem + 2 * A_TILE;

    const int t = threadIdx.x;
    const int tile = blockIdx.x;
    const int nt = tile % a.ntiles, ktile = tile / a.ntiles;
    const int sp = blockIdx.y;
    const int n0 = nt * TN;
    const int k0 = ktile * TK;
    int dh = 0, dw = 0, c0 = 0;
    if constexpr (!SMALLC) {
        const int tap = k0 / a.Cin;
        c0 = k0 - tap * a.Cin;
        dh = tap / a.KW;
        dw = tap - dh * a.KW;
    }
    const int m_begin = sp * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);
    const int HoWo = a.Ho * a.Wo;

    f32x4 areg[A_ITERS], dreg[D_ITERS];

    auto load_chunk = [&](int mb) {
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int f = t + 256 * i;
            const int row = f / (TK / 4), c4 = f - row * (TK / 4);
            const int m = mb + row;
            f32x4 v;
            {
                const bool mok = m < m_end;
                const int mm = mok ? m : m_begin;
                const int n = mm / HoWo, rem = mm - n * HoWo;
                const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
                if constexpr (!SMALLC) {
                    const int hi = ho - a.padT + dh, wi = wo - a.padL + dw;
                    const bool ok = mok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                    v = *reinterpret_cast<const f32x4*>(
                        ok ? a.x + ((size_t)((n * a.H + hi) * a.W + wi) * a.Cin + c0 + c4 * 4) : g_zero_page);
                } else {
                    float e4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = k0 + c4 * 4 + e;
                        const int tap = k / a.Cin, ci = k - tap * a.Cin;
                        const int ddh = tap / a.KW, ddw = tap - ddh * a.KW;
                        const int hi = ho - a.padT + ddh, wi = wo - a.padL + ddw;
                        const bool ok = mok && k < a.K && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                        e4[e] = *(ok ? a.x + ((size_t)((n * a.H + hi) * a.W + wi) * a.Cin + ci) : g_zero_page);
                    }
                    v = f32x4{e4[0], e4[1], e4[2], e4[3]};
                }
            }
            areg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < D_ITERS; ++i) {
            const int f = t + 256 * i;
            const int row = f / (TN / 4), c4 = f - row * (TN / 4);
            const int m = mb + row, n = n0 + c4 * 4;
            f32x4 v;
            {
                const bool mok = m < m_end;
                const size_t base = mok ? (size_t)m * a.Cout : 0;
                if constexpr (NVEC) {
                    const bool ok = mok && n < a.Cout;
                    v = *reinterpret_cast<const f32x4*>(ok ? a.dy + base + n : g_zero_page);
                } else {
                    const bool o0 = mok && n + 0 < a.Cout, o1 = mok && n + 1 < a.Cout;
                    const bool o2 = mok && n + 2 < a.Cout, o3 = mok && n + 3 < a.Cout;
                    v = f32x4{*(o0 ? a.dy + base + n + 0 : g_zero_page), *(o1 ? a.dy + base + n + 1 : g_zero_page),
                              *(o2 ? a.dy + base + n + 2 : g_zero_page), *(o3 ? a.dy + base + n + 3 : g_zero_page)};
                }
            }
            dreg[i] = v;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i)
            *reinterpret_cast<f32x4*>(&As[buf * A_TILE + (t + 256 * i) * 4]) = areg[i];
#pragma unroll
        for (int i = 0; i < D_ITERS; ++i)
            *reinterpret_cast<f32x4*>(&Ds[buf * D_TILE + (t + 256 * i) * 4]) = dreg[i];
    };

    const int wave = t >> 6, lane = t & 63;
    const int wk = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi32 = lane >> 5;

    f32x16 acc[TKm][TNn];
#pragma unroll
    for (int i = 0; i < TKm; ++i)
#pragma unroll
        for (int j = 0; j < TNn; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (m_end - m_begin + WG_MC - 1) / WG_MC;
    if (nchunks > 0) {
        load_chunk(m_begin);
        store_chunk(0);
    }
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < nchunks;
        if (more) load_chunk(m_begin + (ch + 1) * WG_MC);
        const float* Ab = As + buf * A_TILE + hi32 * TK + wk * WK + l31;
        const float* Db = Ds + buf * D_TILE + hi32 * TN + wn * WN + l31;
#pragma unroll
        for (int s2 = 0; s2 < WG_MC / 2; ++s2) {
            float av[TKm], bv[TNn];
#pragma unroll
            for (int i = 0; i < TKm; ++i) av[i] = Ab[s2 * 2 * TK + i * 32];
#pragma unroll
            for (int j = 0; j < TNn; ++j) bv[j] = Db[s2 * 2 * TN + j * 32];
#pragma unroll
            for (int i = 0; i < TKm; ++i)
#pragma unroll
                for (int j = 0; j < TNn; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }

    float* out = a.part + (size_t)sp * a.K * a.Cout;
#pragma unroll
    for (int j = 0; j < TNn; ++j) {
        const int n = n0 + wn * WN + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < TKm; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wk * WK + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi32;
                if (n < a.Cout && k < a.K) out[(size_t)k * a.Cout + n] = acc[i][j][r];
            }
    }
}

__global__ void wgrad_reduce_kernel(const float* part, float* dw, int64_t n, int splits) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int sp = 0; sp < splits; ++sp) s += part[(int64_t)sp * n + i];
        dw[i] = s;
    }
}

// dw = sum over split-K partials, float4 wide and SG-way parallel over the splits (group g adds
// splits g, g+SG, ... in order; the SG group sums are combined in group order through LDS), so the
// result does not depend on scheduling.  n % 4 == 0.
template <int SG>
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const f32x4* __restrict__ part, f32x4* __restrict__ dw,
                                                            int64_t n4, int splits) {
    constexpr int QPB = 256 / SG;                      // float4 columns per block
    __shared__ f32x4 red[256];
    const int q = threadIdx.x % QPB, g = threadIdx.x / QPB;
    const int64_t i = (int64_t)blockIdx.x * QPB + q;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4)
        for (int sp = g; sp < splits; sp += SG) s += __builtin_nontemporal_load(part + (int64_t)sp * n4 + i);
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && i < n4) {
#pragma unroll
        for (int k = 1; k < SG; ++k) s += red[k * QPB + q];
        dw[i] = s;
    }
}

static void wgrad_reduce(const float* part, float* dw, int64_t n, int splits, hipStream_t s) {
    if (n % 4 == 0 && splits > 1) {
        const int64_t n4 = n / 4;
        if (n4 < 1024 && splits >= 256)          // the first layer: a 9 x Cin x 64 filter, a thousand partials
            hipLaunchKernelGGL(wgrad_reduce4_kernel<64>, dim3((unsigned)((n4 + 3) / 4)), dim3(256), 0, s,
                               reinterpret_cast<const f32x4*>(part), reinterpret_cast<f32x4*>(dw), n4, splits);
        else if (n4 / 64 >= 1024 || splits < 16)
            hipLaunchKernelGGL(wgrad_reduce4_kernel<4>, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, s,
                               reinterpret_cast<const f32x4*>(part), reinterpret_cast<f32x4*>(dw), n4, splits);
        else
            hipLaunchKernelGGL(wgrad_reduce4_kernel<16>, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, s,
                               reinterpret_cast<const f32x4*>(part), reinterpret_cast<f32x4*>(dw), n4, splits);
        return;
    }
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, part, dw, n, splits);
}

// ---------------------------------------------------------------------------------
// weight gradient, all nine taps of a 3x3 'same' convolution in one block
// ---------------------------------------------------------------------------------
// The per-tap kernel above re-reads every activation and every output-gradient row nine
// times (once per tap block), which makes the big early layers L2/HBM-bound.  Here one
// block owns a (64 ci x 64 co) slice of dW for ALL nine taps: per 4x4 output patch it
// stages the 6x6 input halo (36 pixel rows x 64 ch) and the 16 dY rows once and issues
// 9 taps x 8 k-steps of MFMAs per wave (wave = 32 ci x 32 co x 9 taps = 144 accumulators).
// Bytes staged per MFMA drop 4.7x versus the per-tap kernel.
struct Wgrad9Args {
    const float* x;
    const float* dy;
    float* part;
    int N, H, W, Cin, Cout;
    int co_tiles, tiles, ph, pw;       // patches per image: ph x pw
    int npatch, per_split, splits;
};

__global__ __launch_bounds__(256) void conv_wgrad9_kernel(Wgrad9Args a) {
    constexpr int A_TILE = 36 * 64, D_TILE = 16 * 64;
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_TILE + D_TILE)];
    float* As = smem;
    float* Ds = smem + 2 * A_TILE;
    const int t = threadIdx.x;
    const int logical = xcd_remap(blockIdx.x, a.tiles * a.splits);
    const int sp = logical / a.tiles, tile = logical - sp * a.tiles;
    const int cit = tile / a.co_tiles, cot = tile - cit * a.co_tiles;
    const int ci0 = cit * 64, co0 = cot * 64;
    const int p_begin = sp * a.per_split;
    const int p_end = min(a.npatch, p_begin + a.per_split);

    // fixed per-thread staging slots: halo float4 f -> (pixel-in-halo, channel quad)
    int hpix[3], hc4[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int f = t + 256 * i;
        hpix[i] = f >> 4;          // 0..35 (valid when f < 576)
        hc4[i] = f & 15;
    }
    const int dpix = t >> 4, dc4 = t & 15;     // dY: 16 pixels x 16 quads
    const int dpy = dpix >> 2, dpx = dpix & 3;

    f32x4 areg[3], dreg;
    auto load_patch = [&](int pidx) {
        const int per_img = a.ph * a.pw;
        const int n = pidx / per_img, rem = pidx - n * per_img;
        const int py = rem / a.pw, px = rem - py * a.pw;
        const int h0 = py * 4, w0 = px * 4;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int hy = hpix[i] / 6, hx = hpix[i] - hy * 6;
            const int hi = h0 + hy - 1, wi = w0 + hx - 1;
            const bool ok = (i < 2 || t + 512 < 576) && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            const float* p = ok ? a.x + ((size_t)((n * a.H + hi) * a.W + wi) * a.Cin + ci0 + hc4[i] * 4) : g_zero_page;
            areg[i] = *reinterpret_cast<const f32x4*>(p);
        }
        {
            const int ho = h0 + dpy, wo = w0 + dpx;
            const bool ok = ho < a.H && wo < a.W;
            const float* p = ok ? a.dy + ((size_t)((n * a.H + ho) * a.W + wo) * a.Cout + co0 + dc4 * 4) : g_zero_page;
            dreg = *reinterpret_cast<const f32x4*>(p);
        }
    };
    auto store_patch = [&](int buf) {
        *reinterpret_cast<f32x4*>(&As[buf * A_TILE + t * 4]) = areg[0];
        *reinterpret_cast<f32x4*>(&As[buf * A_TILE + (t + 256) * 4]) = areg[1];
        if (t + 512 < 576) *reinterpret_cast<f32x4*>(&As[buf * A_TILE + (t + 512) * 4]) = areg[2];
        *reinterpret_cast<f32x4*>(&Ds[buf * D_TILE + t * 4]) = dreg;
    };

    const int wave = t >> 6, lane = t & 63;
    const int wk = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi32 = lane >> 5;

    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    if (p_begin < p_end) {
        load_patch(p_begin);
        store_patch(0);
    }
    __syncthreads();
    for (int pi = p_begin; pi < p_end; ++pi) {
        const int buf = (pi - p_begin) & 1;
        const bool more = pi + 1 < p_end;
        if (more) load_patch(pi + 1);
        const float* Ab = As + buf * A_TILE + hi32 * 64 + wk * 32 + l31;
        const float* Db = Ds + buf * D_TILE + hi32 * 64 + wn * 32 + l31;
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
            // k-step covers patch pixels m = 2*s2 + hi32  ->  (py, px) = (s2/2, 2*(s2%2) + hi32)
            const float bv = Db[s2 * 2 * 64];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dh = tap / 3, dw = tap - dh * 3;
                const float av = Ab[(((s2 >> 1) + dh) * 6 + 2 * (s2 & 1) + dw) * 64];
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
            }
        }
        if (more) store_patch(buf ^ 1);
        __syncthreads();
    }

    float* out = a.part + (size_t)sp * 9 * a.Cin * a.Cout;
    const int n = co0 + wn * 32 + l31;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = tap * a.Cin + ci0 + wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi32;
            out[(size_t)k * a.Cout + n] = acc[tap][r];
        }
}

// Same block tile (64ci x 64co x 9 taps, 4x4 output patches, split-K over patches) with the
// stage buffers TRANSPOSED -- As[ci][6 halo rows x 8] and Ds[co][16 pixels] -- so that the MFMA
// k dimension (pixels) is contiguous per lane: a lane fetches the five halo rows it needs as
// 15 ds_read_b64 and its two dY rows as 4, and the 72 MFMAs of a patch pick their A operand
// H[2g+dh][dw+j] out of registers.  19 LDS reads per 72 MFMAs instead of 80 ds_read_b32; the
// price is 4 ds_write_b32 per staged float4.  Channel strides 50 / 18 floats keep the b64 reads
// conflict free.  Global loads are buffer loads: patch displacement in the scalar offset,
// out-of-image pixels as out-of-range lane offsets (zeros).

// BF16 = true: both operands are rounded to bfloat16 on their way out of the registers and the 16
// pixels of a patch are ONE v_mfma_f32_32x32x16_bf16 per tap (k-block = patch rows 2*half, 2*half+1),
// fp32 accumulate -- the weight-gradient kernel of the mixed-precision mode (conv_bf16.hip).

// INBF = true (with BF16): x and dy are bfloat16 tensors in HBM (mixed-precision storage); the staging
// loads move 8 B per channel quad and widen to fp32 (exact), everything behind LDS is unchanged.
__device__ __forceinline__ f32x4 widen_bf16x4(u32x2 v) {
    f32x4 r;
    r.x = __builtin_bit_cast(float, v.x << 16);
    r.y = __builtin_bit_cast(float, v.x & 0xFFFF0000u);
    r.z = __builtin_bit_cast(float, v.y << 16);
    r.w = __builtin_bit_cast(float, v.y & 0xFFFF0000u);
    return r;
}

template <bool BF16, bool INBF = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad9t_kernel(Wgrad9Args a) {
    static_assert(BF16 || !INBF, "bf16 tensors only feed the bf16 MFMA path");
    constexpr int ES = INBF ? 2 : 4;                       // bytes per stored element
    constexpr int CS = 50, DS = 18;
    constexpr int A_TILE = 64 * CS, D_TILE = 64 * DS;
    // patches per pipeline step.  With bf16 inputs a patch is only 9 MFMAs, too short to cover the latency
    // of the next patch's loads: two patches per step put twice the loads in flight for the same
    // staging registers (the packed bf16 quads are half the size) -- the weight gradient is latency
    // bound in that mode, not MFMA bound.
    constexpr int PP = INBF ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];     // 2 * PP * (A_TILE + D_TILE) floats
    float* As = smem;
    float* Ds = smem + 2 * PP * A_TILE;
    const int t = threadIdx.x;
    const int logical = xcd_remap(blockIdx.x, a.tiles * a.splits);
    const int sp = logical / a.tiles, tile = logical - sp * a.tiles;
    const int cit = tile / a.co_tiles, cot = tile - cit * a.co_tiles;
    const int ci0 = cit * 64, co0 = cot * 64;
    const int p_begin = sp * a.per_split;
    const int p_end = min(a.npatch, p_begin + a.per_split);

    // staging slots: halo float4 f = t + 256 i -> (halo pixel f >> 4, channel quad f & 15)
    const int margin = (a.W + 1) * a.Cin * ES;           // most negative halo displacement, bytes
    int hy1[3], hx1[3], hvo[3], hls[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int f = t + 256 * i;
        const int hp = f >> 4, q = f & 15;
        const int hy = hp / 6, hx = hp - hy * 6;
        hy1[i] = f < 576 ? hy - 1 : 0x40000000;          // invalid slot: never inside the image
        hx1[i] = hx - 1;
        hvo[i] = ((hy - 1) * a.W + (hx - 1)) * a.Cin * ES + (ci0 + q * 4) * ES + margin;
        hls[i] = (4 * q) * CS + hy * 8 + hx;
    }
    const int dpix = t >> 4, dq = t & 15;
    const int dpy = dpix >> 2, dpx = dpix & 3;
    const int dvo = (dpy * a.W + dpx) * a.Cout * ES + (co0 + dq * 4) * ES;
    const int dls = (4 * dq) * DS + dpix;
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.x - margin), 0, (int)((size_t)a.N * a.H * a.W * a.Cin * ES + margin), 0x00020000);
    const __amdgpu_buffer_rsrc_t dsrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((size_t)a.N * a.H * a.W * a.Cout * ES), 0x00020000);

    f32x4 areg[3], dreg;                 // fp32 inputs
    u32x2 araw[PP][3], draw[PP];         // bf16 inputs: packed quads, widened when they are written to LDS
    auto load_patch = [&](int pidx0) {
#pragma unroll
      for (int sub = 0; sub < PP; ++sub) {
        const int pidx = pidx0 + sub;
        const bool live = pidx < p_end;                   // a step past the split's end stages zeros
        const int per_img = a.ph * a.pw;
        const int n = pidx / per_img, rem = pidx - n * per_img;
        const int py = rem / a.pw, px = rem - py * a.pw;
        const int h0 = py * 4, w0 = px * 4;
        const int pix0 = (n * a.H + h0) * a.W + w0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bool ok = live && (unsigned)(h0 + hy1[i]) < (unsigned)a.H && (unsigned)(w0 + hx1[i]) < (unsigned)a.W;
            if constexpr (INBF)
                araw[sub][i] = __builtin_amdgcn_raw_buffer_load_b64(xsrd, ok ? hvo[i] : (int)0x80000000, pix0 * a.Cin * ES, 0);
            else
                areg[i] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrd, ok ? hvo[i] : (int)0x80000000, pix0 * a.Cin * ES, 0));
        }
        const bool okd = live && h0 + dpy < a.H && w0 + dpx < a.W;
        if constexpr (INBF)
            draw[sub] = __builtin_amdgcn_raw_buffer_load_b64(dsrd, okd ? dvo : (int)0x80000000, pix0 * a.Cout * ES, 0);
        else
            dreg = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(dsrd, okd ? dvo : (int)0x80000000, pix0 * a.Cout * ES, 0));
      }
    };
    auto store_patch = [&](int buf) {
#pragma unroll
      for (int sub = 0; sub < PP; ++sub) {
        float* A = As + (buf * PP + sub) * A_TILE;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i < 2 || t < 64) {
                const f32x4 v = INBF ? widen_bf16x4(araw[sub][i]) : areg[i];
                A[hls[i]] = v.x;
                A[hls[i] + CS] = v.y;
                A[hls[i] + 2 * CS] = v.z;
                A[hls[i] + 3 * CS] = v.w;
            }
        }
        float* D = Ds + (buf * PP + sub) * D_TILE;
        const f32x4 d = INBF ? widen_bf16x4(draw[sub]) : dreg;
        D[dls] = d.x;
        D[dls + DS] = d.y;
        D[dls + 2 * DS] = d.z;
        D[dls + 3 * DS] = d.w;
      }
    };

    const int wave = t >> 6, lane = t & 63;
    const int wk = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi32 = lane >> 5;
    const int a_lane = (wk * 32 + l31) * CS + (BF16 ? 2 : 1) * hi32 * 8;
    const int d_lane = (wn * 32 + l31) * DS + (BF16 ? 2 : 1) * hi32 * 4;

    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    if (p_begin < p_end) {
        load_patch(p_begin);
        store_patch(0);
    }
    __syncthreads();
    for (int pi = p_begin; pi < p_end; pi += PP) {
        const int buf = ((pi - p_begin) / PP) & 1;
        const bool more = pi + PP < p_end;
        if (more) load_patch(pi + PP);
        if constexpr (BF16) {
#pragma unroll
          for (int sub = 0; sub < PP; ++sub) {
            const float* Ab = As + (buf * PP + sub) * A_TILE + a_lane;
            const float* Db = Ds + (buf * PP + sub) * D_TILE + d_lane;
            float Hb[4][6], Db2[2][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x2 v = *reinterpret_cast<const f32x2*>(Ab + r * 8 + c * 2);
                    Hb[r][2 * c] = v.x;
                    Hb[r][2 * c + 1] = v.y;
                }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f32x2 v = *reinterpret_cast<const f32x2*>(Db + r * 4 + c * 2);
                    Db2[r][2 * c] = v.x;
                    Db2[r][2 * c + 1] = v.y;
                }
            bf16x8 bv;
#pragma unroll
            for (int k = 0; k < 8; ++k) bv[k] = (__bf16)Db2[k >> 2][k & 3];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dh = tap / 3, dw = tap - dh * 3;
                bf16x8 av;
#pragma unroll
                for (int k = 0; k < 8; ++k) av[k] = (__bf16)Hb[(k >> 2) + dh][(k & 3) + dw];
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[tap], 0, 0, 0);
            }
          }
            if (more) store_patch(buf ^ 1);
            __syncthreads();
            continue;
        }
        const float* Ab = As + buf * A_TILE + a_lane;
        const float* Db = Ds + buf * D_TILE + d_lane;
        float H[5][6], Dv[2][4];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(Ab + r * 8 + c * 2);
                H[r][2 * c] = v.x;
                H[r][2 * c + 1] = v.y;
            }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(Db + g * 8 + c * 2);
                Dv[g][2 * c] = v.x;
                Dv[g][2 * c + 1] = v.y;
            }
        // k-step (g, j): patch pixels (2g + hi32, j)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dh = tap / 3, dw = tap - dh * 3;
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(H[2 * g + dh][dw + j], Dv[g][j], acc[tap], 0, 0, 0);
                }
        if (more) store_patch(buf ^ 1);
        __syncthreads();
    }

    float* out = a.part + (size_t)sp * 9 * a.Cin * a.Cout;
    const int n = co0 + wn * 32 + l31;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = tap * a.Cin + ci0 + wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi32;
            out[(size_t)k * a.Cout + n] = acc[tap][r];
        }
}

static bool wgrad9_ok(const ConvGeom& g) {
    return g.KH == 3 && g.KW == 3 && g.padT == 1 && g.padL == 1 && g.Ho == g.H && g.Wo == g.W && g.Cin % 64 == 0 &&
           g.Cout % 64 == 0;
}

struct Wgrad9Plan {
    int tiles, co_tiles, ph, pw, npatch, per_split, splits;
};
static Wgrad9Plan wgrad9_plan(const ConvGeom& g) {
    Wgrad9Plan p;
    p.co_tiles = g.Cout / 64;
    p.tiles = (g.Cin / 64) * p.co_tiles;
    p.ph = (g.H + 3) / 4;
    p.pw = (g.W + 3) / 4;
    p.npatch = g.N * p.ph * p.pw;
    static int target = l3_knob("L3_WG9_BLOCKS") ? atoi(l3_knob("L3_WG9_BLOCKS")) : 512;
    int splits = (target + p.tiles - 1) / p.tiles;
    const int max_splits = (p.npatch + 15) / 16;         // >= 16 patches per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.per_split = (p.npatch + splits - 1) / splits;
    p.splits = (p.npatch + p.per_split - 1) / p.per_split;
    return p;
}

struct WgradPlan {
    int TK, TN, ktiles, ntiles, splits, m_per_split;
    bool smallc;
};

static WgradPlan wgrad_plan(const ConvGeom& g) {
    WgradPlan p;
    const int K = g.KH * g.KW * g.Cin, M = g.N * g.Ho * g.Wo;
    p.smallc = (g.Cin % 64) != 0;
    p.TK = (!p.smallc && g.Cin % 128 == 0) ? 128 : 64;
    p.TN = g.Cout > 64 ? 128 : 64;
    p.ktiles = (K + p.TK - 1) / p.TK;
    p.ntiles = (g.Cout + p.TN - 1) / p.TN;
    const int tiles = p.ktiles * p.ntiles;
    int splits = (1024 + tiles - 1) / tiles;
    const int max_splits = (M + 255) / 256 > 0 ? (M + 255) / 256 : 1;   // >= 256 rows per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int mps = (M + splits - 1) / splits;
    mps = (mps + WG_MC - 1) / WG_MC * WG_MC;
    p.m_per_split = mps;
    p.splits = (M + mps - 1) / mps;
    return p;
}

// sample ranges whose tensors stay below the 2 GiB that wgrad9t's 32-bit buffer offsets reach
static int wgrad9_chunk_samples(const ConvGeom& g) {
    const size_t per_sample = (size_t)g.H * g.W * (g.Cin > g.Cout ? g.Cin : g.Cout) * 4;
    size_t nc = ((1ull << 31) - 1 - (size_t)(g.W + 1) * g.Cin * 4) / per_sample;
    if (nc > (size_t)g.N) nc = (size_t)g.N;
    return nc < 1 ? 1 : (int)nc;
}
static int wgrad9_total_splits(const ConvGeom& g) {
    const int nc = wgrad9_chunk_samples(g);
    int total = 0;
    for (int n0 = 0; n0 < g.N; n0 += nc) {
        ConvGeom gc = g;
        gc.N = g.N - n0 < nc ? g.N - n0 : nc;
        total += wgrad9_plan(gc).splits;
    }
    return total;
}

// Winograd path: the direct kernel's sample chunks, conv_wgrad_wino_splits() slices of 16 positions each, plus one
// slice for the sum over the splits
static int wgw_total_splits(const ConvGeom& g) {
    const int nc = wgrad9_chunk_samples(g);
    int total = 0;
    for (int n0 = 0; n0 < g.N; n0 += nc) total += conv_wgrad_wino_max_splits(g, g.N - n0 < nc ? g.N - n0 : nc);
    return total;
}

size_t conv_wgrad_scratch_floats(const ConvGeom& g) {
    if (wgrad9_ok(g)) {
        const size_t direct = (size_t)wgrad9_total_splits(g) * 9 * g.Cin * g.Cout;
        const size_t wino = ((size_t)wgw_total_splits(g) + 1) * 16 * g.Cin * g.Cout;      // either path may run (env switches)
        return direct > wino ? direct : wino;
    }
    const WgradPlan p = wgrad_plan(g);
    const size_t generic = (size_t)p.splits * g.KH * g.KW * g.Cin * g.Cout, first = conv_first_wgrad_scratch_floats(g);
    return generic > first ? generic : first;
}

bool conv_wgrad_bf16_ok(const ConvGeom& g) {
    return wgrad9_ok(g) && (size_t)g.H * g.W * (g.Cin > g.Cout ? g.Cin : g.Cout) * 4 + (size_t)(g.W + 1) * g.Cin * 4 < (1ull << 31);
}

double conv_wgrad_executed_flops(const ConvGeom& g, bool bf16) {
    if (!bf16 && wgrad9_ok(g) && conv_wgrad_bf16_ok(g) && conv_wgrad_wino_ok(g)) return conv_wgrad_wino_executed_flops(g);
    return -1.0;
}

void conv_wgrad(const float* x, const float* dy, float* dw, float* part, const ConvGeom& g,
                hipStream_t s, bool bf16, bool in_bf16, const FirstWgFuse* first_fuse) {
    if (wgrad9_ok(g)) {
        static const int use_t = l3_knob("L3_WG9T") ? atoi(l3_knob("L3_WG9T")) : 1;
        const bool fits = conv_wgrad_bf16_ok(g);          // one sample fits the 32-bit offsets
        if (!bf16 && fits && conv_wgrad_wino_ok(g)) {
            // fp32: Winograd F(3x3, 2x2) in the transformed domain (conv_wgrad_wino.hip)
            const int nc = wgrad9_chunk_samples(g);
            const size_t slice = (size_t)16 * g.Cin * g.Cout;
            int total = 0;
            for (int n0 = 0; n0 < g.N; n0 += nc) {
                const int n = g.N - n0 < nc ? g.N - n0 : nc;
                conv_wgrad_wino_launch(x + (size_t)n0 * g.H * g.W * g.Cin, dy + (size_t)n0 * g.H * g.W * g.Cout,
                                       part + (size_t)total * slice, g, n, s);
                total += conv_wgrad_wino_splits(g, n);
            }
            float* sum = part + (size_t)total * slice;
            if (total > 1) wgrad_reduce(part, sum, (int64_t)slice, total, s);
            conv_wgrad_wino_finish(total > 1 ? sum : part, dw, g, 1, s);
            return;
        }
        const int nc = fits ? wgrad9_chunk_samples(g) : g.N;
        const size_t slice = (size_t)9 * g.Cin * g.Cout;
        int total_splits = 0;
        for (int n0 = 0; n0 < g.N; n0 += nc) {
            ConvGeom gc = g;
            gc.N = g.N - n0 < nc ? g.N - n0 : nc;
            const Wgrad9Plan p = wgrad9_plan(gc);
            Wgrad9Args a;
            const size_t es = in_bf16 ? 2 : 4;
            a.x = reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) + (size_t)n0 * g.H * g.W * g.Cin * es);
            a.dy = reinterpret_cast<const float*>(reinterpret_cast<const char*>(dy) + (size_t)n0 * g.H * g.W * g.Cout * es);
            a.part = part + (size_t)total_splits * slice;
            a.N = gc.N; a.H = g.H; a.W = g.W; a.Cin = g.Cin; a.Cout = g.Cout;
            a.co_tiles = p.co_tiles; a.tiles = p.tiles; a.ph = p.ph; a.pw = p.pw;
            a.npatch = p.npatch; a.per_split = p.per_split; a.splits = p.splits;
            constexpr size_t WG9T_LDS = 2 * (64 * 50 + 64 * 18) * sizeof(float);      // one patch per step
            if (bf16 && fits && in_bf16 && conv_wgrad_bf16_tr_enabled()) {
                // transpose-read kernel (conv_wgrad_bf16.hip): same split-K slices, its own 4 x 16 patches
                conv_wgrad_bf16_tr_launch(a.x, a.dy, a.part, g, gc.N, p.splits, s);
            } else if (bf16 && fits && in_bf16) {
                static std::once_flag once[L3_MAX_DEVICES];
                int dev = 0;
                (void)hipGetDevice(&dev);
                std::call_once(once[dev & (L3_MAX_DEVICES - 1)], [] {
                    (void)hipFuncSetAttribute((const void*)conv_wgrad9t_kernel<true, true>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * WG9T_LDS));
                });
                hipLaunchKernelGGL((conv_wgrad9t_kernel<true, true>), dim3(p.tiles * p.splits), dim3(256), 2 * WG9T_LDS, s, a);
            } else if (bf16 && fits) {
                hipLaunchKernelGGL(conv_wgrad9t_kernel<true>, dim3(p.tiles * p.splits), dim3(256), WG9T_LDS, s, a);
            } else if (use_t && fits) {
                hipLaunchKernelGGL(conv_wgrad9t_kernel<false>, dim3(p.tiles * p.splits), dim3(256), WG9T_LDS, s, a);
            } else {
                hipLaunchKernelGGL(conv_wgrad9_kernel, dim3(p.tiles * p.splits), dim3(256), 0, s, a);
            }
            total_splits += p.splits;
        }
        wgrad_reduce(part, dw, (int64_t)slice, total_splits, s);
        return;
    }
    if (!bf16 && conv_first_wgrad_ok(g)) {       // first layer of a tower: dY-streaming MFMA kernel (conv_first.hip)
        const int parts = conv_first_wgrad(x, dy, part, g, s, first_fuse);
        wgrad_reduce(part, dw, (int64_t)g.KH * g.KW * g.Cin * g.Cout, parts, s);
        return;
    }
    const WgradPlan p = wgrad_plan(g);
    WgradArgs a;
    a.x = x; a.dy = dy; a.part = part;
    a.N = g.N; a.H = g.H; a.W = g.W; a.Cin = g.Cin; a.Ho = g.Ho; a.Wo = g.Wo; a.Cout = g.Cout;
    a.KH = g.KH; a.KW = g.KW; a.padT = g.padT; a.padL = g.padL;
    a.M = g.N * g.Ho * g.Wo;
    a.K = g.KH * g.KW * g.Cin;
    a.ktiles = p.ktiles; a.ntiles = p.ntiles; a.splits = p.splits; a.m_per_split = p.m_per_split;
    a.nvec = (g.Cout % 4) == 0;
    dim3 grid(p.ktiles * p.ntiles, p.splits), block(256);
#define L3_WG(TK_, TN_, SC_)                                                                              \
    do {                                                                                                  \
        if (a.nvec)                                                                                       \
            hipLaunchKernelGGL((conv_wgrad_kernel<TK_, TN_, SC_, true>), grid, block, 0, s, a);           \
        else                                                                                              \
            hipLaunchKernelGGL((conv_wgrad_kernel<TK_, TN_, SC_, false>), grid, block, 0, s, a);          \
    } while (0)
    if (p.smallc) {
        if (p.TN == 128) L3_WG(64, 128, true); else L3_WG(64, 64, true);
    } else if (p.TK == 128) {
        if (p.TN == 128) L3_WG(128, 128, false); else L3_WG(128, 64, false);
    } else {
        if (p.TN == 128) L3_WG(64, 128, false); else L3_WG(64, 64, false);
    }
#undef L3_WG
    wgrad_reduce(part, dw, (int64_t)a.K * a.Cout, p.splits, s);
}

}  // namespace l3
