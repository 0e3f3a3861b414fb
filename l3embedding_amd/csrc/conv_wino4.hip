// conv_wino4.hip -- 3x3 / stride 1 / pad 1 convolution as Winograd F(4x4, 3x3) on fp32 MFMA.
//
// Same role as conv_wino.hip (forward and data gradient of the `Conv2D(n, (3, 3), padding='same')` layers of
// l3embedding/audio_model.py:372-445 and vision_model.py:126-205) for the layers with many input channels:
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        per 4x4 output tile (6x6 input patch), summed over channels
//
// 36 transformed positions per 16 output pixels: 4x fewer multiplies than the direct convolution (F(2x2,3x3): 2.25x),
// at several times the rounding error of F(2x2,3x3) (tests/test_layer_parity_gpu.py holds the measured budget).
// Interpolation points {0, 1, -1, 2, -1/2, inf} (see BT4 below).
//
// Mapping (gfx950):
//   block  = 32 consecutive tiles of the flattened (sample, tile row, tile column) order x 64 output channels, 12 waves;
//   wave w = three positions: xi = w / 2 (row of B^T), nu = 3 (w % 2) + {0, 1, 2} -- 3 positions x (32 tiles x 64 channels)
//            = 6 MFMA tiles of 32x32, 96 accumulator registers, three waves per SIMD;
//   k loop = 8 input channels per stage, one workgroup barrier, software-pipelined.  The raw 6x6 patches go HBM -> LDS as
//            k-half planes ([i][j][k-half][tile] x 16 B = the four channels of that half: one buffer_load ... lds of 16 B per
//            lane), double buffered; a lane (tile, k-half) reads the <= 5 x 5 raw pixels its positions need with conflict-free
//            ds_read_b128 (all four channels of its half at once; every (i, j) is an immediate offset), forms
//            t[j] = sum_i B^T[xi][i] d[i][j] and V[nu] += B^T[nu][j] t[j] with compile-time coefficients (12 copies of the
//            stage loop, one per wave) and feeds V straight into the A operand of v_mfma_f32_32x32x2_f32 (V never touches
//            LDS): k-steps 0, 1 against filter sub-slot 0, k-steps 2, 3 against sub-slot 1.  The transform of stage c + 1
//            runs in the gaps between the wave's own MFMAs of stage c (see stage_loop).  The filter slices live in two
//            4-channel sub-slots that only the loading wave reads: once a position's two operands are in registers it
//            requests that position's slice of the next stage into the same place, no barrier involved.
//            (Round-3 history, profiles/r03_wino4_loop_variants.txt and r03_wino4_phase_timing.txt: 4-channel stages read 8
//            bytes of a 16-byte LDS-DMA slot -- a 2-way bank conflict on every raw read -- and no restructuring of that loop
//            helped; 8-channel stages in lockstep (transform, MFMAs, barrier) left the matrix pipe idle while the youngest
//            wave of a SIMD finished its transform; the pipelined loop is worth another 10 %.)
//   output = the 36 positions of a (tile, channel) meet through LDS in FOUR rounds (channel half x tile half, 72 KiB each, in
//            patch slot 1 + filter sub-slot 1); waves 0..7 apply A^T M A, add the bias, accumulate the BatchNorm partials,
//            transpose 4 x 4 (pixel row x channel) across four lanes and store 16 bytes per lane; waves 8..11 meanwhile bring
//            the NEXT tile block's first stage into patch slot 0 / filter sub-slot 0 (patches requested during the last stage,
//            filter slices behind the rounds' barriers) -- the persistent loop enters its next tile block with stage 0 in LDS;
//   tail   = a launch that runs alone splits its last, partial round of tile blocks over input-channel slices
//            (conv_wino4_kernel<0, true> + wino4_tail_reduce_kernel, see launch_wino4);
//   budget = on gfx950 an fp32 VALU instruction takes ~4.3 cycles of the SIMD's matrix time (the fp32 MFMA runs on the vector
//            ALUs: scripts/mfma_mix.hip), so a stage costs 72 MFMA x 64 + 3 waves x ~110 VALU x 4.3 = ~6030 cycles -- what it
//            takes (profiles/r04_wino4_phase_timing.txt): every VALU instruction in the stage loop is matrix time;
//   U layout in HBM: [pos 36][Cin/8][sub 2][k-half 2][Cout][2], channel = 8 q + 4 half + 2 sub + c (conv_wino4 weight
//            transform): a lane's ds_read_b64 of the B operand is conflict free ([half][64 couts] x 8 B) and a 1-KiB LDS-DMA
//            piece = one (position, half-stage).
#include "kernels.h"
#include "device_common.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

namespace l3 {

namespace {

struct Wino4Args {
    const float* x;
    const float* u;
    const float* bias;
    float* y;
    int N, H, W, Cin, Cout;
    int TY, TX;        // 4x4 tiles per image
    int tiles;         // N * TY * TX
    int mblocks, nblocks, nchunks;
    float inv_tpi, inv_tx;      // 1 / (TY * TX), 1 / TX: divisions as one multiply (tiles < 2^22)
    float* stat_part;  // SM != 0: per-(tile block) BatchNorm partial sums [mblock][2][Cout] (bn_fused.hip layout)
    int stat_mode;     // SM == 1: 1 = moments of y, 2 = moments of relu(y)
    BnBwdFuse bb;      // SM == 2: the launch is a data gradient (kernels.h)
    int stagger_cycles, stagger_groups;      // persistent grid: CU group g starts g * stagger_cycles late (see the kernel)
    // The launch covers the physical tile blocks [lt_first, lt_end) of the mblocks * nblocks of the layer.  ksplit > 1: the
    // TAIL launch of a layer whose block count is not a multiple of the CU count -- workgroup b takes tile block lt_first + b /
    // ksplit and the b % ksplit-th slice of its input channels, and leaves A^T M A of that slice (no bias, no statistics) in
    // ypart[b][tile 32][pixel 16][channel 64] for wino4_tail_reduce_kernel (see conv_wino4_launch).
    int lt_first, lt_end, ksplit;
    int* work;         // persistent grid: its work counters (device_common.h wq_*), nullptr = tile blocks by the static stride
    float* ypart;
    int solo;          // ConvGeom::solo: the tail may be split
    int dynamic;       // ConvGeom::dynamic: tile blocks through work counters
    float* tail_buf;   // ConvGeom::tail_scratch / tail_scratch_bytes
    size_t tail_cap;
};

// F(4x4, 3x3) on the interpolation points 0, 1, -1, 2, -1/2, inf.  The textbook set (0, +-1, +-2, inf) has a sparser
// B^T (rows of 3-4 entries against 4-5) but 2.7x the worst-case rounding error: its transforms multiply by up to 5 and 8
// on BOTH signs of 2, the mixed pair (2, -1/2) keeps one of every product pair small (float32 restatement with the
// MFMA's sequential fma chain over 512 channels: max error 4.2e-6 of the output range against 1.1e-5; scripts/
// wino_error_model.py, profiles/r03_wino4_error_model.txt).
//   B^T row j < 5 = coefficients of prod_{l != j} (x - a_l), row 5 = prod_l (x - a_l);  G row j = a_j^k / prod_{l != j}(a_j - a_l);
//   A^T[i][j] = a_j^i   (Toom-Cook; the point at infinity closes each matrix)
__device__ constexpr float BT4[6][6] = {{1.f, 1.5f, -2.f, -1.5f, 1.f, 0.f}, {0.f, -1.f, -2.5f, -0.5f, 1.f, 0.f},
                                        {0.f, 1.f, 0.5f, -2.5f, 1.f, 0.f},  {0.f, -0.5f, -1.f, 0.5f, 1.f, 0.f},
                                        {0.f, 2.f, -1.f, -2.f, 1.f, 0.f},   {0.f, 1.f, 1.5f, -2.f, -1.5f, 1.f}};
__device__ constexpr float AT4[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f},
                                        {0.f, 1.f, -1.f, 2.f, -0.5f, 0.f},
                                        {0.f, 1.f, 1.f, 4.f, 0.25f, 0.f},
                                        {0.f, 1.f, -1.f, 8.f, -0.125f, 1.f}};

// out[r] = sum_k A^T[r][k] in[k], coefficients folded at compile time
__device__ __forceinline__ void at4_apply(const float (&in)[6], float (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float sacc = 0.f;
        bool started = false;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float c = AT4[r][k];
            if (c == 0.f) continue;
            if (!started) {
                sacc = c == 1.f ? in[k] : c * in[k];
                started = true;
            } else if (c == 1.f) {
                sacc += in[k];
            } else if (c == -1.f) {
                sacc -= in[k];
            } else {
                sacc = fmaf(c, in[k], sacc);
            }
        }
        out[r] = sacc;
    }
}

constexpr int W4_TILES = 32, W4_WAVES = 12, W4_THREADS = W4_WAVES * 64;
// A stage = 8 input channels.  Patches: [ij 36][k-half 2][tile 32][4 channels] (k-half h = channels 4h .. 4h + 3 of the stage),
// two slots.  Filters: two 4-channel sub-slots [pos 36][k-half 2][cout 64][2 channels] -- sub-slot s holds the channels
// 4h + 2s, 4h + 2s + 1 of the stage, i.e. the MFMA k-steps 2s and 2s + 1.
constexpr int W4_A_FLOATS = 36 * 2 * W4_TILES * 4;           // 9216 floats = 36 KiB
constexpr int W4_U_FLOATS = 36 * 2 * 64 * 2;                 // 9216 floats = 36 KiB
constexpr int W4_U_BASE = 2 * W4_A_FLOATS;
constexpr int W4_RING_FLOATS = 2 * W4_A_FLOATS + 2 * W4_U_FLOATS;      // 144 KiB
// (the output transform's exchange area lives inside the ring: patch slot 1 for positions 0..17, filter sub-slot 1 for 18..35,
//  [pos][tile pair 8][64 dwords] = 72 KiB per round)
constexpr int W4_PF_BASE = W4_RING_FLOATS;                   // [4 waves][9 pieces][64 lanes]: buffer offsets of the next tile block's first patches (waves 8..11)
constexpr int W4_PF_FLOATS = 4 * 9 * 64;
constexpr int W4_BIAS_BASE = W4_PF_BASE + W4_PF_FLOATS;       // the tile block's 64 bias values
constexpr int W4_BN_BASE = W4_BIAS_BASE + 64;                 // SM == 2: [64 channels] x {scale, shift, mean, 1 / sqrt(var + eps)}
constexpr int W4_Q_BASE = W4_BN_BASE + 256;                   // two ints: the tile blocks this workgroup has claimed (dynamic assignment)
constexpr size_t W4_LDS_BYTES = (size_t)(W4_RING_FLOATS + W4_PF_FLOATS + 64 + 256 + 4) * sizeof(float);

// One ds_read_b64, never half of a ds_read2_b64 / ds_read2st64_b64: the paired forms move 16 B per lane in 16 LDS cycles
// (a ds_read_b64 moves 8 B in 2; MI355X_MICROARCH.md, LDS) and the load/store optimizer pairs every two reads off one base
// register inside a merge region.  A side-effecting (empty) asm statement ends the region.
__device__ __forceinline__ f32x2 lds_read_b64(const float* p) {
    const f32x2 v = *reinterpret_cast<const f32x2*>(p);
    asm volatile("");
    return v;
}
__device__ __forceinline__ f32x4 lds_read_b128(const float* p) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    asm volatile("");
    return v;
}

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier).  __syncthreads() also waits for
// vmcnt(0) -- in the output transform that is the write acknowledgement of every global store issued so far, a full memory
// round trip per exchange round that nothing depends on.  The empty asm statements keep the compiler from moving LDS
// accesses across it.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xF | 0x70 | (0 << 8) | (0x3 << 14));      // vmcnt 63 (no wait), expcnt 7, lgkmcnt 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

struct TrueT { static constexpr bool value = true; };
struct FalseT { static constexpr bool value = false; };
template <int V> struct IntT { static constexpr int value = V; };

// PART: the channel-sliced tail launch (Wino4Args::ksplit > 1; SM == 0 only)
template <int SM, bool PART = false>
__global__ __launch_bounds__(W4_THREADS) void conv_wino4_kernel(Wino4Args a) {
    constexpr bool STATS = SM != 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane0 = t & 63;
    const int total_blocks = a.mblocks * a.nblocks;
    // have0: the patches of stage 0 and the first filter sub-slot of this tile block are already in LDS -- requested by
    // waves 8..11 of the PREVIOUS tile block while waves 0..7 ran its output transform (see the epilogue)
    bool have0 = false;
    if (a.stagger_cycles > 0) {
        const long long wait = (long long)(((int)blockIdx.x >> 3) % a.stagger_groups) * a.stagger_cycles;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while ((long long)(__builtin_amdgcn_s_memtime() - t0) < wait) __builtin_amdgcn_s_sleep(16);
    }
    const int lt_begin = a.lt_first + (PART ? (int)blockIdx.x / a.ksplit : (int)blockIdx.x);
    const int lt_step = PART ? a.lt_end : (int)gridDim.x;                  // a split workgroup takes one slice of one tile block
    // Dynamic assignment (device_common.h wq_*): the workgroup always holds TWO claims -- the tile block it works on and the one
    // after it, whose first stage it requests ahead -- in the LDS words Q[0..1] (Q[it & 1] = block of iteration it).  Wave 11
    // (no part in the output transform) claims the block after next at the start of the transform and parks it in the word
    // the current block came from; the barrier at the top of the next block publishes it.
    const bool dyn = !PART && a.work != nullptr;
    const int xcd = (int)blockIdx.x & 7;
    int* const Qw = reinterpret_cast<int*>(smem + W4_Q_BASE);
    if (dyn) {
        if (t == 0) {
            int c0, c1;
            wq_claim2(a.work, xcd, a.lt_first, a.lt_end, c0, c1);
            Qw[0] = c0;
            Qw[1] = c1;
        }
        __syncthreads();
    }
    // this workgroup's 8-channel stages [c8_0, c8_0 + n8)
    const int ks = PART ? (int)blockIdx.x % a.ksplit : 0;
    const int c8_0 = PART ? a.nchunks * ks / a.ksplit : 0;
    const int n8_own = PART ? a.nchunks * (ks + 1) / a.ksplit - c8_0 : a.nchunks;
    for (int it = 0;; ++it) {
    if (it != 0) lds_barrier();                          // the previous tile block's last LDS reads are done (and Q is published)
    int lt, lt2;                                         // this tile block and the next one of this workgroup (-1: none)
    if (dyn) {
        lt = __builtin_amdgcn_readfirstlane(Qw[it & 1]);
        lt2 = __builtin_amdgcn_readfirstlane(Qw[(it + 1) & 1]);
    } else {
        lt = lt_begin + it * lt_step;
        lt2 = !PART && lt + lt_step < a.lt_end ? lt + lt_step : -1;
        if (PART && it != 0) lt = -1;
    }
    if (lt < 0 || lt >= a.lt_end) break;
    int lane = lane0;
    asm volatile("" : "+v"(lane));                       // keep lane-derived addresses inside the loop (see conv_wino.hip)
    const int logical = xcd_remap(lt, total_blocks);
    const int nb = logical % a.nblocks, mb = logical / a.nblocks;
    const int n0 = nb * 64, T0 = mb * W4_TILES;

    // ---- staging descriptors ------------------------------------------------------------------------------------
    // patches: 36 one-KiB pieces per stage, piece ij = 32 tiles x 2 k-halves x 16 B; wave w sends pieces w, w + 12, w + 24
    unsigned avoff[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int ij = wave + 12 * q, tile = lane & 31;
        unsigned vo = 0x80000000u;
        const int T = T0 + tile;
        if (T < a.tiles) {
            const int n = (int)(((float)T + 0.5f) * a.inv_tpi), rem = T - n * a.TY * a.TX;
            const int ty = (int)(((float)rem + 0.5f) * a.inv_tx), tx = rem - ty * a.TX;
            const int i = ij / 6, j = ij - i * 6;
            const int yy = 4 * ty - 1 + i, xx = 4 * tx - 1 + j;
            if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W)
                vo = (unsigned)(((n * a.H + yy) * a.W + xx) * a.Cin * 4 + (lane >> 5) * 16);
        }
        avoff[q] = vo;
    }
    // waves 8..11 request the NEXT tile block's first patches during this block's last stage (stage_loop): the per-lane
    // buffer offsets of their nine pieces are worked out here, beside this block's, and parked in LDS --
    // computed at the point of use it costs the stage loops of every wave registers (scalar spills take vector registers)
    if (wave >= 8) {
        const int mb2 = xcd_remap(lt2 < 0 ? 0 : lt2, total_blocks) / a.nblocks;
        const int T2 = mb2 * W4_TILES + (lane & 31);
        const int n2 = (int)(((float)T2 + 0.5f) * a.inv_tpi), rem = T2 - n2 * a.TY * a.TX;
        const int ty2 = (int)(((float)rem + 0.5f) * a.inv_tx), tx2 = rem - ty2 * a.TX;
        const bool valid = lt2 >= 0 && T2 < a.tiles;
        unsigned* pfo = reinterpret_cast<unsigned*>(smem + W4_PF_BASE) + (wave - 8) * 9 * 64 + lane;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int ij = (wave - 8) * 9 + q;
            const int i = ij / 6, j = ij - i * 6;
            const int yy = 4 * ty2 - 1 + i, xx = 4 * tx2 - 1 + j;
            const bool in = valid && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            pfo[q * 64] = in ? (unsigned)(((n2 * a.H + yy) * a.W + xx) * a.Cin * 4 + (lane >> 5) * 16) : 0x80000000u;
        }
    }
    // filters: this wave's positions 3 w .. 3 w + 2; lane -> (k-half, cout pair)
    const unsigned uvoff = (unsigned)((lane >> 5) * a.Cout * 8 + (n0 + (lane & 31) * 2) * 8);
    const __amdgpu_buffer_rsrc_t xsrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)a.N * a.H * a.W * a.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t usrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, (int)((size_t)36 * a.Cin * a.Cout * 4), 0x00020000);
    const int c8_total = a.Cin >> 3;

    auto issue_a = [&](int slot, int c8) {               // three pieces of the patches of 8-channel stage c8
        float* As = smem + slot * W4_A_FLOATS;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (__attribute__((address_space(3))) void*)(As + (wave + 12 * q) * 256),
                                                     16, (int)avoff[q], (c8_0 + c8) * 32, 0, 0);
    };
    auto issue_u = [&](int sub, int c8) {                // this wave's three filter slices of half-stage (c8, sub)
        float* Us = smem + W4_U_BASE + sub * W4_U_FLOATS;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int pos = wave * 3 + p;
            const int usoff = ((pos * c8_total + c8_0 + c8) * 2 + sub) * a.Cout * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(usrd, (__attribute__((address_space(3))) void*)(Us + pos * 256), 16,
                                                     (int)uvoff, usoff, 0, 0);
        }
    };

    const int l31 = lane & 31, half = lane >> 5;
    // bias of this thread's two output channels of the output transform, requested now: asked for in the epilogue, the
    // load's round trip is exposed twice per tile block
    // (SM == 2 launches are data gradients: no bias, and no registers to spare)
    // the tile block's 64 bias values, parked in LDS for the output transform (held in registers they cost the stage loops two)
    if (wave == 11) smem[W4_BIAS_BASE + lane] = a.bias != nullptr && !PART ? a.bias[n0 + lane] : 0.f;
    if constexpr (SM == 2) {        // ... and the constants of the fused BatchNorm-backward reduction (four loads per lane and round otherwise)
        if (wave == 10)
            *reinterpret_cast<f32x4*>(smem + W4_BN_BASE + lane * 4) =
                f32x4{a.bb.scale[n0 + lane], a.bb.shift[n0 + lane], a.bb.mean[n0 + lane], rsqrtf(a.bb.var[n0 + lane] + a.bb.eps)};
    }
    const int lane_a = (half * 32 + l31) * 4;                          // floats: [ij][k-half][tile][4 channels]
    const int lane_b = W4_U_BASE + wave * 3 * 256 + half * 128 + l31 * 2;

    f32x16 acc[3][2];

    // Stage = 8 input channels, one workgroup barrier.  A lane (tile, k-half) reads its half's FOUR channels of a raw pixel
    // with one conflict-free ds_read_b128 (the round-3 first version staged 4 channels and read 8 bytes of a 16-byte slot:
    // a 2-way bank conflict on every read, twice the read instructions and twice the barriers), transforms them to
    // V[3 positions][4 channels] and multiplies in two halves: k-steps 0, 1 against filter sub-slot 0, k-steps 2, 3 against
    // sub-slot 1.  The filter sub-slots are private to the wave that loads and reads them: as soon as its six operands of a
    // half-stage are in registers it requests the same half of the next stage into the sub-slot.
    auto stage_loop = [&](auto XIT, auto NHT) {
        constexpr int XI = decltype(XIT)::value, NH = decltype(NHT)::value;
        // Column jj of the patches at SA: the raw pixels d[i][NH + jj] the row B^T[XI] touches ...
        auto col_read = [&](const float* SA, auto JJT, f32x4 (&d)[5]) {
            constexpr int jj = decltype(JJT)::value;
            int k = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (BT4[XI][i] == 0.f) continue;
                d[k++] = lds_read_b128(SA + (i * 6 + NH + jj) * 256);
            }
        };
        // ... combined to t = sum_i B^T[XI][i] d[i], then V[nu] += B^T[nu][NH + jj] t (the same fma chain, in the same order,
        // as forming all five t first)
        // Channel by channel, and this file is built with -fno-slp-vectorize (_build.py): plain v_fma_f32 / v_add_f32.  Packed
        // fp32 arithmetic (v_pk_fma_f32) in the gaps between MFMAs costs the matrix pipe more than twice as many plain ones
        // (MI355X_MICROARCH.md; measured here: 628 -> 616 us per forward launch).
        auto col_comb = [&](auto JJT, const f32x4 (&d)[5], f32x4 (&vn)[3]) {
            constexpr int jj = decltype(JJT)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float tc = 0.f;
                int k = 0;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const float c = BT4[XI][i];
                    if (c == 0.f) continue;
                    const float dv = d[k][q];
                    if (k == 0)
                        tc = c == 1.f ? dv : c * dv;
                    else if (c == 1.f)
                        tc = tc + dv;
                    else if (c == -1.f)
                        tc = tc - dv;
                    else
                        tc = fmaf(c, dv, tc);
                    ++k;
                }
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const int nu = NH * 3 + p;
                    const float c = BT4[nu][NH + jj];
                    if (c == 0.f) continue;
                    bool first = true;                     // no earlier column contributes to V[nu]
#pragma unroll
                    for (int j2 = 0; j2 < 5; ++j2)
                        if (j2 < jj && BT4[nu][NH + j2] != 0.f) first = false;
                    if (first)
                        vn[p][q] = c == 1.f ? tc : c * tc;
                    else if (c == 1.f)
                        vn[p][q] = vn[p][q] + tc;
                    else if (c == -1.f)
                        vn[p][q] = vn[p][q] - tc;
                    else
                        vn[p][q] = fmaf(c, tc, vn[p][q]);
                }
            }
        };
        // Position p of half-stage `sub` (k-steps 2 sub, 2 sub + 1): its two filter operands out of the wave's sub-slot
        auto b_read = [&](int sub, int pp, f32x2 (&b)[2]) {
            const float* SU = smem + sub * W4_U_FLOATS + lane_b;
            b[0] = lds_read_b64(SU + pp * 256);
            b[1] = lds_read_b64(SU + pp * 256 + 64);
        };
        auto u_piece = [&](int sub, int pp, int c8) {       // the 1-KiB filter slice (position, half-stage) of stage c8
            const int pos = wave * 3 + pp;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                usrd, (__attribute__((address_space(3))) void*)(smem + W4_U_BASE + sub * W4_U_FLOATS + pos * 256), 16, (int)uvoff,
                ((pos * c8_total + c8_0 + c8) * 2 + sub) * a.Cout * 16, 0, 0);
        };
        auto a_piece = [&](int slot, int c8, int q) {       // patch piece q of this wave
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                xsrd, (__attribute__((address_space(3))) void*)(smem + slot * W4_A_FLOATS + (wave + 12 * q) * 256), 16, (int)avoff[q],
                (c8_0 + c8) * 32, 0, 0);
        };
        auto mfma2 = [&](int sub, auto PT, int ks, const f32x4 (&v)[3], const f32x2 (&b)[2]) {
            constexpr int pp = decltype(PT)::value;
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
                acc[pp][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[pp][2 * sub + ks], b[jn][ks], acc[pp][jn], 0, 0, 0);
        };
        auto mfma4 = [&](int sub, auto PT, const f32x4 (&v)[3], const f32x2 (&b)[2]) {
            constexpr int pp = decltype(PT)::value;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
                    acc[pp][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[pp][2 * sub + ks], b[jn][ks], acc[pp][jn], 0, 0, 0);
        };
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][jn][r] = 0.f;
        const int n8 = n8_own;                             // 8-channel stages
        constexpr int NZ = (BT4[XI][0] != 0.f) + (BT4[XI][1] != 0.f) + (BT4[XI][2] != 0.f) + (BT4[XI][3] != 0.f) +
                           (BT4[XI][4] != 0.f) + (BT4[XI][5] != 0.f);      // raw pixels per patch column
        auto SB = [] { __builtin_amdgcn_sched_barrier(0); };
        // s_waitcnt immediates: vmcnt in bits 3:0 and 15:14, lgkmcnt in 11:8, expcnt 6:4 left at 7
        auto WAIT = [](auto VMT, auto LGKMT) {
            constexpr int vm = decltype(VMT)::value, lgkm = decltype(LGKMT)::value;
            __builtin_amdgcn_s_waitcnt((vm & 0xF) | 0x70 | (lgkm << 8) | ((vm >> 4) << 14));
        };
        // Software pipeline.  While the matrix pipe works through stage c (V(c) in registers) the wave reads and transforms
        // the patches of stage c + 1 in the gaps between its own MFMAs: a stage is six groups (half-stage, position) of four
        // MFMAs; a group's filter operands were read during the group before; a patch column's raw pixels (and the LDS-DMA
        // requests) go out between the group's two MFMA pairs -- in the shadow of the wave's own MFMAs -- and are combined behind
        // the second pair.  VMEM order per stage and wave: [patch piece q of stage c + 2, filter
        // slice (0, q) of stage c + 1] for q = 0..2, then the slices (1, 0..2) -- each slice as soon as the operands it
        // replaces are in registers.  The end of a stage waits for everything but the second half's slices (vmcnt 3), which
        // stay in flight across the barrier; a second-half operand read waits for its own slice (vmcnt 6 / 7 / 7).
        f32x4 v[3];
        f32x2 bc[2], bn[2];
        if (n8 > 1) issue_a(1, 1);
        {
            f32x4 d[5];
            col_read(smem + lane_a, IntT<0>{}, d); col_comb(IntT<0>{}, d, v);
            col_read(smem + lane_a, IntT<1>{}, d); col_comb(IntT<1>{}, d, v);
            col_read(smem + lane_a, IntT<2>{}, d); col_comb(IntT<2>{}, d, v);
            col_read(smem + lane_a, IntT<3>{}, d); col_comb(IntT<3>{}, d, v);
            col_read(smem + lane_a, IntT<4>{}, d); col_comb(IntT<4>{}, d, v);
        }
        b_read(0, 0, bc);
        __syncthreads();                                   // patches of stage 1 landed; slot 0 read by every wave
        SB();
        // One stage of the pipeline.  PEN: the second-to-last stage -- no patches are left to request (stage c + 2 does not
        // exist), so its VMEM sequence is the six filter slices only and the counted waits shrink accordingly.
        // (no branch inside a stage body: with more than one basic block the transform arithmetic is sunk to its use at the end
        // of the stage and every raw pixel spills -- hence two instantiations instead of a condition.)
        auto stage_body = [&](int c, auto PENT) __attribute__((always_inline)) {
            constexpr bool PEN = decltype(PENT)::value;
            const float* SA = smem + ((c + 1) & 1) * W4_A_FLOATS + lane_a;
            f32x4 vn[3];
            f32x4 d[5];
            // (0, 0) + column 0
            mfma2(0, IntT<0>{}, 0, v, bc); SB();
            col_read(SA, IntT<0>{}, d); b_read(0, 1, bn); SB();
            WAIT(IntT<63>{}, IntT<NZ + 2>{}); if constexpr (!PEN) a_piece(c & 1, c + 2, 0); u_piece(0, 0, c + 1); SB();
            mfma2(0, IntT<0>{}, 1, v, bc); SB();
            col_comb(IntT<0>{}, d, vn); SB();
            // (0, 1) + column 1
            mfma2(0, IntT<1>{}, 0, v, bn); SB();
            col_read(SA, IntT<1>{}, d); b_read(0, 2, bc); SB();
            WAIT(IntT<63>{}, IntT<NZ + 2>{}); if constexpr (!PEN) a_piece(c & 1, c + 2, 1); u_piece(0, 1, c + 1); SB();
            mfma2(0, IntT<1>{}, 1, v, bn); SB();
            col_comb(IntT<1>{}, d, vn); SB();
            // (0, 2) + column 2; the next operands are the second half's: their slice was requested a stage ago
            mfma2(0, IntT<2>{}, 0, v, bc); SB();
            col_read(SA, IntT<2>{}, d); WAIT(IntT<PEN ? 4 : 6>{}, IntT<15>{}); b_read(1, 0, bn); SB();
            WAIT(IntT<63>{}, IntT<NZ + 2>{}); if constexpr (!PEN) a_piece(c & 1, c + 2, 2); u_piece(0, 2, c + 1); SB();
            mfma2(0, IntT<2>{}, 1, v, bc); SB();
            col_comb(IntT<2>{}, d, vn); SB();
            // (1, 0) + column 3
            mfma2(1, IntT<0>{}, 0, v, bn); SB();
            col_read(SA, IntT<3>{}, d); WAIT(IntT<PEN ? 4 : 7>{}, IntT<15>{}); b_read(1, 1, bc); SB();
            WAIT(IntT<63>{}, IntT<NZ + 2>{}); u_piece(1, 0, c + 1); SB();
            mfma2(1, IntT<0>{}, 1, v, bn); SB();
            col_comb(IntT<3>{}, d, vn); SB();
            // (1, 1) + column 4
            mfma2(1, IntT<1>{}, 0, v, bc); SB();
            col_read(SA, IntT<4>{}, d); WAIT(IntT<PEN ? 4 : 7>{}, IntT<15>{}); b_read(1, 2, bn); SB();
            WAIT(IntT<63>{}, IntT<NZ + 2>{}); u_piece(1, 1, c + 1); SB();
            mfma2(1, IntT<1>{}, 1, v, bc); SB();
            col_comb(IntT<4>{}, d, vn); SB();
            // (1, 2)
            WAIT(IntT<63>{}, IntT<0>{}); u_piece(1, 2, c + 1); SB();
            mfma4(1, IntT<2>{}, v, bn); SB();
            WAIT(IntT<3>{}, IntT<15>{});                   // the patches of stage c + 2 and the first half's slices of stage c + 1
            SB();
            b_read(0, 0, bc);                              // first operands of stage c + 1: in flight across the barrier
#pragma unroll
            for (int p = 0; p < 3; ++p) v[p] = vn[p];
            SB();
            // every wave's patch reads of this stage are in registers (combined above); the operand read just issued is
            // wave-private and stays in flight across the barrier
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            SB();
        };
#pragma unroll 1
        for (int c = 0; c + 2 < n8; ++c) stage_body(c, FalseT{});
        stage_body(n8 - 2, TrueT{});
        {
            // Last stage: nothing left to prepare.  Both patch slots are free from here on, and the waves 8..11 (XI >= 4) -- the
            // ones without a part in the output transform -- request the patches of the NEXT tile block's first stage into slot
            // 0 now, one or two pieces behind each group of MFMAs: they are the block's only loads that miss L2 (every 128-byte
            // line of the tile block's raw pixels comes from HBM with them; a CU gets ~13 bytes per clock of those), and asked for
            // at the top of the tile block they took 12 000 cycles with nothing to overlap.  From here they have this stage and
            // the whole output transform.  Nine pieces per wave whether or not a next tile block exists (out-of-range addresses
            // then): the counted waits below are compile-time numbers.
            constexpr bool LOADER = XI >= 4;
            auto pf_a = [&](auto QT) {
                if constexpr (LOADER) {
                    constexpr int q = decltype(QT)::value, ij = (2 * XI + NH - 8) * 9 + q;
                    const unsigned vo = reinterpret_cast<const unsigned*>(smem + W4_PF_BASE)[((2 * XI + NH - 8) * 9 + q) * 64 + lane];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (__attribute__((address_space(3))) void*)(smem + ij * 256), 16, (int)vo, 0,
                                                             0, 0);
                }
            };
            b_read(0, 1, bn); SB(); mfma4(0, IntT<0>{}, v, bc); SB(); pf_a(IntT<0>{}); pf_a(IntT<1>{}); SB();
            b_read(0, 2, bc); SB(); mfma4(0, IntT<1>{}, v, bn); SB(); pf_a(IntT<2>{}); pf_a(IntT<3>{}); SB();
            // the second half's slices are older than the four pieces just requested
            WAIT(IntT<LOADER ? 4 : 0>{}, IntT<15>{}); b_read(1, 0, bn); SB(); mfma4(0, IntT<2>{}, v, bc); SB(); pf_a(IntT<4>{}); pf_a(IntT<5>{}); SB();
            b_read(1, 1, bc); SB(); mfma4(1, IntT<0>{}, v, bn); SB(); pf_a(IntT<6>{}); pf_a(IntT<7>{}); SB();
            b_read(1, 2, bn); SB(); mfma4(1, IntT<1>{}, v, bc); SB(); pf_a(IntT<8>{}); SB();
            mfma4(1, IntT<2>{}, v, bn); SB();
            lds_barrier();                                 // LDS only: the nine pieces stay in flight into the output transform
            SB();
        }
    };
    // prologue: patches of stage 0 and both filter halves of stage 0 (the first two came with the previous tile block's
    // epilogue when have0; the second half and the patches of stage 1 then land during the first transform)
    if (!have0) {
        issue_a(0, 0);
        issue_u(0, 0);
    }
    issue_u(1, 0);
    if (!have0) __syncthreads();
    switch (wave) {          // wave-uniform; every copy executes the same barriers
        case 0: stage_loop(IntT<0>{}, IntT<0>{}); break;
        case 1: stage_loop(IntT<0>{}, IntT<1>{}); break;
        case 2: stage_loop(IntT<1>{}, IntT<0>{}); break;
        case 3: stage_loop(IntT<1>{}, IntT<1>{}); break;
        case 4: stage_loop(IntT<2>{}, IntT<0>{}); break;
        case 5: stage_loop(IntT<2>{}, IntT<1>{}); break;
        case 6: stage_loop(IntT<3>{}, IntT<0>{}); break;
        case 7: stage_loop(IntT<3>{}, IntT<1>{}); break;
        case 8: stage_loop(IntT<4>{}, IntT<0>{}); break;
        case 9: stage_loop(IntT<4>{}, IntT<1>{}); break;
        case 10: stage_loop(IntT<5>{}, IntT<0>{}); break;
        default: stage_loop(IntT<5>{}, IntT<1>{}); break;
    }

    // ---- output transform ------------------------------------------------------------------------------------------
    // The 36 positions of a (tile, channel) meet through LDS in FOUR rounds (channel half jn, tile half th) of 16 tiles x 32
    // channels.  The exchange area of a round is 72 KiB and lives in patch slot 1 (positions 0..17) and filter sub-slot 1
    // (positions 18..35): patch slot 0 and filter sub-slot 0 stay free, and waves 8..11 -- which have no part in the
    // transform -- request the NEXT tile block's first stage into them right away (round 3 measured ~12 500 cycles per tile
    // block between requesting the first stage and having it, 4 800 of them behind the block's own output stores: 15 % of a
    // 64-channel tile block).  A wave that stores never loads here and vice versa: vmcnt counts loads and stores together and
    // they complete out of order with respect to each other, so a wave doing both would have to wait for its stores.
    //   E[pos][pair 8][64 dwords]: dword (2 c + 4 (pair & 1)) % 64 + e for channel c, tile 2 pair + e -- a wave writes the
    //   two adjacent tile rows an accumulator register pair holds as one ds_write_b64; the rotation of the odd pairs makes
    //   the transform's ds_read_b32 conflict free (lanes = 4 quads x 2 pairs x 2 tiles per channel-in-quad).
    //   Transform: lane (cq, Q', psel, e) of wave ww < 8 owns tile 2 (2 (ww / 2) + psel) + e and channel 16 (ww % 2) + 4 Q' +
    //   cq: reads its 36 values, applies A^T M A, adds the bias, accumulates the BatchNorm partials; then the four lanes cq =
    //   0..3 (16 apart) transpose their 4 x 4 (pixel row x channel) blocks with v_permlane32_swap / v_permlane16_swap so that
    //   lane cq holds pixel row cq x four consecutive channels, and stores FOUR 16-byte pieces instead of sixteen 4-byte ones
    //   (the output stores were issue bound: 512 buffer_store_dword per tile block).  The BatchNorm input of the fused
    //   backward reduction (SM == 2) is loaded the same way and transposed back.
    // output-transform role of a lane of waves 0..7: channel quad Q' = lane & 3 of the wave's 16-channel half, channel-in-quad
    // cq = lane >> 4 (the four lanes of a 4x4 register transpose are 16 apart), tile pair psel, tile-in-pair e
    int lane_o = lane0;
    asm volatile("" : "+v"(lane_o));                         // derived here, not carried through the stage loops
    const int o_cq = lane_o >> 4, o_psel = (lane_o >> 2) & 1, o_e = (lane_o >> 3) & 1;
    const int o_c32 = 16 * (wave & 1) + 4 * (lane_o & 3) + o_cq;         // channel inside a 32-channel round
    float* const E0 = smem + W4_A_FLOATS;                    // positions 0..17
    float* const E1 = smem + W4_U_BASE + W4_U_FLOATS;        // positions 18..35
    constexpr bool partial = PART;                           // a slice of the channels: A^T M A of the slice goes to ypart (Wino4Args)
    const __amdgpu_buffer_rsrc_t ysrd =
        partial ? __builtin_amdgcn_make_buffer_rsrc((void*)a.ypart, 0, (int)((size_t)gridDim.x * W4_TILES * 16 * 64 * 4), 0x00020000)
                : __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (int)((size_t)a.N * a.H * a.W * a.Cout * 4), 0x00020000);
    const int so_x = partial ? 64 * 4 : a.Cout * 4;
    const bool worker = wave < 8;
    const bool has_next = lt2 >= 0;
    // the claim after next, issued now by one lane of wave 11; its answer is read at the end of the block, behind the wait for the
    // next block's first stage (vmcnt(0) of the non-transforming waves)
    int claim_k = 0;
    if (dyn && wave == 11 && has_next && lane0 == 0) claim_k = atomicAdd(a.work + xcd, 1);
    // The 36 filter slices of the next tile block's first half-stage (L2 hits), nine per wave 8..11, in four chunks, one behind
    // the barrier of every round while waves 0..7 transform (a wave sits in the ISSUE of LDS-DMA pieces for a few hundred
    // cycles each while the output stores drain: all eighteen pieces of a wave in one go held up the round's barrier for
    // everybody by 6 000 cycles).  Its patches were requested during the last stage (stage_loop).
    unsigned pf_uvoff = 0;
    if (!worker && has_next) {
        const int nb2 = xcd_remap(lt2, total_blocks) % a.nblocks;
        pf_uvoff = (unsigned)((lane >> 5) * a.Cout * 8 + (nb2 * 64 + (lane & 31) * 2) * 8);
    }
    auto prefetch_chunk = [&](auto QLO, auto QHI) {          // slices QLO .. QHI - 1 of this wave
#pragma unroll
        for (int q = decltype(QLO)::value; q < decltype(QHI)::value; ++q) {
            const int ij = (wave - 8) * 9 + q;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(usrd, (__attribute__((address_space(3))) void*)(smem + W4_U_BASE + ij * 256),
                                                     16, (int)pf_uvoff, (ij * c8_total * 2) * a.Cout * 16, 0, 0);
        }
    };
    const int o_prl = 2 * (wave >> 1) + o_psel;                                       // tile pair inside a round
    const int o_rd = o_prl * 64 + (((2 * o_c32 + 4 * o_psel) & 63) + o_e);            // dwords inside a position's 512
    const int w_b0 = half * 128 + ((2 * l31) & 63), w_b1 = half * 128 + ((2 * l31 + 4) & 63);
    float stv[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float stq[2][2][4];                                      // SM == 2: [channel half][which][channel of the quad]
    const bool srelu = SM == 1 && a.stat_mode == 2;
    __amdgpu_buffer_rsrc_t bxsrd = ysrd;
    if constexpr (SM == 2)
        bxsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.bb.x, 0, (int)((size_t)a.N * a.H * a.W * a.Cout * 4), 0x00020000);
    // 4 x 4 transpose across the lanes cq = 0..3 (16 apart): v[4 g + k] of lane cq = element (g, cq) -> element (cq, g).
    // v_permlane32_swap a, b: a's lanes 32..63 <-> b's lanes 0..31; v_permlane16_swap: a's odd rows of 16 lanes <-> b's even rows
    // (scripts/probes/permlane_swap_probe.hip).  Inline asm, not __builtin_amdgcn_permlane32_swap / 16_swap: hipcc (ROCm 7.2)
    // merges the second result of a chained pair of the builtins with the first (half the swaps disappear and the four stored
    // channels come out equal).  `s_nop 1` = the two wait states a VALU write of either operand needs ahead of the swap's read.
    auto swap32 = [](float& x, float& y) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y)); };
    auto swap16 = [](float& x, float& y) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y)); };
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        float st0 = 0.f, st1 = 0.f;
        float sq0[4] = {0.f, 0.f, 0.f, 0.f}, sq1[4] = {0.f, 0.f, 0.f, 0.f};      // SM == 2: per channel of the lane's quad
#pragma unroll
        for (int th = 0; th < 2; ++th) {
            // this round's tile of a transforming lane
            unsigned ok = 0, qbase = 0, okq = 0;     // ok: bit 4 y + x = the pixel exists; q*: pixel row o_cq, four channels (after the transpose)
            float xl[16];
            if (worker) {
                const int T = T0 + 16 * th + 2 * o_prl + o_e;
                const bool tok = T < a.tiles;
                const int n = (int)(((float)T + 0.5f) * a.inv_tpi), rem = T - n * a.TY * a.TX;
                const int ty = (int)(((float)rem + 0.5f) * a.inv_tx), tx = rem - ty * a.TX;
                const int oy = 4 * ty, ox = 4 * tx;
#pragma unroll
                for (int yy = 0; yy < 4; ++yy)
#pragma unroll
                    for (int xx = 0; xx < 4; ++xx)
                        if (tok && oy + yy < a.H && ox + xx < a.W) ok |= 1u << (4 * yy + xx);
                const int qch = n0 + jn * 32 + 16 * (wave & 1) + 4 * (lane_o & 3);
                qbase = (unsigned)((((n * a.H + oy + o_cq) * a.W + ox) * a.Cout + qch) * 4);
                okq = (ok >> (4 * o_cq)) & 15u;
                if (partial) {           // [workgroup][tile][pixel 4 y + x][channel of the block]
                    qbase = (unsigned)(((((int)blockIdx.x * W4_TILES + 16 * th + 2 * o_prl + o_e) * 16 + 4 * o_cq) * 64 + qch - n0) * 4);
                    okq = 15u;
                }
            }
            auto load_x = [&]() {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 xv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                   bxsrd, (okq >> k) & 1 ? (int)qbase : (int)0x80000000u, k * so_x, 0));
#pragma unroll
                    for (int c = 0; c < 4; ++c) xl[4 * c + k] = xv[c];
                }
            };
            if (jn + th != 0) lds_barrier();          // the previous round's reads are done before this one overwrites them
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int pos = wave * 3 + p;
                float* Ep = pos < 18 ? E0 + pos * 512 : E1 + (pos - 18) * 512;
#pragma unroll
                for (int rr = 0; rr < 8; rr += 2) {   // accumulator rows (rr & 3) + 8 (rr >> 2) + 4 half of tile half th, and the next one
                    const int r = 8 * th + rr;
                    const int cpair = ((rr & 3) >> 1) + 4 * (rr >> 2);
                    *reinterpret_cast<f32x2*>(Ep + cpair * 64 + ((cpair & 1) ? w_b1 : w_b0)) = f32x2{acc[p][jn][r], acc[p][jn][r + 1]};
                }
            }
            lds_barrier();
            if (!worker && has_next) {
                if (jn == 0 && th == 0) prefetch_chunk(IntT<0>{}, IntT<3>{});
                if (jn == 0 && th == 1) prefetch_chunk(IntT<3>{}, IntT<6>{});
                if (jn == 1 && th == 0) prefetch_chunk(IntT<6>{}, IntT<8>{});
                if (jn == 1 && th == 1) prefetch_chunk(IntT<8>{}, IntT<9>{});
            }
            if (worker) {
                const float bz = smem[W4_BIAS_BASE + jn * 32 + o_c32];
                if constexpr (SM == 2) load_x();        // the BatchNorm input at this tile's pixels: in flight during the reads
                float m[36];
#pragma unroll
                for (int p = 0; p < 36; ++p) m[p] = (p < 18 ? E0 + p * 512 : E1 + (p - 18) * 512)[o_rd];
                // s[xi][x] = sum_nu A^T[x][nu] M[xi][nu], then y[yy][x] = sum_xi A^T[yy][xi] s[xi][x] + bias
                float sx[6][4];
#pragma unroll
                for (int xi = 0; xi < 6; ++xi) {
                    const float in[6] = {m[xi * 6], m[xi * 6 + 1], m[xi * 6 + 2], m[xi * 6 + 3], m[xi * 6 + 4], m[xi * 6 + 5]};
                    at4_apply(in, sx[xi]);
                }
                float yv[16];
#pragma unroll
                for (int xx = 0; xx < 4; ++xx) {
                    const float in[6] = {sx[0][xx], sx[1][xx], sx[2][xx], sx[3][xx], sx[4][xx], sx[5][xx]};
                    float col[4];
                    at4_apply(in, col);
#pragma unroll
                    for (int yy = 0; yy < 4; ++yy) yv[yy * 4 + xx] = col[yy] + bz;
                }
                if constexpr (SM == 1) {
                    const float pv = srelu ? fmaxf(bz, 0.f) : bz;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const float val = srelu ? fmaxf(yv[k], 0.f) : yv[k];
                        const float d = (ok >> k) & 1 ? val - pv : 0.f;
                        st0 += d;
                        st1 = fmaf(d, d, st1);
                    }
                }
                // (pixel row g, channel cq) -> (pixel row cq, channel g)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    swap32(yv[k], yv[8 + k]);
                    swap32(yv[4 + k], yv[12 + k]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    swap16(yv[k], yv[4 + k]);
                    swap16(yv[8 + k], yv[12 + k]);
                }
                if constexpr (SM == 2) {
                    // The fused BatchNorm-backward reduction, in the transposed domain: this lane now holds pixel row o_cq x the four
                    // channels of its quad, exactly as the BatchNorm input was loaded (xl[4 c + k] = channel c, pixel column k) -- no
                    // transpose back; the four lanes of a quad each carry a partial of the same four channels.
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const f32x4 bnc = *reinterpret_cast<const f32x4*>(smem + W4_BN_BASE + (jn * 32 + 16 * (wave & 1) + 4 * (lane_o & 3) + c) * 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float xv = xl[4 * c + k];
                            const bool pass = a.bb.relu != 1 || fmaf(xv, bnc[0], bnc[1]) > 0.f;
                            const float d = ((okq >> k) & 1) && pass ? yv[4 * c + k] : 0.f;
                            sq0[c] += d;
                            sq1[c] = fmaf(d, (xv - bnc[2]) * bnc[3], sq1[c]);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    __builtin_amdgcn_raw_buffer_store_b128(
                        __builtin_bit_cast(u32x4, f32x4{yv[k], yv[4 + k], yv[8 + k], yv[12 + k]}), ysrd,
                        (okq >> k) & 1 ? (int)qbase : (int)0x80000000u, k * so_x, 0);
            }
        }
        if constexpr (SM == 1) {
            stv[jn][0] = st0;
            stv[jn][1] = st1;
        }
        if constexpr (SM == 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                stq[jn][0][c] = sq0[c];
                stq[jn][1][c] = sq1[c];
            }
        }
    }
    if constexpr (STATS) {
        // red[round 2][which 2][slot NS][cout 32] -> one partial per (tile block, channel), the NS slots of a channel summed in
        // order; once per tile block, both channel halves together.  SM == 1: 16 slots (wave pair, tile pair, tile); SM == 2: 64
        // (... and the pixel row o_cq).  In patch slot 1 (slot 0 is being filled for the next tile block).
        constexpr int NS = SM == 2 ? 64 : 16;
        lds_barrier();
        float* red = E0;
        if (worker) {
            const int slot = (wave >> 1) * 4 + o_psel * 2 + o_e;
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    if constexpr (SM == 2) {
                        *reinterpret_cast<f32x4*>(red + ((jn * 2 + which) * NS + o_cq * 16 + slot) * 32 + 16 * (wave & 1) + 4 * (lane_o & 3)) =
                            f32x4{stq[jn][which][0], stq[jn][which][1], stq[jn][which][2], stq[jn][which][3]};
                    } else {
                        red[((jn * 2 + which) * NS + slot) * 32 + o_c32] = stv[jn][which];
                    }
                }
        }
        lds_barrier();
        if (t < 128) {
            const int c32 = t & 31, which = (t >> 5) & 1, jn = t >> 6;
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < NS; ++r) sum += red[((jn * 2 + which) * NS + r) * 32 + c32];
            a.stat_part[((size_t)mb * 2 + which) * a.Cout + n0 + jn * 32 + c32] = sum;
        }
    }
    if (!worker) __builtin_amdgcn_s_waitcnt(0x0070 | (0xF << 8));     // vmcnt(0): the next tile block's first stage landed
    if (dyn && wave == 11 && lane0 == 0) {
        asm volatile("" ::: "memory");
        int nx = -1;
        if (has_next) {
            nx = wq_base(a.lt_first, xcd) + 8 * claim_k;
            if (nx >= a.lt_end) nx = wq_claim(a.work, xcd, 1, a.lt_first, a.lt_end);      // own queue empty: the other XCDs' in turn
        }
        Qw[it & 1] = nx;
    }
    have0 = has_next;
    }   // tile-block loop
    if (dyn && t == 0) wq_leave(a.work, (int)gridDim.x);
}

// U[pos][c/8][(c/2)%2][(c/4)%2][k][c%2] = (G g G^T)[pos] for every (input channel c, output channel k); G of F(4x4,3x3).
// from_fwd_for_dgrad: g[kh][kw][c][k] = w[2-kh][2-kw][k][c] with w the FORWARD filter (Cin_fwd = Cout here).
__global__ __launch_bounds__(256) void wino4_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin,
                                                            int Cout, int dgrad) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int k = idx % Cout, c = idx / Cout;
    float g[3][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            g[kh][kw] = dgrad ? w[((size_t)((2 - kh) * 3 + (2 - kw)) * Cout + k) * Cin + c]
                              : w[((size_t)(kh * 3 + kw) * Cin + c) * Cout + k];
    // G = [1 0 0; -1/3 -1/3 -1/3; 1/3 -1/3 1/3; 1/15 2/15 4/15; -16/15 8/15 -4/15; 0 0 1]
    auto gmul = [](float a0, float a1, float a2, float* o) {
        const float s02 = a0 + a2;
        o[0] = a0;
        o[1] = (-1.f / 3.f) * (s02 + a1);
        o[2] = (1.f / 3.f) * (s02 - a1);
        o[3] = (1.f / 15.f) * fmaf(2.f, a1, fmaf(4.f, a2, a0));
        o[4] = (-4.f / 15.f) * fmaf(-2.f, a1, fmaf(4.f, a0, a2));
        o[5] = a2;
    };
    float gg[6][3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        float o[6];
        gmul(g[0][kw], g[1][kw], g[2][kw], o);
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) gg[xi][kw] = o[xi];
    }
    const int c8 = c >> 3, hf = (c >> 2) & 1, sub = (c >> 1) & 1, c2 = c & 1;
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) {
        float o[6];
        gmul(gg[xi][0], gg[xi][1], gg[xi][2], o);
#pragma unroll
        for (int nu = 0; nu < 6; ++nu)
            u[(((((size_t)(xi * 6 + nu) * (Cin >> 3) + c8) * 2 + sub) * 2 + hf) * Cout + k) * 2 + c2] = o[nu];
    }
}

// The tail of a layer whose tile blocks do not divide by the CU count (conv_wino4_launch): sums the channel slices of every
// tail tile block in slice order, adds the bias, stores the valid pixels and leaves the block's BatchNorm partials exactly where
// the one-pass epilogue leaves them (SM as conv_wino4_kernel; another, fixed, summation order).  Statistics are per channel, so a
// tail tile block is cut into FOUR workgroups of 16 channels (a single workgroup per block read its 2 MiB of slices from one
// CU: 500 us); thread = (pixel slot, channel quad): 16-byte loads of the slices (independent: the loop over slices unrolls),
// 16-byte stores, eight (tile, pixel) pairs per thread.
template <int SM>
__global__ __launch_bounds__(256) void wino4_tail_reduce_kernel(Wino4Args a) {
    __shared__ float red[2][64][16];
    const int t = threadIdx.x, q = t & 3, slot = t >> 2;          // channel quad of the group, (tile, pixel) slot 0..63
    const int blk = (int)blockIdx.x >> 2, cg = (int)blockIdx.x & 3;
    const int lt = a.lt_first + blk;
    const int logical = xcd_remap(lt, a.mblocks * a.nblocks);
    const int nb = logical % a.nblocks, mb = logical / a.nblocks;
    const int n0 = nb * 64, T0 = mb * W4_TILES;
    const int cl = cg * 16 + 4 * q;                                // first of this thread's four channels inside the block
    const int ch = n0 + cl;
    f32x4 bz = {0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) bz = *reinterpret_cast<const f32x4*>(a.bias + ch);
    f32x4 bsc = bz, bsh = bz, bmu = bz, brs = bz;
    if constexpr (SM == 2) {
        bsc = *reinterpret_cast<const f32x4*>(a.bb.scale + ch);
        bsh = *reinterpret_cast<const f32x4*>(a.bb.shift + ch);
        bmu = *reinterpret_cast<const f32x4*>(a.bb.mean + ch);
        const f32x4 var = *reinterpret_cast<const f32x4*>(a.bb.var + ch);
#pragma unroll
        for (int c = 0; c < 4; ++c) brs[c] = rsqrtf(var[c] + a.bb.eps);
    }
    const bool srelu = SM == 1 && a.stat_mode == 2;
    float st0[4] = {0.f, 0.f, 0.f, 0.f}, st1[4] = {0.f, 0.f, 0.f, 0.f};
    const float* part = a.ypart + (size_t)blk * a.ksplit * (W4_TILES * 16 * 64) + cl;
    for (int it = 0; it < 8; ++it) {
        const int tp = it * 64 + slot, tl = tp >> 4, px = tp & 15;          // (tile, pixel) of the block
        const int T = T0 + tl;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int sl = 0; sl < a.ksplit; ++sl) sum += *reinterpret_cast<const f32x4*>(part + ((size_t)sl * W4_TILES * 16 + tp) * 64);
        const int n = (int)(((float)T + 0.5f) * a.inv_tpi), rem = T - n * a.TY * a.TX;
        const int ty = (int)(((float)rem + 0.5f) * a.inv_tx), tx = rem - ty * a.TX;
        const int yy = 4 * ty + (px >> 2), xx = 4 * tx + (px & 3);
        if (T < a.tiles && yy < a.H && xx < a.W) {
            const size_t o = (((size_t)n * a.H + yy) * a.W + xx) * a.Cout + ch;
            const f32x4 val = sum + bz;
            *reinterpret_cast<f32x4*>(a.y + o) = val;
            if constexpr (SM == 1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float pv = srelu ? fmaxf(bz[c], 0.f) : bz[c];
                    const float d = (srelu ? fmaxf(val[c], 0.f) : val[c]) - pv;
                    st0[c] += d;
                    st1[c] = fmaf(d, d, st1[c]);
                }
            }
            if constexpr (SM == 2) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(a.bb.x + o);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool pass = a.bb.relu != 1 || fmaf(xv[c], bsc[c], bsh[c]) > 0.f;
                    const float d = pass ? val[c] : 0.f;
                    st0[c] += d;
                    st1[c] = fmaf(d, (xv[c] - bmu[c]) * brs[c], st1[c]);
                }
            }
        }
    }
    if constexpr (SM != 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            red[0][slot][4 * q + c] = st0[c];
            red[1][slot][4 * q + c] = st1[c];
        }
        __syncthreads();
        if (t < 32) {
            const int which = t >> 4, c = t & 15;
            float sum = 0.f;
            for (int r = 0; r < 64; ++r) sum += red[which][r][c];
            a.stat_part[((size_t)mb * 2 + which) * a.Cout + n0 + cg * 16 + c] = sum;
        }
    }
}

// Scratch of the tail launches of callers that bring none (the stand-alone operator entry points; an engine owns its buffer and
// passes it in ConvGeom::tail_scratch): one buffer per (device, stream), grown on demand (hipMalloc synchronises: only until the
// largest layer has been seen once), never freed.
float* tail_scratch(int dev, hipStream_t s, size_t bytes) {
    struct Buf {
        int dev;
        hipStream_t s;
        float* p;
        size_t cap;
    };
    static std::mutex mu;
    static std::vector<Buf> pool;
    std::lock_guard<std::mutex> lock(mu);
    for (auto& b : pool)
        if (b.dev == dev && b.s == s) {
            if (b.cap < bytes) {
                (void)hipStreamSynchronize(s);
                (void)hipFree(b.p);
                b.p = nullptr;
                b.cap = 0;
                if (hipMalloc((void**)&b.p, bytes) != hipSuccess) return nullptr;
                b.cap = bytes;
            }
            return b.p;
        }
    Buf b{dev, s, nullptr, 0};
    if (hipMalloc((void**)&b.p, bytes) != hipSuccess) return nullptr;
    b.cap = bytes;
    pool.push_back(b);
    return b.p;
}

}  // namespace

int* persistent_work_counters(hipStream_t s) {
    struct Slot {
        int dev;
        hipStream_t s;
        int* p;
    };
    static std::mutex mu;
    static std::vector<Slot> table;
    static int* arena[L3_MAX_DEVICES] = {nullptr};
    static int used[L3_MAX_DEVICES] = {0};
    constexpr int SLOTS = 1024, INTS = 16;
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= L3_MAX_DEVICES - 1;
    std::lock_guard<std::mutex> lock(mu);
    for (auto& e : table)
        if (e.dev == dev && e.s == s) return e.p;
    if (arena[dev] == nullptr) {
        if (hipMalloc((void**)&arena[dev], SLOTS * INTS * sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemset(arena[dev], 0, SLOTS * INTS * sizeof(int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
    }
    if (used[dev] >= SLOTS) return nullptr;
    table.push_back(Slot{dev, s, arena[dev] + (size_t)INTS * used[dev]++});
    return table.back().p;
}

namespace {

template <int SM>
void launch_wino4(const Wino4Args& a, hipStream_t s) {
    static std::once_flag once[L3_MAX_DEVICES];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & (L3_MAX_DEVICES - 1)], [] {
        (void)hipFuncSetAttribute((const void*)conv_wino4_kernel<SM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W4_LDS_BYTES);
    });
    static const int persist_env = l3_knob("L3_WINO_PERSIST") ? atoi(l3_knob("L3_WINO_PERSIST")) : -1;
    const int persist = persist_env >= 0 ? persist_env : 1;
    static int cus[L3_MAX_DEVICES] = {0};
    int& ncu = cus[dev & (L3_MAX_DEVICES - 1)];
    if (ncu == 0) {
        hipDeviceProp_t prop;
        ncu = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8 ? prop.multiProcessorCount / 8 * 8 : 256;
    }
    // L3_W4_NCU (debug knob, read per call: the tests emulate a small chip to reach the tail path with small tensors)
    const int ncu_eff = l3_knob("L3_W4_NCU") ? atoi(l3_knob("L3_W4_NCU")) / 8 * 8 > 0 ? atoi(l3_knob("L3_W4_NCU")) / 8 * 8 : ncu : ncu;
    const int total = a.mblocks * a.nblocks;
    // L3_W4_GRID (debug knob): persistent grid smaller than the chip, for the co-residency A/B of DESIGN.md 4c (CUs left to the
    // other tower's HBM-bound kernels); multiples of 8 keep the XCD mapping of xcd_remap
    static const int grid_cap = l3_knob("L3_W4_GRID") ? atoi(l3_knob("L3_W4_GRID")) / 8 * 8 : 0;
    const int grid = grid_cap >= 8 && grid_cap < ncu_eff ? grid_cap : ncu_eff;
    Wino4Args m = a;
    m.lt_first = 0;
    m.lt_end = total;
    m.ksplit = 1;
    m.ypart = nullptr;
    // ---- the tail ------------------------------------------------------------------------------------------------------
    // total = q * grid + R tile blocks: after q full rounds R < grid CUs would work through a whole tile block each while the
    // others idle -- at 64 pairs 784 tile blocks on 256 CUs (V.conv4a/4b) are 3.06 rounds, i.e. four: 0.49-0.52 of peak where the
    // audio tower's 768 (3.00 rounds) reach 0.61-0.66.  The R tail blocks instead go to a second launch of grid / R workgroups
    // EACH, every one taking a slice of the input channels and leaving A^T M A of its slice in scratch; a small kernel sums the
    // slices, adds the bias, stores and takes the statistics.  Worth it when the slices are long enough to pay for the block's
    // fixed cost (first fetch, output transform: ~25 000 cycles) and the extra pass: stages saved per tail block >= 13
    // (measured per layer at 64 pairs, profiles/r04_wino4_tail.txt: -6 ... -16 % where taken, +-1 % at 12).  Only for launches with
    // nothing queued beside them (ConvGeom::solo).
    const int tail_env = l3_knob("L3_W4_TAIL") ? atoi(l3_knob("L3_W4_TAIL")) : 1;        // (debug knob, read per call: the tests switch it)
    const int tail_on = tail_env == 2 || (tail_env == 1 && a.solo);        // L3_W4_TAIL: 0 never, 1 solo launches (default), 2 always
    const int R = persist && tail_on ? total % grid : 0;
    int split = R > 0 ? grid / R : 1;
    if (split > a.nchunks / 2) split = a.nchunks / 2;              // >= 2 stages per slice (the pipeline's shortest loop)
    const bool tail = split >= 2 && a.nchunks * (split - 1) >= 13 * split;
    if (tail) m.lt_end = total - R;
    // more tile blocks than workgroups: the blocks are handed out by work counters (device_common.h wq_*), so that a workgroup
    // that starts late -- a collective holding its CU -- costs its share and not a second round.  L3_W4_DYNAMIC=0: static stride (A/B)
    const int dyn_on = l3_knob("L3_W4_DYNAMIC") ? atoi(l3_knob("L3_W4_DYNAMIC")) : a.dynamic;       // (read per call: the tests switch it)
    m.work = persist && m.lt_end > grid && dyn_on ? persistent_work_counters(s) : nullptr;
    if (m.lt_end > 0)
        hipLaunchKernelGGL((conv_wino4_kernel<SM>), dim3(persist && m.lt_end > grid ? grid : m.lt_end), dim3(W4_THREADS), W4_LDS_BYTES, s, m);
    if (tail) {
        static std::once_flag once0[L3_MAX_DEVICES];
        std::call_once(once0[dev & (L3_MAX_DEVICES - 1)], [] {
            (void)hipFuncSetAttribute((const void*)conv_wino4_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W4_LDS_BYTES);
        });
        Wino4Args tl = a;
        tl.work = nullptr;
        tl.lt_first = total - R;
        tl.lt_end = total;
        tl.ksplit = split;
        const size_t need = (size_t)R * split * W4_TILES * 16 * 64 * sizeof(float);
        tl.ypart = a.tail_buf != nullptr && a.tail_cap >= need ? a.tail_buf : tail_scratch(dev, s, need);
        tl.stat_part = nullptr;
        if (tl.ypart == nullptr) {          // no scratch: the plain way
            Wino4Args rest = m;
            rest.work = nullptr;
            rest.lt_first = total - R;
            rest.lt_end = total;
            hipLaunchKernelGGL((conv_wino4_kernel<SM>), dim3(R), dim3(W4_THREADS), W4_LDS_BYTES, s, rest);
            return;
        }
        hipLaunchKernelGGL((conv_wino4_kernel<0, true>), dim3(R * split), dim3(W4_THREADS), W4_LDS_BYTES, s, tl);
        tl.stat_part = a.stat_part;
        hipLaunchKernelGGL((wino4_tail_reduce_kernel<SM>), dim3(R * 4), dim3(256), 0, s, tl);
    }
}

}  // namespace

// F(4x4,3x3) is taken where it pays.  The per-block epilogue -- 36 positions through LDS -- is amortised over Cin / 4
// stages; measured on the training step (profiles/r03_wino4_min_cin.txt): Cin >= 128 only 1660 pairs/s, Cin >= 64 (all 14
// layers) 1690.  L3_WINO4 (debug knob, read per call: the tests switch it): 0 = never, n = minimum Cin.
bool conv_wino4_selected(const ConvGeom& g) {
    const char* env = l3_knob("L3_WINO4");
    const int min_cin = env ? atoi(env) : 64;
    return !g.f2x2 && min_cin > 0 && g.Cin >= min_cin && g.Cin >= 16 && g.Cin % 8 == 0 && g.Cout % 64 == 0 &&
           (size_t)36 * g.Cin * g.Cout * 4 < (1ull << 31);
}

double conv_wino4_executed_flops(const ConvGeom& g) {
    return 2.0 * 36.0 * (double)g.N * ((g.H + 3) / 4) * ((g.W + 3) / 4) * (double)g.Cin * (double)g.Cout;
}

// R tail blocks x grid / R slices <= one workgroup per CU (<= 304 on any gfx9 part), each leaving 32 tiles x 16 pixels x 64 channels
size_t conv_wino4_tail_scratch_bytes() { return (size_t)304 * W4_TILES * 16 * 64 * sizeof(float); }

int conv_wino4_blocks(const ConvGeom& g, int n) { return (n * ((g.H + 3) / 4) * ((g.W + 3) / 4) + W4_TILES - 1) / W4_TILES; }

void conv_wino4_transform_weights(const float* w, float* u, const ConvGeom& g, bool from_fwd_for_dgrad, hipStream_t s) {
    const int total = g.Cin * g.Cout;
    hipLaunchKernelGGL(wino4_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, u, g.Cin, g.Cout,
                       from_fwd_for_dgrad ? 1 : 0);
}

// n samples of geometry g starting at x / y; writes conv_wino4_blocks(g, n) statistics blocks
void conv_wino4_launch(const float* x, const float* u, const float* bias, float* y, const ConvGeom& g, int n, hipStream_t s,
                       float* stat_part, int stat_mode, const BnBwdFuse* bn_bwd) {
    Wino4Args a;
    a.solo = g.solo;
    a.dynamic = g.dynamic;
    a.tail_buf = g.tail_scratch;
    a.tail_cap = g.tail_scratch_bytes;
    a.x = x; a.u = u; a.bias = bias; a.y = y;
    a.N = n; a.H = g.H; a.W = g.W; a.Cin = g.Cin; a.Cout = g.Cout;
    a.TY = (g.H + 3) / 4;
    a.TX = (g.W + 3) / 4;
    a.tiles = n * a.TY * a.TX;
    a.mblocks = conv_wino4_blocks(g, n);
    a.nblocks = g.Cout / 64;
    a.nchunks = g.Cin / 8;
    a.inv_tpi = 1.0f / (float)(a.TY * a.TX);
    a.inv_tx = 1.0f / (float)a.TX;
    a.stat_part = stat_part;
    a.stat_mode = stat_mode;
    a.bb = BnBwdFuse{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0};
    a.stagger_cycles = 0;
    a.stagger_groups = 1;
    a.lt_first = 0;
    a.lt_end = a.mblocks * a.nblocks;
    a.ksplit = 1;
    a.work = nullptr;
    a.ypart = nullptr;
    if (const char* sg = l3_knob("L3_W4_STAGGER")) {        // "<cycles>,<groups>"
        a.stagger_cycles = atoi(sg);
        const char* c = strchr(sg, ',');
        a.stagger_groups = c ? atoi(c + 1) : 4;
        if (a.stagger_groups < 1) a.stagger_groups = 1;
    }
    if (bn_bwd != nullptr && stat_part != nullptr) a.bb = *bn_bwd;
    if (a.stat_part != nullptr && a.bb.x != nullptr)
        launch_wino4<2>(a, s);
    else if (a.stat_part != nullptr)
        launch_wino4<1>(a, s);
    else
        launch_wino4<0>(a, s);
}

}  // namespace l3
