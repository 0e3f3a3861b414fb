// conv_bf16_halo.hip -- mixed-precision 3x3 'same' convolution with an LDS-resident input halo.
//
// Same arithmetic as conv_bf16.hip's stored-operand kernel (bf16 x bf16 products, fp32 accumulate:
// BASELINE.json configs[4]; Conv2D layers of l3embedding/audio_model.py:372-445, vision_model.py:126-205
// with Cin % 64 == 0 and Cout % 64 == 0, forward and data gradient), restructured around what rocprof
// showed that kernel to be bound by: the HBM/L2 -> LDS load path (8 LDS-DMA pieces per wave for every 16
// MFMAs: each tap re-fetched its own 128 x 64-channel A tile), not the matrix cores.
//
//   block  = one PH x PW patch of output pixels (256 pixels: 8 x 32 or 16 x 16, whichever pads the image
//            least) x 128 output channels with 8 waves as 4 (M) x 2 (N), or x 64 output channels with 4 waves
//            (the Cout = 64 layers: two such blocks share a CU and cover each other's prologue / epilogue);
//            every wave owns 64 pixels x 64 channels;
//   A      = the (PH+2) x (PW+2) input halo of the patch for one 64-channel chunk, fetched ONCE per chunk
//            (buffer_load ... lds, out-of-image pixels = out-of-range offsets = zeros) and read by all 9 taps:
//            a tap is a constant byte displacement of the lane's halo row, i.e. an immediate in the
//            ds_read_b128 -- 9x fewer A bytes and 9x fewer A load instructions than tap-by-tap tiles;
//            rows are 144 B apart (128 B of channels + 16 B pad) so the 16 lanes of a ds_read_b128 group,
//            which always read 16 rows that are distinct modulo 16, hit 16 different 16-B slots of the
//            256-B bank line (no XOR swizzle needed, so the tap displacement stays an immediate);
//   B      = per tap the 128 x 64 filter slice [flipped tap][Cout][Cin] (XOR-swizzled 128-B rows as in
//            conv_bf16.hip) through a 3-deep LDS ring: the slice for tap t+2 is issued while tap t computes;
//   sync   = one raw s_barrier per tap behind a COUNTED s_waitcnt vmcnt(n) (n = loads issued during this tap),
//            never __syncthreads(): its fence would drain the loads that are meant to stay in flight;
//   halo double buffer (128-channel blocks): the next chunk's halo is issued piece by piece over the first taps
//            of the current one; the 64-channel blocks keep ONE halo buffer (73 KB of LDS per block, so that two
//            blocks fit a CU) and refill it between chunks -- the neighbour block computes meanwhile.
//   flat tiles (MODE 3, round 5): a 2-D patch pads every image to a multiple of its sides -- 56 x 56 -> 56 x 64 (12.5 % of
//            the matrix work on zeros), 28 x 28 -> 32 x 32 (23 %), the audio tower's 64 x 49 -> 64 x 64 (23 %) and 32 x 24 ->
//            32 x 32 (25 %) -- and the chip is power-limited under this kernel (PMC: ~1.8 GHz), so wasted MFMAs cost twice.
//            A flat tile is 256 CONSECUTIVE pixels of the (n, y, x) order, whatever rows and images they span: nothing is
//            padded but the last tile of the tensor.  Its halo is every image row it touches plus one above and one below,
//            each with a zero column at both ends (LDS row = virtual row * (W + 2) + x + 1), with ONE zero row between two
//            images (virtual row v = n (H + 1) + y + 1; v % (H + 1) == 0 is that row: the lower halo of image n and the
//            upper halo of image n + 1) -- a tap is then the same displacement for every lane, as in the 2-D patch, and
//            the output tile is one contiguous [256][Cout] slab.  The halo image starts at the first tap of the first pixel
//            and ends at the last tap of the last one (256 + 2 (W + 2) + 2 per row crossed + W + 2 per zero row crossed: up to
//            442 pixels for W = 56 against 10 x 34 = 340), hence 32-channel chunks and one halo buffer.
#include "kernels.h"
#include "device_common.h"

#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <type_traits>

namespace l3 {

namespace {

struct HaloArgs {
    const void* x;          // bf16 NHWC
    const void* wn;         // bf16 [flipped tap][Cout][Cin]
    const float* bias;
    void* y;                // fp32 or bf16 NHWC
    int N, H, W, Cin, Cout;
    int pyt, pxt;           // patches per image (rows, columns)
    int patches, ntiles, nchunks;
    float* stat_part;       // [patch][2][Cout] BatchNorm partials about the pivot bias[c] (bn_fused.hip layout)
    int stat_mode;          // SM == 1: 1 = moments of y, 2 = moments of relu(y)
    BnBwdFuse bb;           // SM == 2: the launch is a data gradient, the partials are those of the BatchNorm backward
                            // reduction (kernels.h BnBwdFuse; bb.x is the bf16-stored BatchNorm input)
    // flat tiles (MODE 3): P = N * H * W pixels, halo_bytes of LDS for the largest halo (a whole number of 1-KiB pieces),
    // reciprocals for the exact float divisions of the prologue (every dividend is far below 2^22)
    int P, halo_bytes;
    float inv_w, inv_h, inv_w2, inv_h1;
};

// MODE 0: 64-channel chunks; 128-channel blocks double-buffer the halo (146 KiB, one block per CU), 64-channel blocks
//         keep one halo buffer (73 KiB, two per CU).
// MODE 1: 64-channel chunks; 128-channel blocks with one halo buffer and a 2-deep filter ring -- 80 KiB, two blocks per
//         CU (16 waves) that cover each other's prologue, halo refills and epilogue.
// MODE 2: 32-channel chunks (80-B halo rows, 64-B filter rows): 128-channel blocks double-buffer the halo in 78 KiB
//         (two per CU), 64-channel blocks fit four per CU (39 KiB) -- 16 waves per CU either way.
template <int PW, int WN, int MODE = 0>
struct HaloGeom {
    static constexpr int NWAVES = 4 * WN;                     // 4 (M) x WN (N) waves of 64 pixels x 64 channels
    static constexpr int BN = 64 * WN;
    static constexpr int PH = 256 / PW;
    static constexpr int PITCH = PW + 2;                      // halo rows per patch row (34 / 18)
    static constexpr int HROWS = (PH + 2) * PITCH;
    static constexpr bool FLAT = MODE >= 3;                   // 256 consecutive pixels instead of a PH x PW patch
    // MODE 4: flat tiles with a 4-deep filter ring: the slice of tap t + 1 is complete (for every wave) one barrier early, so a
    // wave reads its first operands of tap t + 1 BEFORE the barrier that ends tap t and the matrix pipe restarts without the
    // LDS round trip every wave of the block would otherwise pay at the same moment.
    static constexpr bool PRE = MODE == 4;
    // MODE 5: flat tiles with BOTH halo buffers: rows of 64 B without the pad slot (442 pixels = 28 KiB at W = 56: two of them and
    // a 3-deep filter ring are exactly 80 KiB), the 16-B slots of a row XOR-swizzled by (row >> 2) & 3 instead -- 16 consecutive
    // rows at one logical slot still cover the 16 slots of the bank line once; the tap displacement is no immediate on flat
    // tiles anyway.  The next chunk's halo arrives piece by piece over the first taps of the current one: no refill a block waits for.
    // Measured 1-4 % SLOWER than MODE 4 on all 16 launches it fits (profiles/r05_bf16_conv_notes.txt): the exposed refill is not what
    // parks the waves; it stays selectable (L3_HALO_FLAT_MODE=5), MODE 4 is the product path.
    static constexpr bool SWZA = MODE == 5;
    static constexpr int KC = MODE >= 2 ? 32 : 64;            // input channels per chunk
    static constexpr int ROWB = SWZA ? KC * 2 : KC * 2 + 16;  // bytes between halo rows (144 / 80): channels + 16 B pad; MODE 5: 64, no pad
    static constexpr int SLOTS = ROWB / 16;                   // 16-B slots per halo row, the last one is the pad (MODE 5: none)
    static constexpr int CSLOTS = SWZA ? SLOTS : SLOTS - 1;   // slots that carry channels
    static constexpr int PIECES = (HROWS * ROWB + 1023) / 1024;
    static constexpr int PER_WAVE = (PIECES + NWAVES - 1) / NWAVES;   // LDS-DMA pieces per wave and chunk
    // piece q of wave w is piece q * NWAVES + w of the halo image; MODE 2 allocates exactly PIECES (pieces beyond are
    // skipped), the older modes a whole number of pieces per wave
    static constexpr int HALO_BYTES = (MODE == 2 ? PIECES : PER_WAVE * NWAVES) * 1024;
    static constexpr int HALO_BUFS = (WN == 2 && MODE != 1 && MODE < 3) || MODE == 5 ? 2 : 1;
    static constexpr int RING = MODE == 1 ? 2 : MODE == 4 ? 4 : 3;     // filter slices in LDS; the slice RING-1 taps ahead is in flight
    static constexpr int FLAT_MAXQ = 7;                       // flat tiles: at most 7 halo pieces per wave (56 KiB for 8 waves)
    static constexpr int BROWB = KC * 2;                      // bytes per filter row (one output channel, KC inputs)
    static constexpr int B_BYTES = BN * BROWB;                // one tap
    static constexpr int BPW = B_BYTES / 1024 / NWAVES;       // filter pieces per wave and tap (2 / 1)
    static constexpr int LDS_BYTES = HALO_BUFS * HALO_BYTES + RING * B_BYTES;      // (flat tiles: the launch sizes the halo)
    static constexpr int BLOCKS_PER_CU = MODE == 2 ? (WN == 1 ? 4 : 2) : (WN == 1 || MODE == 1 || MODE >= 3 ? 2 : 1);
    static_assert(HALO_BUFS == 1 || (FLAT ? FLAT_MAXQ : PER_WAVE) <= 9, "one halo piece per tap at most");
    static_assert(FLAT || LDS_BYTES * BLOCKS_PER_CU <= 160 * 1024, "LDS budget");
    static_assert(FLAT || LDS_BYTES >= NWAVES * 32 * 64 * 4, "stage buffers must hold the epilogue");
    static_assert(!FLAT || (WN == 2 && PW == 32), "flat tiles: 128-channel blocks, M-tile = 32 consecutive pixels");
    // pixel of M row `row` (0..31) of m-tile `mt` (0..7) inside the patch
    __device__ static __forceinline__ void pixel(int mt, int row, int& py, int& px) {
        if constexpr (PW == 32) {
            py = mt;
            px = row;
        } else {                                              // two patch rows per m-tile; the second one is
            py = 2 * mt + (row >> 4);                         // rotated by 14 columns so that every 16-lane
            px = ((row & 15) + ((row >> 4) ? 14 : 0)) & 15;   // read group still sees 16 rows distinct mod 16
        }
    }
};

#ifdef L3_HALO_STAMPS
// Instrumented build (scripts/probes/build_halo_stamps.sh): wave 0 of every block stamps the 100-MHz wall clock at its phase
// boundaries -- entry, halo + first slices landed, end of chunk 0's taps, halo refilled, last tap done, tile stored, exit.
__device__ unsigned long long g_halo_stamps[(1 << 16) * 8];
#define HALO_STAMP(k) do { if (t == 0 && blockIdx.x < (1u << 16)) g_halo_stamps[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
// slot 7: where the block ran -- HW_ID (wave / SIMD / CU / SH / SE) and XCC_ID
#define HALO_STAMP_WHERE() do { if (t == 0 && blockIdx.x < (1u << 16)) g_halo_stamps[blockIdx.x * 8 + 7] = \
    (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(6164) << 32); } while (0)
#else
#define HALO_STAMP_WHERE() do { } while (0)
#define HALO_STAMP(k) do { } while (0)
#endif

constexpr int VMCNT(int n) { return (n & 0xF) | 0x70 | (0xF << 8) | ((n >> 4) << 14); }   // s_waitcnt vmcnt(n) only

// a / b for 0 <= a < 2^22 with inv = 1.0f / b: (a + 0.5) / b is at least 0.5 / b away from an integer, the float product is
// within a few 2^-24 of it relative -- exact for the prologue's quotients (all below 2^13 here)
__device__ __forceinline__ int fdiv(int a, float inv) { return (int)(((float)a + 0.5f) * inv); }

template <int PW, int WN, int SM, bool OBF, int MODE>
__global__ __launch_bounds__(256 * WN, (HaloGeom<PW, WN, MODE>::BLOCKS_PER_CU * WN)) void conv_bf16_halo_kernel(HaloArgs a) {
    constexpr bool STATS = SM == 1;
    static_assert(SM != 2 || OBF, "the BatchNorm-backward partials are those of the stored (bf16) gradient");
    using G = HaloGeom<PW, WN, MODE>;
    constexpr int RING = G::RING, AHEAD = RING - 1, KC = G::KC, BPW = G::BPW;
    constexpr int PH = G::PH, PITCH = G::PITCH, HROWS = G::HROWS, ROWB = G::ROWB, PER_WAVE = G::PER_WAVE;
    constexpr bool DBUF = G::HALO_BUFS == 2, FLAT = G::FLAT, PRE = G::PRE;
    constexpr int NQ = FLAT ? G::FLAT_MAXQ : PER_WAVE;         // halo pieces per wave and chunk (flat tiles: at most)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Hs = smem;                                     // [HALO_BUFS][HALO_BYTES]
    const int hbytes = FLAT ? a.halo_bytes : G::HALO_BYTES;    // one halo buffer
    char* const Bs = smem + G::HALO_BUFS * hbytes;             // [RING][B_BYTES]

    const int t = threadIdx.x;
    HALO_STAMP(0);
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int logical = xcd_remap(blockIdx.x, a.patches * a.ntiles);
    const int nt = logical % a.ntiles, patch = logical / a.ntiles;
    const int per_img = FLAT ? 1 : a.pyt * a.pxt;
    const int img = patch / per_img, prem = patch - img * per_img;
    const int y0 = FLAT ? 0 : (prem / a.pxt) * PH, x0 = FLAT ? 0 : (prem % a.pxt) * PW;
    const int n0 = nt * G::BN;
    // flat tile: pixels p0 .. p0 + 255 of the (n, y, x) order.  Virtual row of pixel (n, y, .) = n (H + 1) + y + 1 (the rows
    // that are multiples of H + 1 are the zero rows between images); with a zero column at both ends of a row, position
    // (v, hx) of that padded space, hx = x + 1, is halo pixel (v - vbase) (W + 2) + hx - hx0: the image starts at tap (0, 0) of
    // pixel p0 -- row vbase = v(p0) - 1, column hx0 = x(p0) -- and ends at tap (2, 2) of the tile's last pixel.
    const int p0 = patch * 256, w2 = a.W + 2;
    int vbase = 0, hx0 = 0, nhalo = 0, npieces = 0;
    if constexpr (FLAT) {
        const int gr0 = fdiv(p0, a.inv_w), pl = min(p0 + 255, a.P - 1), gr1 = fdiv(pl, a.inv_w);
        vbase = gr0 + fdiv(gr0, a.inv_h);
        hx0 = p0 - gr0 * a.W;
        nhalo = (gr1 + fdiv(gr1, a.inv_h) + 2 - vbase) * w2 + (pl - gr1 * a.W) + 2 - hx0 + 1;
        npieces = (nhalo * ROWB + 1023) >> 10;
    }

    // ---- loop-invariant lane offsets of this wave's halo pieces and filter pieces ----------------------------
    unsigned hvoff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int s = (q * G::NWAVES + wave) * 64 + lane;      // 16-B slot of the halo image
        const int r = G::SWZA ? s >> 2 : FLAT ? (int)(((unsigned)s * 52429u) >> 18) : s / G::SLOTS;       // (s / 5 for s < 2^16)
        // row, 8-channel group (the last slot is the pad; MODE 5: physical slot s & 3 holds group (s & 3) ^ ((row >> 2) & 3))
        const int c = G::SWZA ? (s & 3) ^ ((r >> 2) & 3) : s - r * G::SLOTS;
        unsigned vo = 0x80000000u;
        if constexpr (FLAT) {
            const int vrel = fdiv(r + hx0, a.inv_w2), hx = r + hx0 - vrel * w2;
            const int v = vbase + vrel, im = fdiv(v, a.inv_h1), yv = v - im * (a.H + 1);
            if (c < G::CSLOTS && r < nhalo && yv != 0 && hx >= 1 && hx <= a.W && im < a.N)
                vo = (unsigned)((((im * a.H + yv - 1) * a.W + hx - 1) * a.Cin) * 2 + c * 16);
        } else if (c < G::CSLOTS && r < HROWS) {
            const int hy = r / PITCH, hx = r - hy * PITCH;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                vo = (unsigned)((((img * a.H + gy) * a.W + gx) * a.Cin) * 2 + c * 16);
        }
        hvoff[q] = vo;
    }
    unsigned bvoff[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
        // a 1-KiB piece = 8 filter rows of 128 B (KC = 64) or 16 rows of 64 B (KC = 32); the 16-B groups of a row are
        // XOR-swizzled so that the 16 lanes of a ds_read_b128 group hit 16 different slots of the 256-B bank line
        const int r = KC == 64 ? (wave * 2 + j) * 8 + (lane >> 3) : wave * 16 + (lane >> 2);   // output channel n0 + r
        const int c = KC == 64 ? (lane & 7) ^ ((r >> 1) & 7) : (lane & 3) ^ ((r >> 2) & 3);
        // 32-channel chunks read the chunk-major copy [tap][chunk][Cout][32] (conv_weights_bf16): a piece = 16 whole rows of 64 B
        bvoff[j] = KC == 64 ? (unsigned)((n0 + r) * a.Cin * 2 + c * 16) : (unsigned)((n0 + r) * 64 + c * 16);
    }
    const __amdgpu_buffer_rsrc_t xsrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)a.N * a.H * a.W * a.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.wn + (KC == 64 ? (size_t)0 : (size_t)9 * a.Cin * a.Cout * 2)), 0, (int)((size_t)9 * a.Cin * a.Cout * 2), 0x00020000);

    auto has_piece = [&](int q) {                                                              // wave-uniform
        return FLAT ? q * G::NWAVES + wave < npieces : (MODE != 2 || q * G::NWAVES + wave < G::PIECES);
    };
    auto issue_halo = [&](int buf, int chunk, int q) {
        if (has_piece(q))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                xsrd, (__attribute__((address_space(3))) void*)(Hs + buf * hbytes + (q * G::NWAVES + wave) * 1024), 16,
                (int)hvoff[q], chunk * KC * 2, 0, 0);
    };
    auto issue_b = [&](int ring, int chunk, int tap) {
        const int bsoff = KC == 64 ? ((8 - tap) * a.Cout * a.Cin + chunk * KC) * 2 : ((8 - tap) * a.nchunks + chunk) * a.Cout * 64;
#pragma unroll
        for (int j = 0; j < BPW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                wsrd, (__attribute__((address_space(3))) void*)(Bs + ring * G::B_BYTES + (wave * BPW + j) * 1024), 16,
                (int)bvoff[j], bsoff, 0, 0);
    };

    // ---- this lane's operand addresses ---------------------------------------------------------------------
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
    const int l31 = lane & 31, hi32 = lane >> 5;
    int a_lane[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if constexpr (FLAT) {       // tap (0, 0) of pixel p0 + 32 (2 wm + i) + l31 (the last tile clamps: those rows are not stored)
            const int pp = min(p0 + (2 * wm + i) * 32 + l31, a.P - 1), gr = fdiv(pp, a.inv_w);
            const int hrow = (gr + fdiv(gr, a.inv_h) - vbase) * w2 + (pp - gr * a.W) - hx0;        // halo pixel of tap (0, 0)
            a_lane[i] = G::SWZA ? hrow : hrow * ROWB + hi32 * 16;
        } else {
            int py, px;
            G::pixel(2 * wm + i, l31, py, px);
            a_lane[i] = (py * PITCH + px) * ROWB + hi32 * 16;
        }
    }
    const int pitch = FLAT ? w2 : PITCH;                        // halo rows between two image rows
    const int swz = KC == 64 ? (l31 >> 1) & 7 : (l31 >> 2) & 3;
    const int b_lane = (wn * 64 + l31) * G::BROWB;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: halo of chunk 0 and the first two filter slices --------------------------------------
#pragma unroll
    for (int q = 0; q < NQ; ++q) issue_halo(0, 0, q);
#pragma unroll
    for (int k = 0; k < AHEAD; ++k) issue_b(k, 0, k);
    __builtin_amdgcn_s_waitcnt(VMCNT(0));
    __builtin_amdgcn_s_barrier();
    HALO_STAMP(1);

    // operands of k-step s4 of a tap: A from the halo at the tap's displacement, B from the tap's ring slot
    // (taprows = the tap's displacement in halo pixels)
    auto read_ops = [&](const char* Hb, int taprows, const char* Bb, int s4, bf16x8* av, bf16x8* bv) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (G::SWZA) {
                // slot (2 s4 + hi32) ^ f: the k-step flips bit 1 of the slot, i.e. the address by 32 -- one address per (tap, m-tile)
                const int rt = a_lane[i] + taprows;
                const int ad0 = rt * 64 + ((hi32 ^ ((rt >> 2) & 3)) << 4);
                av[i] = *reinterpret_cast<const bf16x8*>(Hb + (ad0 ^ (s4 << 5)));
            } else {
                av[i] = *reinterpret_cast<const bf16x8*>(Hb + taprows * ROWB + a_lane[i] + s4 * 32);
            }
        }
        const int ob = ((2 * s4 + hi32) ^ swz) * 16;
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * G::BROWB + ob);
    };
    bf16x8 pav[2], pbv[2];                       // (PRE) k-step 0 of the next tap, read before the barrier
    if constexpr (PRE) read_ops(Hs, 0, Bs + b_lane, 0, pav, pbv);

    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        const char* Hc = Hs + (DBUF ? (chunk & 1) : 0) * hbytes;
        const bool more = chunk + 1 < a.nchunks;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // loads issued during this tap: the filter slice AHEAD taps ahead and (double-buffered halo) one piece of
            // the next chunk's halo.  Ring slot of (chunk, tap) = (9 * chunk + tap) % RING.
            const int tap2 = tap + AHEAD < 9 ? tap + AHEAD : tap + AHEAD - 9;
            const bool b_more = tap + AHEAD < 9 || more;
            const bool h_more = DBUF && more && tap < NQ && has_piece(tap);
            const int slot0 = RING == 3 ? 0 : chunk & (RING - 1);       // 9 % 3 == 0, 9 % 2 == 9 % 4 == 1
            __builtin_amdgcn_sched_barrier(0);
            if (b_more) issue_b((slot0 + tap + AHEAD) % RING, tap + AHEAD < 9 ? chunk : chunk + 1, tap2);
            if (h_more) issue_halo((chunk + 1) & 1, chunk + 1, tap);
            __builtin_amdgcn_sched_barrier(0);
            const int dh = tap / 3, dw = tap - dh * 3;
            const char* Bb = Bs + ((slot0 + tap) % RING) * G::B_BYTES + b_lane;
#pragma unroll
            for (int s4 = 0; s4 < KC / 16; ++s4) {
                bf16x8 av[2], bv[2];
                if (PRE && s4 == 0) {
                    av[0] = pav[0]; av[1] = pav[1]; bv[0] = pbv[0]; bv[1] = pbv[1];
                } else {
                    read_ops(Hc, dh * pitch + dw, Bb, s4, av, bv);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
            if constexpr (PRE) {
                // the next tap's slice has been complete since the previous barrier and the halo does not change inside a
                // chunk: its first operands are read now and arrive while this wave waits at the barrier
                if (tap < 8) {
                    const int dh1 = (tap + 1) / 3, dw1 = tap + 1 - dh1 * 3;
                    read_ops(Hc, dh1 * pitch + dw1, Bs + ((slot0 + tap + 1) % RING) * G::B_BYTES + b_lane, 0, pav, pbv);
                }
            }
            // everything issued BEFORE this tap has landed once at most this tap's own loads are outstanding;
            // the barrier then publishes it to the other waves (and retires this tap's reads of ring slot tap % 3)
            __builtin_amdgcn_sched_barrier(0);
            if (RING >= 3 && b_more && h_more)
                __builtin_amdgcn_s_waitcnt(VMCNT(BPW + 1));
            else if (RING >= 3 && b_more)
                __builtin_amdgcn_s_waitcnt(VMCNT(BPW));
            else if (h_more)
                __builtin_amdgcn_s_waitcnt(VMCNT(1));
            else
                __builtin_amdgcn_s_waitcnt(VMCNT(0));      // (2-deep ring: the slice issued in this tap is the next tap's)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (chunk == 0) HALO_STAMP(2);
        if constexpr (!DBUF) {
            // single halo buffer: every wave is past its last read of it (the tap-8 barrier); refill it for the
            // next chunk -- the co-resident block keeps the matrix cores busy meanwhile
            if (more) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) issue_halo(0, chunk + 1, q);
                __builtin_amdgcn_s_waitcnt(VMCNT(0));
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (PRE) read_ops(Hs, 0, Bs + (((chunk + 1) & (RING - 1)) % RING) * G::B_BYTES + b_lane, 0, pav, pbv);
                if (chunk == 0) HALO_STAMP(3);
            }
        }
    }
    HALO_STAMP(4);

    // ---- epilogue: 64 x 64 wave tile leaves through an LDS transpose as 16-B (fp32) / 8-B (bf16) stores ------------
    float* Es = reinterpret_cast<float*>(smem) + wave * (32 * 64);
    const int n_base = n0 + wn * 64;
    f32x4 st0 = {0.f, 0.f, 0.f, 0.f}, st1 = {0.f, 0.f, 0.f, 0.f};
    const bool srelu = STATS && a.stat_mode == 2;
    // this thread's four output channels are the same for every row it stores: bias and statistics pivot once, up
    // front (a load inside the row loop is a full L2 round trip per row -- it cannot be hoisted over the bounds test)
    const int c4 = (lane & 15) * 4;
    f32x4 bz = {0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) bz = *reinterpret_cast<const f32x4*>(a.bias + n_base + c4);
    f32x4 pvt = bz;
    if (srelu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pvt[e] = fmaxf(bz[e], 0.f);
    }
    // SM == 2: the output is dL/dy of a BatchNorm(+ReLU) whose (bf16-stored) input is bb.x: st0 += d, st1 += d * x_hat with
    // d = the stored gradient where the forward ReLU let the value through -- what bn_bwd_fast's reduce pass computes
    f32x4 bsc = {0.f, 0.f, 0.f, 0.f}, bsh = bsc, bmu = bsc, brs = bsc;
    // output pixel (index in the N x H x W order) of row `row` of m-tile `mt`; false: outside the tensor
    auto out_pixel = [&](int mt, int row, int& pix) {
        if constexpr (FLAT) {
            pix = p0 + mt * 32 + row;
            return pix < a.P;
        } else {
            int py, px;
            G::pixel(mt, row, py, px);
            const int gy = y0 + py, gx = x0 + px;
            pix = (img * a.H + gy) * a.W + gx;
            return gy < a.H && gx < a.W;
        }
    };
    __amdgpu_buffer_rsrc_t bxsrd = xsrd;
    if constexpr (SM == 2) {
        bxsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.bb.x, 0, (int)((size_t)a.N * a.H * a.W * a.Cout * 2), 0x00020000);
        bsc = *reinterpret_cast<const f32x4*>(a.bb.scale + n_base + c4);
        bsh = *reinterpret_cast<const f32x4*>(a.bb.shift + n_base + c4);
        bmu = *reinterpret_cast<const f32x4*>(a.bb.mean + n_base + c4);
        const f32x4 vv = *reinterpret_cast<const f32x4*>(a.bb.var + n_base + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) brs[e] = rsqrtf(vv[e] + a.bb.eps);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        u32x2 xr[8];
        if constexpr (SM == 2) {         // the BatchNorm input of this round's 8 rows: in flight during the LDS transpose
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                int pix;
                const bool in = out_pixel(2 * wm + i, p * 4 + (lane >> 4), pix);
                const unsigned vo = in ? (unsigned)((pix * a.Cout + n_base + c4) * 2) : 0x80000000u;
                xr[p] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(bxsrd, (int)vo, 0, 0));
            }
        }
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) Es[((r & 3) + 8 * (r >> 2) + 4 * hi32) * 64 + jn * 32 + l31] = acc[i][jn][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = p * 4 + (lane >> 4), n = n_base + c4;
            int pix;
            const bool in = out_pixel(2 * wm + i, row, pix);
            f32x4 v = *reinterpret_cast<const f32x4*>(Es + row * 64 + c4);
            if (in) {
                const size_t o = (size_t)pix * a.Cout + n;
                if constexpr (OBF) {
                    v += bz;
                    bf16x4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
                    if constexpr (STATS) {
                        f32x4 d;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float yr = (float)h[e];
                            d[e] = (srelu ? fmaxf(yr, 0.f) : yr) - pvt[e];
                        }
                        st0 += d;
                        st1 += d * d;
                    }
                    if constexpr (SM == 2) {
                        const f32x4 xv = {__uint_as_float(xr[p].x << 16), __uint_as_float(xr[p].x & 0xffff0000u),
                                          __uint_as_float(xr[p].y << 16), __uint_as_float(xr[p].y & 0xffff0000u)};
                        f32x4 d;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bool pass = a.bb.relu != 1 || fmaf(xv[e], bsc[e], bsh[e]) > 0.f;
                            d[e] = pass ? (float)h[e] : 0.f;
                        }
                        st0 += d;
                        st1 += d * ((xv - bmu) * brs);
                    }
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.y) + o) = h;
                } else {
                    if constexpr (STATS) {
                        f32x4 d = v;
                        if (srelu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) d[e] = fmaxf(v[e] + bz[e], 0.f) - pvt[e];
                        }
                        st0 += d;
                        st1 += d * d;
                    }
                    v += bz;
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + o) = v;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    HALO_STAMP(5);
    if constexpr (SM != 0) {
#pragma unroll
        for (int off = 16; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                st0[e] += __shfl_xor(st0[e], off, 64);
                st1[e] += __shfl_xor(st1[e], off, 64);
            }
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);               // [which 2][wave][64]
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[(0 * G::NWAVES + wave) * 64 + lane * 4 + e] = st0[e];
                red[(1 * G::NWAVES + wave) * 64 + lane * 4 + e] = st1[e];
            }
        }
        __syncthreads();
        if (t < 2 * G::BN) {
            const int which = t / G::BN, col = t - which * G::BN;    // column inside the block's channel tile
            const int cw = col >> 6, ch = col & 63;
            float sum = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) sum += red[(which * G::NWAVES + w2 * WN + cw) * 64 + ch];   // the 4 M waves, in order
            a.stat_part[((size_t)patch * 2 + which) * a.Cout + n0 + col] = sum;
        }
    }
    HALO_STAMP(6);
    HALO_STAMP_WHERE();
}


#ifdef L3_EXPERIMENTS
// ---- 64-output-channel blocks with the filter in REGISTERS (round 6) ---------------------------------------------------------
// The Cout = 64 launches (forward and data gradient of conv1b, data gradient of conv2a: the towers' largest tensors) are where the
// ring-fed kernel above is furthest from both of its roofs: a block lives 22 us for 4.6k matrix cycles per wave, and the stamps of
// scripts/probes/build_halo_stamps.sh show why -- a tap of 8 MFMAs (256 matrix cycles) takes ~1000 cycles, because every tap ends in
// a block-wide barrier behind the LDS-DMA of a 4-KiB filter slice, and the 4 x 4 co-resident waves of a SIMD are seldom in their tap
// phase together (entry -> halo landed 4.2 us, taps 9.5 us, tile out 6 us).  Here the filter never enters LDS:
//   wave   = 128 pixels x 32 output channels (4 m-tiles x 1 n-tile; waves 2 (M) x 2 (N)): its B operand of a k-step is ONE 16-B
//            load per lane from the chunk-major filter copy [tap][Cin/32][Cout][32] (a wave's two k-steps read 32 whole 64-B rows;
//            the two M waves of a block read the same rows, L1 hits) -- three taps of B live in registers, the tap two ahead is in flight;
//   taps   = no barrier, no ring: a wave runs its 9 taps x 2 k-steps x 4 MFMAs of a chunk against the halo at its own pace; the
//            block meets only to refill the single halo buffer between 32-channel chunks and before the tile leaves;
//   A      = the same LDS halo image as MODE 2 (80-B rows, tap = immediate displacement), four reads per k-step;
//   tile   = 32 x 32 fp32 per m-tile through LDS as 16 lines of 64 floats (accumulator register r IS line r: rows r' and r' + 4 side
//            by side, every write and read conflict-free), read back linearly: a lane owns 4 channels of pixel row 8p + (lane >> 4) +
//            4 ((lane >> 3) & 1); interior blocks (the patch inside the image) take a branch-free path -- the eight reads of a round
//            are issued together instead of one L2-latency-deep dependent chain per row.
// Same products in the same order as the ring-fed kernel: the convolution output is bit-identical, the BatchNorm partials are the
// same sums taken in another order (within fp32 rounding of each other).
template <int PW, int SM, bool OBF>
__global__ __launch_bounds__(256, 4) void conv_bf16_halo64_kernel(HaloArgs a) {
    constexpr bool STATS = SM == 1;
    static_assert(SM != 2 || OBF, "the BatchNorm-backward partials are those of the stored (bf16) gradient");
    using G = HaloGeom<PW, 1, 2>;
    constexpr int KC = G::KC, PITCH = G::PITCH, HROWS = G::HROWS, ROWB = G::ROWB, NQ = G::PER_WAVE, PH = G::PH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Hs = smem;

    const int t = threadIdx.x;
    HALO_STAMP(0);
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int logical = xcd_remap(blockIdx.x, a.patches * a.ntiles);
    const int nt = logical % a.ntiles, patch = logical / a.ntiles;
    const int per_img = a.pyt * a.pxt;
    const int img = patch / per_img, prem = patch - img * per_img;
    const int y0 = (prem / a.pxt) * PH, x0 = (prem % a.pxt) * PW;
    const int n0 = nt * 64;
    const int wm2 = wave >> 1, wn2 = wave & 1;
    const int l31 = lane & 31, hi32 = lane >> 5;

    // the lane's offsets of its halo pieces live in LDS behind the halo image (7 registers the tap loop needs; read back per refill)
    unsigned* const hv_lds = reinterpret_cast<unsigned*>(smem + G::PIECES * 1024) + t;
    unsigned hvoff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int s = (q * 4 + wave) * 64 + lane;              // 16-B slot of the halo image
        const int r = s / G::SLOTS, c = s - r * G::SLOTS;      // halo row, 8-channel group (the last slot is the pad)
        unsigned vo = 0x80000000u;
        if (c < G::CSLOTS && r < HROWS) {
            const int hy = r / PITCH, hx = r - hy * PITCH;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                vo = (unsigned)((((img * a.H + gy) * a.W + gx) * a.Cin) * 2 + c * 16);
        }
        hvoff[q] = vo;
        hv_lds[q * 256] = vo;
    }
    const __amdgpu_buffer_rsrc_t xsrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)a.N * a.H * a.W * a.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.wn + (size_t)9 * a.Cin * a.Cout * 2), 0, (int)((size_t)9 * a.Cin * a.Cout * 2), 0x00020000);
    auto issue_halo = [&](int chunk, const unsigned* hv) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (q * 4 + wave < G::PIECES)                                                            // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (__attribute__((address_space(3))) void*)(Hs + (q * 4 + wave) * 1024), 16,
                                                         (int)hv[q], chunk * KC * 2, 0, 0);
    };
    // B of (chunk, tap): rows n0 + 32 wn2 + l31 of the slice, k-step s4 = bytes (2 s4 + hi32) * 16 of the 64-B row
    const int b_voff = (n0 + 32 * wn2 + l31) * 64 + hi32 * 16;
    auto load_b = [&](int chunk, int tap, bf16x8* dst) {
        const int soff = ((8 - tap) * a.nchunks + chunk) * a.Cout * 64;
#pragma unroll
        for (int s4 = 0; s4 < 2; ++s4)
            dst[s4] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wsrd, b_voff + s4 * 32, soff, 0));
    };
    // m-tile i of this wave is (PW == 32 ? 1 : 2) halo image rows below m-tile 0: one address register, immediates for the rest
    constexpr int MT_STEP = (PW == 32 ? 1 : 2) * PITCH * ROWB;
    int a_lane0;
    {
        int py, px;
        G::pixel(4 * wm2, l31, py, px);
        a_lane0 = (py * PITCH + px) * ROWB + hi32 * 16;
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // Software pipeline over the 18 k-steps of a chunk (tap = j / 2, k-step s4 = j % 2): the four halo reads of k-step j + 1 are issued
    // BEFORE the four MFMAs of k-step j (left to itself hipcc reads, waits and multiplies m-tile by m-tile: an LDS round trip per
    // MFMA, ~700 cycles a tap for 256 matrix cycles); the filter rows of tap t + 1 are requested when tap t starts.  Two register
    // sets for B alternate with the tap; 9 is odd, so the loop body covers two chunks (Cin % 64 == 0: their number is even).
    bf16x8 breg[2][2], av[2][4];
    auto read_a = [&](int j, bf16x8* dst) {
        const int tap = j >> 1, s4 = j & 1, dh = tap / 3, dw = tap - dh * 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(Hs + a_lane0 + ((dh * PITCH + dw) * ROWB + i * MT_STEP + s4 * 32));
    };
    issue_halo(0, hvoff);
    load_b(0, 0, breg[0]);
    __builtin_amdgcn_s_waitcnt(VMCNT(0));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    HALO_STAMP(1);

    for (int chunk0 = 0; chunk0 < a.nchunks; chunk0 += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int chunk = chunk0 + half;
            const bool more = chunk + 1 < a.nchunks;
            read_a(0, av[0]);
#pragma unroll
            for (int j = 0; j < 18; ++j) {
                const int tap = j >> 1, gt = 9 * half + tap;             // gt & 1: this tap's B set
                if ((j & 1) == 0) {
                    if (tap + 1 < 9) load_b(chunk, tap + 1, breg[(gt + 1) & 1]);
                    else if (more) load_b(chunk + 1, 0, breg[(gt + 1) & 1]);
                }
                if (j + 1 < 18) read_a(j + 1, av[(j + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[j & 1][i], breg[gt & 1][j & 1], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (chunk == 0) HALO_STAMP(2);
            if (more) {
                // every wave is past its last read of the halo: refill it with the next 32 channels (the co-resident blocks compute meanwhile)
                __builtin_amdgcn_s_barrier();
                unsigned hv[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) hv[q] = hv_lds[q * 256];
                issue_halo(chunk + 1, hv);
                __builtin_amdgcn_s_waitcnt(VMCNT(0));
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (chunk == 0) HALO_STAMP(3);
            }
        }
    }
    HALO_STAMP(4);

    // ---- the tile leaves: two m-tiles per round through this wave's 8 KiB of LDS (which overlay the halo: the block meets first) ----
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    float* Es = reinterpret_cast<float*>(smem) + wave * 2048;
    const int n_base = n0 + 32 * wn2;
    const int c4 = (lane & 7) * 4, rowl = (lane >> 4) + 4 * ((lane >> 3) & 1);
    f32x4 st0 = {0.f, 0.f, 0.f, 0.f}, st1 = {0.f, 0.f, 0.f, 0.f};
    const bool srelu = STATS && a.stat_mode == 2;
    const float lo = srelu ? 0.f : -__builtin_inff();         // moments of relu(y): one v_max either way
    f32x4 bz = {0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) bz = *reinterpret_cast<const f32x4*>(a.bias + n_base + c4);
    f32x4 pvt;
#pragma unroll
    for (int e = 0; e < 4; ++e) pvt[e] = fmaxf(bz[e], lo);
    f32x4 bsc = {0.f, 0.f, 0.f, 0.f}, bsh = bsc, bmr = bsc, brs = bsc;
    __amdgpu_buffer_rsrc_t bxsrd = xsrd;
    if constexpr (SM == 2) {
        bxsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.bb.x, 0, (int)((size_t)a.N * a.H * a.W * a.Cout * 2), 0x00020000);
        bsc = *reinterpret_cast<const f32x4*>(a.bb.scale + n_base + c4);
        bsh = *reinterpret_cast<const f32x4*>(a.bb.shift + n_base + c4);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(a.bb.mean + n_base + c4);
        const f32x4 vv = *reinterpret_cast<const f32x4*>(a.bb.var + n_base + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            brs[e] = rsqrtf(vv[e] + a.bb.eps);
            bmr[e] = mu[e];
        }
    }
    const __amdgpu_buffer_rsrc_t ysrd =
        __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((size_t)a.N * a.H * a.W * a.Cout * (OBF ? 2 : 4)), 0x00020000);
    // byte offset of this lane's 4 channels of pixel row `row` of m-tile `mt` (element size esz), out-of-range when the pixel is outside the image
    auto out_off = [&](int mt, int row, int esz) {
        int py, px;
        G::pixel(mt, row, py, px);
        const int gy = y0 + py, gx = x0 + px;
        const unsigned o = (unsigned)((((img * a.H + gy) * a.W + gx) * a.Cout + n_base + c4) * esz);
        return (gy < a.H) & (gx < a.W) ? o : 0x80000000u;
    };
    // max without the canonicalising v_max hipcc puts in front of fmaxf on a value made by bit operations
    auto vmax = [](float x, float m) {
        float r;
        asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m));
        return r;
    };
    // MASK = false: the patch lies inside the image (block-uniform): no lane is masked anywhere below
    auto tile_out = [&](auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        // m-tiles per round: two (eight reads in flight) -- one where the BatchNorm-backward partials' operands take the registers
        constexpr int RT = 2, NI = RT * 4;
#pragma unroll
        for (int h = 0; h < 4 / RT; ++h) {
            u32x2 xr[NI];
            unsigned off[NI];
#pragma unroll
            for (int k = 0; k < RT; ++k)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    off[k * 4 + p] = out_off(4 * wm2 + RT * h + k, 8 * p + rowl, 2);
                    if constexpr (SM == 2)       // the BatchNorm input of this round's rows: in flight during the LDS transpose
                        xr[k * 4 + p] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(bxsrd, (int)off[k * 4 + p], 0, 0));
                }
#pragma unroll
            for (int k = 0; k < RT; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) Es[k * 1024 + r * 64 + hi32 * 32 + l31] = acc[RT * h + k][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
            f32x4 v[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) v[j] = *reinterpret_cast<const f32x4*>(Es + (j >> 2) * 1024 + (j & 3) * 256 + lane * 4);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const bool in = !MASK || off[j] != 0x80000000u;
                if constexpr (OBF) {
                    const f32x4 vb = v[j] + bz;
                    bf16x4 hb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hb[e] = (__bf16)vb[e];
                    const u32x2 hw = __builtin_bit_cast(u32x2, hb);
                    const f32x4 yr = {__uint_as_float(hw.x << 16), __uint_as_float(hw.x & 0xffff0000u), __uint_as_float(hw.y << 16),
                                      __uint_as_float(hw.y & 0xffff0000u)};
                    if constexpr (STATS) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float d0 = vmax(yr[e], lo) - pvt[e];
                            const float d = in ? d0 : 0.f;
                            st0[e] += d;
                            st1[e] = fmaf(d, d, st1[e]);
                        }
                    }
                    if constexpr (SM == 2) {
                        const f32x4 xv = {__uint_as_float(xr[j].x << 16), __uint_as_float(xr[j].x & 0xffff0000u), __uint_as_float(xr[j].y << 16),
                                          __uint_as_float(xr[j].y & 0xffff0000u)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bool pass = in & (a.bb.relu != 1 || fmaf(xv[e], bsc[e], bsh[e]) > 0.f);
                            const float d = pass ? yr[e] : 0.f;
                            st0[e] += d;
                            st1[e] = fmaf(d, (xv[e] - bmr[e]) * brs[e], st1[e]);
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(hw, ysrd, (int)off[j], 0, 0);
                } else {
                    if constexpr (STATS) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float d0 = srelu ? fmaxf(v[j][e] + bz[e], 0.f) - pvt[e] : v[j][e];
                            const float d = in ? d0 : 0.f;
                            st0[e] += d;
                            st1[e] = fmaf(d, d, st1[e]);
                        }
                    }
                    const f32x4 vb = v[j] + bz;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vb), ysrd, in ? (int)(off[j] * 2) : (int)0x80000000u, 0, 0);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    };
    // (SM == 2: one path -- two copies of the larger epilogue spill)
    if (SM != 2 && y0 + PH <= a.H && x0 + PW <= a.W) tile_out(std::false_type{}); else tile_out(std::true_type{});
    HALO_STAMP(5);
    if constexpr (SM != 0) {
        // lanes with equal (lane & 7) hold the same 4 channels: fold them, then the block's two M waves per channel half, in order
#pragma unroll
        for (int off2 = 8; off2 < 64; off2 <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                st0[e] += __shfl_xor(st0[e], off2, 64);
                st1[e] += __shfl_xor(st1[e], off2, 64);
            }
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);               // [which 2][wave][32]
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[(0 * 4 + wave) * 32 + lane * 4 + e] = st0[e];
                red[(1 * 4 + wave) * 32 + lane * 4 + e] = st1[e];
            }
        }
        __syncthreads();
        if (t < 128) {
            const int which = t >> 6, col = t & 63, cw = col >> 5, ch = col & 31;
            const float sum = red[(which * 4 + cw) * 32 + ch] + red[(which * 4 + 2 + cw) * 32 + ch];
            a.stat_part[((size_t)patch * 2 + which) * a.Cout + n0 + col] = sum;
        }
    }
    HALO_STAMP(6);
    HALO_STAMP_WHERE();
}

template <int PW, int SM, bool OBF>
void launch_halo64(const HaloArgs& a_, hipStream_t s) {
    using G = HaloGeom<PW, 1, 2>;
    HaloArgs a = a_;
    a.nchunks = a.Cin / G::KC;
    // the halo image + 7 KiB of piece offsets (G::PER_WAVE per thread), or the 32 KiB of the tile's way out
    constexpr int lds = G::PIECES * 1024 + G::PER_WAVE * 1024 > 4 * 8192 ? G::PIECES * 1024 + G::PER_WAVE * 1024 : 4 * 8192;
    hipLaunchKernelGGL((conv_bf16_halo64_kernel<PW, SM, OBF>), dim3(a.patches * a.ntiles), dim3(256), lds, s, a);
}

#endif  // L3_EXPERIMENTS

template <int PW, int WN, int SM, bool OBF, int MODE>
void launch_halo3(const HaloArgs& a_, hipStream_t s) {
    using G = HaloGeom<PW, WN, MODE>;
    HaloArgs a = a_;
    a.nchunks = a.Cin / G::KC;
    static std::once_flag once[L3_MAX_DEVICES];
    int dev = 0;
    (void)hipGetDevice(&dev);
    constexpr int lds_max = G::FLAT ? 80 * 1024 : G::LDS_BYTES;
    std::call_once(once[dev & (L3_MAX_DEVICES - 1)], [] {
        (void)hipFuncSetAttribute((const void*)conv_bf16_halo_kernel<PW, WN, SM, OBF, MODE>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    });
    // (flat tiles: at least the 8 KiB per wave the epilogue's transpose takes)
    const int lds = G::FLAT ? std::max(G::HALO_BUFS * a.halo_bytes + G::RING * G::B_BYTES, G::NWAVES * 32 * 64 * 4) : G::LDS_BYTES;
    // L3_HALO_LDS_MIN (debug knob, KiB): ask for at least this much LDS per block -- fewer blocks per CU, i.e. registers and wave
    // slots left for the other tower's elementwise kernels (co-residency experiment, profiles/r05_bf16_conv_notes.txt)
    static const int lds_min = l3_knob("L3_HALO_LDS_MIN") ? atoi(l3_knob("L3_HALO_LDS_MIN")) * 1024 : 0;
    const int lds_req = lds_min > lds && lds_min <= 160 * 1024 ? lds_min : lds;
    if (lds_req > lds_max) (void)hipFuncSetAttribute((const void*)conv_bf16_halo_kernel<PW, WN, SM, OBF, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_req);
    hipLaunchKernelGGL((conv_bf16_halo_kernel<PW, WN, SM, OBF, MODE>), dim3(a.patches * a.ntiles), dim3(256 * WN), lds_req, s, a);
}

template <int PW, int WN, int SM, bool OBF>
void launch_halo2(const HaloArgs& a, hipStream_t s) {
    // default: 64-channel blocks in MODE 2 (four per CU), 128-channel blocks in MODE 1 (measured 5 % faster than their
    // MODE 2 form: half as many barriers per MFMA)
    static const int mode = l3_knob("L3_HALO_MODE") ? atoi(l3_knob("L3_HALO_MODE")) : (WN == 1 ? 2 : 1);
#ifdef L3_EXPERIMENTS
    if constexpr (WN == 1) {
        // 64-channel blocks with the filter in registers: measured slower than the ring-fed form (profiles/r06_halo64_regfilter.txt);
        // L3_HALO_BREG=1 (read per call) selects it in an L3_BUILD_EXPERIMENTS=1 library
        const char* breg = l3_knob("L3_HALO_BREG");
        if (breg != nullptr && atoi(breg) != 0) {
            launch_halo64<PW, SM, OBF>(a, s);
            return;
        }
    }
#endif
    if (mode == 2) {
        launch_halo3<PW, WN, SM, OBF, 2>(a, s);
        return;
    }
    if constexpr (WN == 2) {
        if (mode == 1) {
            launch_halo3<PW, WN, SM, OBF, 1>(a, s);
            return;
        }
    }
    launch_halo3<PW, WN, SM, OBF, 0>(a, s);
}

template <int PW, int WN>
void launch_halo(const HaloArgs& a, hipStream_t s, bool out_bf16) {
    if (a.stat_part != nullptr && a.bb.x != nullptr && out_bf16) {
        launch_halo2<PW, WN, 2, true>(a, s);
    } else if (a.stat_part != nullptr) {
        if (out_bf16) launch_halo2<PW, WN, 1, true>(a, s); else launch_halo2<PW, WN, 1, false>(a, s);
    } else {
        if (out_bf16) launch_halo2<PW, WN, 0, true>(a, s); else launch_halo2<PW, WN, 0, false>(a, s);
    }
}

// flat tiles (MODE 3): SM / OBF as launch_halo
template <int MODE>
void launch_flat2(const HaloArgs& a, hipStream_t s, bool out_bf16) {
    if (a.stat_part != nullptr && a.bb.x != nullptr && out_bf16) {
        launch_halo3<32, 2, 2, true, MODE>(a, s);
    } else if (a.stat_part != nullptr) {
        if (out_bf16) launch_halo3<32, 2, 1, true, MODE>(a, s); else launch_halo3<32, 2, 1, false, MODE>(a, s);
    } else {
        if (out_bf16) launch_halo3<32, 2, 0, true, MODE>(a, s); else launch_halo3<32, 2, 0, false, MODE>(a, s);
    }
}
// largest halo image of a flat tile, pixels (0: the float divisions of the prologue do not reach): a tile touches
// R = ceil((W + 255) / W) image rows at most and crosses Z <= floor((R - 1) / H) + 1 zero rows; from the first tap of the first
// pixel to the last tap of the last one that is at most 256 + 2 (R - 1) + Z (W + 2) + 2 (W + 2) + 2 pixels
int flat_halo_pixels(const ConvGeom& g, int n) {
    if ((size_t)n * g.H * g.W >= (1u << 22)) return 0;
    const int R = (g.W + 255 + g.W - 1) / g.W, Z = (R - 1) / g.H + 1;
    return 256 + 2 * (R - 1) + (Z + 2) * (g.W + 2) + 2;
}
// flat-tile form for this geometry: 4 (one halo buffer of 80-B rows, 4-deep ring, operands read across the barrier) where it fits
// 80 KiB, else 0; L3_HALO_FLAT_MODE forces 3 / 4 / 5 (read per call; 5 = both halo buffers, 64-B rows, 3-deep ring)
int flat_mode(const ConvGeom& g, int n) {
    const int px = flat_halo_pixels(g, n);
    if (px == 0) return 0;
    const char* env = l3_knob("L3_HALO_FLAT_MODE");
    const int force = env != nullptr ? atoi(env) : 0;
    auto fits = [&](int mode) {
        const int rowb = mode == 5 ? 64 : 80, bufs = mode == 5 ? 2 : 1, ring = (mode == 4 ? 4 : 3) * 8192;
        const int bytes = (px * rowb + 1023) / 1024 * 1024;
        return bytes <= HaloGeom<32, 2, 3>::FLAT_MAXQ * 8 * 1024 && bufs * bytes + ring <= 80 * 1024 ? bytes : 0;
    };
#ifndef L3_EXPERIMENTS
    if (force == 5) return fits(4) ? 4 : 0;     // MODE 5 is compiled only into an L3_BUILD_EXPERIMENTS=1 library
#endif
    if (force >= 3 && force <= 5) return fits(force) ? force : 0;
    return fits(4) ? 4 : 0;        // MODE 5 is the measured answer, not the default: 1-4 % slower than MODE 4 on every layer it fits
}
int flat_halo_bytes_of(const ConvGeom& g, int n, int mode) {
    const int px = flat_halo_pixels(g, n);
    return (px * (mode == 5 ? 64 : 80) + 1023) / 1024 * 1024;
}
void launch_flat(const HaloArgs& a, int mode, hipStream_t s, bool out_bf16) {
#ifdef L3_EXPERIMENTS
    if (mode == 5) return launch_flat2<5>(a, s, out_bf16);
#endif
    if (mode == 3) launch_flat2<3>(a, s, out_bf16); else launch_flat2<4>(a, s, out_bf16);
}

// patch width with the least padded area (ties: 32)
int halo_pw(const ConvGeom& g) {
    const char* fenv = l3_knob("L3_HALO_PW");          // read per call: the tests switch it inside one process
    const int force = fenv ? atoi(fenv) : 0;
    if (force == 16 || force == 32) return force;
    auto padded = [&](int pw) {
        const int ph = 256 / pw;
        return (size_t)((g.H + ph - 1) / ph * ph) * ((g.W + pw - 1) / pw * pw);
    };
    return padded(16) < padded(32) ? 16 : 32;
}

// Flat tiles where a 2-D patch would pad the image by more than 4 % and the halo of 256 consecutive pixels fits (flat_mode):
// returns the flat-tile form, 0 = 2-D patches.  L3_HALO_FLAT = 0 never, 2 wherever it fits (read per call: the tests switch it).
int halo_flat_mode(const ConvGeom& g, int n) {
    const char* env = l3_knob("L3_HALO_FLAT");
    if (env != nullptr && atoi(env) == 0) return 0;
    const char* wide = l3_knob("L3_HALO_WIDE");
    if (g.Cout % 128 != 0 || (wide != nullptr && atoi(wide) == 0)) return 0;
    const int pw = halo_pw(g), ph = 256 / pw;
    const size_t padded = (size_t)((g.H + ph - 1) / ph * ph) * ((g.W + pw - 1) / pw * pw);
    if (!(env != nullptr && atoi(env) == 2) && padded * 100 <= (size_t)g.H * g.W * 104) return 0;
    return flat_mode(g, n);
}

}  // namespace

bool conv_bf16_halo_ok(const ConvGeom& g) {
    const char* env = l3_knob("L3_BF16_HALO");           // read per call: the tests switch it inside one process
    return (env ? atoi(env) : 1) && conv_bf16_ok(g);
}

int conv_bf16_halo_patches(const ConvGeom& g, int n) {
    if (halo_flat_mode(g, n) != 0) return (int)(((size_t)n * g.H * g.W + 255) / 256);
    const int pw = halo_pw(g), ph = 256 / pw;
    return n * ((g.H + ph - 1) / ph) * ((g.W + pw - 1) / pw);
}

void conv_bf16_halo_launch(const void* x, const void* wn, const float* bias, void* y, const ConvGeom& g, int n,
                           hipStream_t s, float* stat_part, int stat_mode, bool out_bf16, const BnBwdFuse* bn_bwd) {
    HaloArgs a;
    a.bb = bn_bwd != nullptr ? *bn_bwd : BnBwdFuse{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0};
    a.x = x; a.wn = wn; a.bias = bias; a.y = y;
    a.N = n; a.H = g.H; a.W = g.W; a.Cin = g.Cin; a.Cout = g.Cout;
    a.stat_part = stat_part;
    a.stat_mode = stat_mode;
    a.P = n * g.H * g.W;
    const int fmode = halo_flat_mode(g, n);
    a.halo_bytes = fmode != 0 ? flat_halo_bytes_of(g, n, fmode) : 0;
    a.inv_w = 1.0f / (float)g.W; a.inv_h = 1.0f / (float)g.H;
    a.inv_w2 = 1.0f / (float)(g.W + 2); a.inv_h1 = 1.0f / (float)(g.H + 1);
    if (a.halo_bytes != 0) {
        a.pyt = a.pxt = 0;
        a.patches = (a.P + 255) / 256;
        a.ntiles = g.Cout / 128;
        launch_flat(a, fmode, s, out_bf16);
        return;
    }
    const int pw = halo_pw(g), ph = 256 / pw;
    a.pyt = (g.H + ph - 1) / ph;
    a.pxt = (g.W + pw - 1) / pw;
    a.patches = n * a.pyt * a.pxt;
    const int allow_wide = l3_knob("L3_HALO_WIDE") ? atoi(l3_knob("L3_HALO_WIDE")) : 1;      // read per call, as halo_flat_mode() does
    const bool wide = allow_wide && g.Cout % 128 == 0;
    a.ntiles = g.Cout / (wide ? 128 : 64);
    a.nchunks = g.Cin / 64;
    a.stat_part = stat_part;
    a.stat_mode = stat_mode;
    if (pw == 32) {
        if (wide) launch_halo<32, 2>(a, s, out_bf16); else launch_halo<32, 1>(a, s, out_bf16);
    } else {
        if (wide) launch_halo<16, 2>(a, s, out_bf16); else launch_halo<16, 1>(a, s, out_bf16);
    }
}

}  // namespace l3

#ifdef L3_HALO_STAMPS
extern "C" int l3_dbg_halo_stamps(unsigned long long* dst, int nblocks) {
    if (nblocks > (1 << 16)) nblocks = 1 << 16;
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(l3::g_halo_stamps), (size_t)nblocks * 8 * sizeof(unsigned long long));
}
#endif
