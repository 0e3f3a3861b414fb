// conv_wgrad_bf16.hip -- mixed-precision 3x3 weight gradient on bfloat16-stored tensors.
//
//   dW[tap][ci][co] = sum over (n, y, x) of X[n, y + dy - 1, x + dx - 1, ci] * dY[n, y, x, co]
//
// (Conv2D kernel gradient of the 14 mixed-precision layers, l3embedding/audio_model.py:372-445 /
// vision_model.py:126-205, BASELINE.json configs[4]: bf16 x bf16 products, fp32 accumulate; same split-K
// partials + ordered reduction as conv.hip's conv_wgrad9t_kernel, which this replaces for stored operands.)
//
// The reduction index of this GEMM is the PIXEL, while both tensors are stored pixel-major with the channels
// contiguous -- every MFMA operand is a transposed read.  conv_wgrad9t_kernel transposes with scalar fp32
// LDS writes and converts back to bf16 in registers (36 v_cvt_pk + 18 ds_read_b64 per 9 MFMAs: VALU/LDS
// bound, 0.19 of the bf16 peak).  Here the tiles go HBM -> LDS untouched (buffer_load ... lds, pixel-major
// rows of 144 B) and gfx950's transpose read does the rest: one ds_read_b64_tr_b16 hands lane i of a 16-lane
// group the 4 pixels x 1 channel column of a [4 pixels][16 channels] block, i.e. half an MFMA operand, with
// no VALU work at all.  The four pixels of one read are taken 4 apart (x, x+4, x+8, x+12): with 144-B rows
// they start 16 banks apart, so the 32 lanes of an LDS cycle cover all 64 banks once.
//
//   block  = 64 input channels x 64 output channels x all 9 taps, 4 waves as 2 (ci) x 2 (co), 9 accumulator
//            tiles of 32 x 32 per wave; K loop over 4 x 16-pixel patches (4 k-steps of 16 pixels), the patch's
//            6 x 18 input halo and its 4 x 16 dY rows double-buffered in LDS; split-K over patches.
//   SQ (round 5): the patch as 8 x 8 pixels instead of 4 x 16 -- 56-, 24-, 112-, 224-wide images tile exactly, 49 / 99 / 199
//            pad to 56 / 104 / 200 instead of 64 / 112 / 208 (the 4 x 16 patch put 12.5 % of the matrix work of the 56-wide and
//            25 % of the 24-wide layers on padding); the launch takes whichever pads less.  A k-step is then patch rows ks and
//            ks + 4 (2 x 8 pixels): with the 10-pixel halo pitch the four pixels of a transpose read -- (ks, x), (ks, x + 4),
//            (ks + 4, x), (ks + 4, x + 4) -- sit 0 / 4 / 40 / 44 halo rows apart, i.e. 0 / 4 / 8 / 12 modulo 16: the same
//            bank picture as four pixels taken 4 apart in a row.  The dY rows are stored permuted (row of pixel (py, px) =
//            16 (py % 4) + 8 (py / 4) + px) for the same reason.  Halo 10 x 10 = 100 pixels (108 before): same LDS.
//   TS = 2 (round 5): 8 waves as 2 (ci) x 2 (co) x 2 (taps 0-4 / taps 5-8).  PMC of the 4-wave form: matrix pipe 0.59 busy -- a
//            wave spends about as long issuing its 7 LDS-DMA pieces per patch as its 36 MFMAs take, 206 registers allow two
//            waves per SIMD and only another wave's MFMAs can fill the issue time.  With the taps split a wave carries 5 (4)
//            accumulator tiles -- 128 registers, FOUR waves per SIMD at the same LDS per block -- and issues half the pieces.
#include "kernels.h"
#include "device_common.h"

#include <stdlib.h>

#include <mutex>
#include <type_traits>

namespace l3 {

namespace {

struct WgTrArgs {
    const void* x;      // bf16 NHWC
    const void* dy;     // bf16 NHWC
    float* part;        // [split][9][Cin][Cout]
    int N, H, W, Cin, Cout;
    int co_tiles, tiles;
    int pyt, pxt;       // patches per image
    int npatch, per_split, splits;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int ROWB = 144;                       // bytes per pixel row in LDS (64 channels + 16 B pad)
constexpr int HPITCH = 18;                      // halo pixels per patch row (16 + 2); 8 x 8 patches: 10
constexpr int X_PIECES = 16;                    // 6 x 18 = 108 pixel rows = 15552 B -> 16 KiB
constexpr int D_PIECES = 12;                    // 64 pixel rows = 9216 B = 9 pieces, padded to 3 per wave
constexpr int STAGE_BYTES = (X_PIECES + D_PIECES) * 1024;
constexpr int WGTR_LDS = 2 * STAGE_BYTES;       // 56 KiB: two blocks per CU

__device__ __forceinline__ bf16x8 tr_operand(const char* p0, const char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
}

template <int TS>
struct WgTrPieces {
    static constexpr int XQ = 4 / TS, DQ = TS == 1 ? 3 : 2;
};

// TS = waves per (ci, co) quarter of the block: 1 = all nine taps in one wave, 2 = taps 0-4 and 5-8 in two waves
template <int TS, bool SQ>
__global__ __launch_bounds__(256 * TS, 2) void conv_wgrad_bf16_tr_kernel(WgTrArgs a) {
    constexpr int PH = SQ ? 8 : 4, PWD = SQ ? 8 : 16, HP = PWD + 2, HROWS = (PH + 2) * HP;      // patch, halo pitch, halo pixels
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int logical = xcd_remap(blockIdx.x, a.tiles * a.splits);
    const int sp = logical / a.tiles, tile = logical - sp * a.tiles;
    const int cit = tile / a.co_tiles, cot = tile - cit * a.co_tiles;
    const int ci0 = cit * 64, co0 = cot * 64;
    const int p_begin = sp * a.per_split;
    const int p_end = min(a.npatch, p_begin + a.per_split);

    // ---- staging: 4 halo pieces + 3 dY pieces per wave and patch (slot s = 9 * pixel row + 16-B chunk; chunk 8 = pad)
    const int margin = (a.W + 1) * a.Cin * 2;             // most negative halo displacement, bytes
    // pieces per wave: X pieces wave * XQ + q; dY pieces wave * 3 + q (TS 1) or wave + 8 q.  The offset arrays below keep FIXED
    // bounds (4, 3): with bounds that depend on TS, captured by the lambdas, hipcc 7.2 silently drops the kernel's HOST stub and
    // the library fails to load with the kernel as an undefined symbol.
    using PW = WgTrPieces<TS>;
    auto dpiece = [&](int q) { return TS == 1 ? wave * 3 + q : wave + 8 * q; };
    int xhy[4], xhx[4], xvo[4];                     // (fixed bounds: see WgTrPieces)
#pragma unroll
    for (int q = 0; q < PW::XQ; ++q) {
        const int s = (wave * PW::XQ + q) * 64 + lane;
        const int r = s / 9, c = s - r * 9;
        const int hy = r / HP, hx = r - hy * HP;
        const bool slot = c < 8 && r < HROWS;
        xhy[q] = slot ? hy - 1 : 0x40000000;              // never inside the image
        xhx[q] = hx - 1;
        xvo[q] = ((hy - 1) * a.W + (hx - 1)) * a.Cin * 2 + (ci0 + c * 8) * 2 + margin;
    }
    int dqy[3], dqx[3], dvo[3];
#pragma unroll
    for (int q = 0; q < PW::DQ; ++q) {
        const int s = dpiece(q) * 64 + lane;
        const int r = s / 9, c = s - r * 9;
        const bool slot = c < 8 && r < 64;
        const int dpy = SQ ? 4 * ((r >> 3) & 1) + (r >> 4) : r >> 4, dpx = SQ ? r & 7 : r & 15;      // pixel of dY row r
        dqy[q] = slot ? dpy : 0x40000000;
        dqx[q] = dpx;
        dvo[q] = (dpy * a.W + dpx) * a.Cout * 2 + (co0 + c * 8) * 2;
    }
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.x - margin), 0, (int)((size_t)a.N * a.H * a.W * a.Cin * 2 + margin), 0x00020000);
    const __amdgpu_buffer_rsrc_t dsrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((size_t)a.N * a.H * a.W * a.Cout * 2), 0x00020000);
    const int per_img = a.pyt * a.pxt;

    auto issue = [&](int buf, int pidx) {
        const int n = pidx / per_img, rem = pidx - n * per_img;
        const int py = rem / a.pxt, px = rem - py * a.pxt;
        const int y0 = py * PH, x0 = px * PWD;
        const int pix0 = (n * a.H + y0) * a.W + x0;
        char* Xs = smem + buf * STAGE_BYTES;
        char* Dsm = Xs + X_PIECES * 1024;
#pragma unroll
        for (int q = 0; q < PW::XQ; ++q) {
            const bool ok = (unsigned)(y0 + xhy[q]) < (unsigned)a.H && (unsigned)(x0 + xhx[q]) < (unsigned)a.W;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (__attribute__((address_space(3))) void*)(Xs + (wave * PW::XQ + q) * 1024),
                                                     16, ok ? xvo[q] : (int)0x80000000, pix0 * a.Cin * 2, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < PW::DQ; ++q) {
            if (dpiece(q) >= D_PIECES) continue;         // (wave-uniform)
            const bool ok = (unsigned)(y0 + dqy[q]) < (unsigned)a.H && x0 + dqx[q] < a.W;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dsrd, (__attribute__((address_space(3))) void*)(Dsm + dpiece(q) * 1024),
                                                     16, ok ? dvo[q] : (int)0x80000000, pix0 * a.Cout * 2, 0, 0);
        }
    };

    // ---- operand addresses: lane i of 16-lane group g4 points at pixel 4 (i >> 2) + 2 khalf (+ j) of its k-step row,
    //      channel quad (i & 3) of the group's 16 channels, and receives channel i of 4 pixels taken 4 apart.
    //      k index 8 khalf + 4 j + t  <->  pixel 4 t + 2 khalf + j of the row: the same map on both operands.
    const int wq = wave & 3, wk = wq >> 1, wn = wq & 1;       // (ci, co) quarter; wave >> 2 = tap half (TS 2)
    const int i16 = lane & 15, g4 = lane >> 4, khalf = g4 >> 1;
    // 4 x 16 patch: pixel 4 t + 2 khalf of the k-step's row.  8 x 8: t = 0, 1 -> patch row ks, columns 2 khalf + 4 t; t = 2, 3 -> row ks + 4
    const int tq = i16 >> 2;
    const int a_pix = SQ ? 4 * (tq >> 1) * HP + 4 * (tq & 1) + 2 * khalf : 4 * tq + 2 * khalf;
    const int d_pix = SQ ? 8 * (tq >> 1) + 4 * (tq & 1) + 2 * khalf : 4 * tq + 2 * khalf;
    const int a_lane = a_pix * ROWB + (wk * 32 + (g4 & 1) * 16 + (i16 & 3) * 4) * 2;
    const int d_lane = d_pix * ROWB + (wn * 32 + (g4 & 1) * 16 + (i16 & 3) * 4) * 2;

    // the K loop and the output of this wave: taps T0 .. T0 + NT - 1
    auto taps = [&](auto t0_, auto nt_) {
        constexpr int T0 = decltype(t0_)::value, NT = decltype(nt_)::value;
        f32x16 acc[NT];
    #pragma unroll
        for (int i = 0; i < NT; ++i)
    #pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

        if (p_begin < p_end) issue(0, p_begin);
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        for (int pi = p_begin; pi < p_end; ++pi) {
            const int buf = (pi - p_begin) & 1;
            __builtin_amdgcn_sched_barrier(0);
            if (pi + 1 < p_end) issue(buf ^ 1, pi + 1);
            __builtin_amdgcn_sched_barrier(0);
            const char* Xs = smem + buf * STAGE_BYTES + a_lane;
            const char* Dsm = smem + buf * STAGE_BYTES + X_PIECES * 1024 + d_lane;
    #pragma unroll
            for (int ks = 0; ks < 4; ++ks) {                  // k-step = patch row ks (16 pixels)
                const bf16x8 bv = tr_operand(Dsm + ks * 16 * ROWB, Dsm + ks * 16 * ROWB + ROWB);
    #pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int tap = T0 + i, dh = tap / 3, dw = tap - dh * 3;
                    const char* Ap = Xs + ((ks + dh) * HP + dw) * ROWB;
                    const bf16x8 av = tr_operand(Ap, Ap + ROWB);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0F70);              // the next patch has landed ...
            __builtin_amdgcn_s_barrier();                    // ... for every wave, and this one's reads are retired
            __builtin_amdgcn_sched_barrier(0);
        }

        float* out = a.part + (size_t)sp * 9 * a.Cin * a.Cout;
        const int l31 = lane & 31, hi32 = lane >> 5;
        const int n = co0 + wn * 32 + l31;
    #pragma unroll
        for (int i = 0; i < NT; ++i)
    #pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = (T0 + i) * a.Cin + ci0 + wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi32;
                out[(size_t)k * a.Cout + n] = acc[i][r];
            }
    };
    using std::integral_constant;
    if constexpr (TS == 1) {
        taps(integral_constant<int, 0>{}, integral_constant<int, 9>{});
    } else {
        // both halves run the same sequence of barriers
        if (wave < 4)
            taps(integral_constant<int, 0>{}, integral_constant<int, 5>{});
        else
            taps(integral_constant<int, 5>{}, integral_constant<int, 4>{});
    }
}

}  // namespace

bool conv_wgrad_bf16_tr_enabled() {
    const char* env = l3_knob("L3_WG_TR");                 // read per call: the tests switch it inside one process
    return env ? atoi(env) != 0 : true;
}

// `n` samples of geometry g (x / dy already offset to the first one), `splits` split-K slices of `part`
void conv_wgrad_bf16_tr_launch(const void* x, const void* dy, float* part, const ConvGeom& g, int n, int splits, hipStream_t s) {
    WgTrArgs a;
    a.x = x; a.dy = dy; a.part = part;
    a.N = n; a.H = g.H; a.W = g.W; a.Cin = g.Cin; a.Cout = g.Cout;
    a.co_tiles = g.Cout / 64;
    a.tiles = (g.Cin / 64) * a.co_tiles;
    // 8 x 8 patches where they pad the image less than 4 x 16 ones
    const size_t pad4 = (size_t)((g.H + 3) / 4 * 4) * ((g.W + 15) / 16 * 16), pad8 = (size_t)((g.H + 7) / 8 * 8) * ((g.W + 7) / 8 * 8);
    const char* sqenv = l3_knob("L3_WG_TR_SQ");            // read per call: 0 / 1 force a shape
    const bool sq = sqenv != nullptr ? atoi(sqenv) != 0 : pad8 < pad4;
    a.pyt = sq ? (g.H + 7) / 8 : (g.H + 3) / 4;
    a.pxt = sq ? (g.W + 7) / 8 : (g.W + 15) / 16;
    a.npatch = n * a.pyt * a.pxt;
    a.splits = splits;
    a.per_split = (a.npatch + splits - 1) / splits;
    static std::once_flag once[L3_MAX_DEVICES];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & (L3_MAX_DEVICES - 1)], [] {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_tr_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, WGTR_LDS);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_tr_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WGTR_LDS);
#ifdef L3_EXPERIMENTS
        (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_tr_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, WGTR_LDS);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_tr_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WGTR_LDS);
#endif
    });
    // The tap-split form is the measured answer, not the product path: alone it is 5.6 % faster (4054 against 4296 us over the 14
    // layers, scripts/probes/halo_bench), in the training step it is not (serialised 24.01 against 24.05 ms; with the two towers in
    // flight 23.46 against 23.27 ms: sixteen 128-register waves fill the register file and the other tower's BatchNorm kernels no
    // longer share the CU).  Compiled only into an L3_BUILD_EXPERIMENTS=1 library, where L3_WG_TR_TS=2 (debug knob, read per call) selects it.
    using Fn = void (*)(WgTrArgs);
#ifdef L3_EXPERIMENTS
    const char* env = l3_knob("L3_WG_TR_TS");
    if (env != nullptr && atoi(env) == 2) {
        static const Fn fns2[2] = {conv_wgrad_bf16_tr_kernel<2, false>, conv_wgrad_bf16_tr_kernel<2, true>};
        hipLaunchKernelGGL(fns2[sq ? 1 : 0], dim3(a.tiles * splits), dim3(512), WGTR_LDS, s, a);
        return;
    }
#endif
    static const Fn fns[2] = {conv_wgrad_bf16_tr_kernel<1, false>, conv_wgrad_bf16_tr_kernel<1, true>};
    hipLaunchKernelGGL(fns[sq ? 1 : 0], dim3(a.tiles * splits), dim3(256), WGTR_LDS, s, a);
}

}  // namespace l3
