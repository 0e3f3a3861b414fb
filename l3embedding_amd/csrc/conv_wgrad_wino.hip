// conv_wgrad_wino.hip -- fp32 weight gradient of a 3x3 'same' convolution as Winograd F(3x3, 2x2) on fp32 MFMA.
//
// dW of the Conv2D layers of l3embedding/audio_model.py:372-445, vision_model.py:126-205 (what Keras' backward pass
// computes for `Conv2D(n, (3, 3), padding='same')` under train.py:282-284) -- the transpose of conv_wino.hip's
// F(2x2, 3x3): per 2x2 tile of dY and the 4x4 input tile d around it
//
//     dW[kh][kw] = sum_{a,b} d[a+kh][b+kw] dY[a][b]  =  G^T [ (B^T d B) .* (A dY A^T) ] G
//
// with the SAME B^T as the forward transform, A = (forward A^T)^T and G^T = (forward G)^T.  The sum over tiles and
// samples happens in the transformed domain: 16 independent GEMMs dU_p[c][k] = sum_t V_p[t][c] Z_p[t][k] (M = Cin,
// N = Cout, K = every tile of the batch), 16 multiplies per (tile, c, k) instead of 36 -- 2.25x fewer MFMA
// flops than the direct kernel (conv.hip conv_wgrad9t), still plain fp32 arithmetic (measured error against
// float64: ~1.4x the direct kernel's, profiles/r02_parity_distances.txt).
//
// Mapping (gfx950):
//   block  = 64 input channels x 64 output channels of dU for all 16 positions, 16 waves, wave p = position
//            p = (xi, nu): 2 x 2 MFMA tiles of 32 x 32 (64 accumulator registers, 4 waves per SIMD); one slice of
//            the K range (split-K over `splits` blocks, partials reduced by wgw_finish_kernel);
//   stage  = 8 tiles (UR x UC of them: 1x8, 2x4 or 4x2, whichever pads the image least): the raw
//            (2UR+2) x (2UC+2) input pixels x 64 channels and the 2UR x 2UC dY pixels x 64 channels go
//            HBM -> LDS with buffer_load ... lds in their NHWC order (out-of-image = out-of-range offset =
//            zeros), double buffered;
//   k      = the tile: MFMA k-step j of lane half h is tile 4h + j.  The K index must sit in a lane's
//            registers and the M / N index (channel) across lanes: a lane owns the channel PAIR (2 l, 2 l + 1) -- element
//            i of the pair is row / column l of MFMA tile i -- and reads, per raw pixel, the pair as one
//            ds_read_b64 (lanes = consecutive 8-byte slots: conflict free, 256 B/clk; the round-2 version owned
//            channels l and l + 32 and read single floats, twice the LDS instructions at half the rate): the 4 raw
//            pixels of its position's input transform and the 4 pixels of the dY tile, combined with wave-uniform
//            +-1 / 0 factors (fma(+-1, x, y) == y +- x exactly) -- both channels of the pair per v_pk_fma_f32, 3 + 0..3
//            instructions per 4 MFMAs (round 3: 6 + 2..8 plain ones; every VALU instruction takes matrix-pipe time beside
//            the fp32 MFMA) -- and fed straight to v_mfma_f32_32x32x2_f32;
//   output = dU partials [split][pos][Cin][Cout]; wgw_finish_kernel sums the splits in order and applies
//            G^T . G per (c, k).
#include "kernels.h"
#include "device_common.h"
#include "wgw_common.h"

#include <stdlib.h>

#include <mutex>

namespace l3 {

namespace {

// One ds_read_b64 (8 bytes per lane, 256 B/clk), never half of a ds_read2_b64 (128 B/clk: MI355X_MICROARCH.md, LDS): the
// empty asm statement ends the load/store optimizer's merge region.
__device__ __forceinline__ f32x2 lds_f32x2(unsigned lds_offset) {
    const f32x2 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x2*>((uintptr_t)lds_offset);
    asm volatile("");
    return v;
}

// Packed fp32 on a channel pair, as inline assembly (the compiler splits its own v_pk_* behind an MFMA back into two plain
// instructions); c = an SGPR pair holding the coefficient twice
__device__ __forceinline__ f32x2 pk_fma(f32x2 c, f32x2 x, f32x2 y) {
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y) : "s"(c), "v"(x));
    return y;
}
// ... and the last instruction of an operand's chain with the wait states a VALU result needs before an MFMA may read it as A / B
// (two; the hazard recognizer inserts them for instructions it can see, not behind inline assembly)
__device__ __forceinline__ f32x2 pk_fma_w(f32x2 c, f32x2 x, f32x2 y) {
    asm("v_pk_fma_f32 %0, %1, %2, %0\n\ts_nop 3" : "+v"(y) : "s"(c), "v"(x));
    return y;
}
template <int UC>
__global__ __launch_bounds__(1024) void conv_wgrad_wino_kernel(WgwArgs a) {
    using G = WgwGeom<UC>;
    constexpr int UR = G::UR, XPITCH = G::XPITCH, YPITCH = G::YPITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int tiles = a.ctiles * a.ktiles;
    const int logical = xcd_remap(blockIdx.x, tiles * a.splits);         // split-major: a split's tiles share an L2
    const int sp = logical / tiles, tile = logical - sp * tiles;
    const int ct = tile / a.ktiles, kt = tile - ct * a.ktiles;
    const int c0 = ct * 64, k0 = kt * 64;
    const int u_begin = sp * a.per_split, u_end = min(a.units, u_begin + a.per_split);

    // ---- LDS-DMA pieces of this wave: piece `wave` and piece `16 + wave` ------------------------------------
    // per lane: pixel (py, px) of the strip (input strip: its origin is one pixel up-left of the unit; pieces >= XPIECES: the dY
    // strip) as a byte offset pc_voff >= 0 from that origin -- the instruction's vector offset; the unit's base is its scalar offset.
    // Whether the pixel lies inside the image depends on the unit only through its border class (first / last unit row, first /
    // last unit column: 4 x 4 classes), so the lane keeps one bit per class (1 = outside) and a stage costs it two VALU per piece
    // -- extract the class's bit, or it into bit 31 of the offset (>= num_records: the load returns zeros) -- where round 3
    // recomputed the coordinates and compared them every stage (six per piece; a VALU instruction is matrix time here).
    unsigned pc_voff[2], pc_out[2];
    bool pc_isx[2];
    const int y_last = 2 * UR * (a.uy - 1), x_last = 2 * UC * (a.ux - 1);        // first row / column of the last unit row / column
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int piece = wave + 16 * q;
        const bool isx = piece < G::XPIECES;
        const int pix = (isx ? piece : piece - G::XPIECES) * 4 + (lane >> 4);
        const int pitch = isx ? XPITCH : YPITCH;
        const int py = pix / pitch, px = pix - py * pitch;
        const int C = isx ? a.Cin : a.Cout;
        const int ry = py - (isx ? 1 : 0), rx = px - (isx ? 1 : 0);      // relative to the unit's first pixel
        pc_isx[q] = isx;
        pc_voff[q] = (unsigned)((py * a.W + px) * C * 4 + (isx ? c0 : k0) * 4 + (lane & 15) * 16);
        unsigned out = 0;
#pragma unroll
        for (int cls = 0; cls < 16; ++cls) {
            const int rc = cls >> 2, cc = cls & 3;                        // bit 0: first, bit 1: last unit row / column
            const bool bad = ((rc & 1) && ry < 0) || ((rc & 2) && y_last + ry >= a.H) || ((cc & 1) && rx < 0) ||
                             ((cc & 2) && x_last + rx >= a.W) || (isx && pix >= G::XPIX);      // (padding of the last input piece)
            out |= bad ? 1u << cls : 0u;
        }
        pc_out[q] = out;
    }
    const bool second = wave + 16 < G::PIECES;
    // the input strip's origin is one row and one pixel before the unit: the descriptor starts that much before the tensor (those
    // bytes are never read -- the pixels in front of a first unit row / column are masked out) so that every offset is >= 0
    const size_t xshift = (size_t)(a.W + 1) * a.Cin * 4;
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const char*>(a.x) - xshift), 0, (int)((size_t)a.N * a.H * a.W * a.Cin * 4 + xshift), 0x00020000);
    const __amdgpu_buffer_rsrc_t ysrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((size_t)a.N * a.H * a.W * a.Cout * 4), 0x00020000);

    // unit counters (wave-uniform): sample, unit row, unit column
    const int per_img = a.uy * a.ux;
    int un = u_begin / per_img;
    int uyi = (u_begin - un * per_img) / a.ux;
    int uxi = u_begin - un * per_img - uyi * a.ux;

    auto issue = [&](int buf) {          // loads the unit (un, uyi, uxi) and advances the counters
        const int Y0 = 2 * UR * uyi, X0 = 2 * UC * uxi;
        const int xbase = ((un * a.H + Y0) * a.W + X0) * a.Cin * 4;
        const int ybase = ((un * a.H + Y0) * a.W + X0) * a.Cout * 4;
        // (min / subtract, not comparisons: a wave-uniform boolean turned into an integer goes through the vector ALU)
        const int rcls = (1 - min(uyi, 1)) + 2 * (1 - min(a.uy - 1 - uyi, 1)), ccls = (1 - min(uxi, 1)) + 2 * (1 - min(a.ux - 1 - uxi, 1));
        const int cls = rcls * 4 + ccls;
        char* S = smem + buf * G::STAGE;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (q == 1 && !second) break;
            const unsigned vo = (__builtin_amdgcn_ubfe(pc_out[q], (unsigned)cls, 1u) << 31) | pc_voff[q];
            if (pc_isx[q])
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (__attribute__((address_space(3))) void*)(S + (wave + 16 * q) * 1024),
                                                         16, (int)vo, xbase, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd, (__attribute__((address_space(3))) void*)(S + (wave + 16 * q) * 1024),
                                                         16, (int)vo, ybase, 0, 0);
        }
        if (++uxi == a.ux) {
            uxi = 0;
            if (++uyi == a.uy) {
                uyi = 0;
                ++un;
            }
        }
    };

    // ---- this wave's position ---------------------------------------------------------------------------------
    //   V = (d[ra][ca] + sa d[rb][ca]) + sb (d[ra][cb] + sa d[rb][cb])       rows of B^T: d0-d2, d1+d2, d2-d1, d1-d3
    //   Z = w00 y[0][0] + w01 y[0][1] + w10 y[1][0] + w11 y[1][1]             rows of A:   y0,    y0+y1, y0-y1, -y1
    // waves w, w + 4, w + 8, w + 12 share a SIMD: the skew gives every SIMD one wave of each nu (and each xi) -- the positions differ
    // in how many dY pixels and input columns they read (NR, NC below), and a SIMD of four centre positions would set the pace
    const int xi = wave >> 2, nu = (wave + (wave >> 2)) & 3;
    const int ra = xi == 0 ? 0 : xi == 2 ? 2 : 1, rb = xi == 0 ? 2 : xi == 1 ? 2 : xi == 2 ? 1 : 3;
    const int ca = nu == 0 ? 0 : nu == 2 ? 2 : 1, cb = nu == 0 ? 2 : nu == 1 ? 2 : nu == 2 ? 1 : 3;
    auto sgpr = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    const float sa = sgpr(xi == 1 ? 1.f : -1.f), sb = sgpr(nu == 1 ? 1.f : -1.f);
    // Z: the rows of A touch one dY pixel per direction (y0, -y1) or two (y0 + y1, y0 - y1) -- the kernel body below is
    // instantiated per (NR, NC) = how many rows / columns, so a corner position reads one dY pixel and multiplies by its sign
    // where the centre positions read four: on average 2.25 reads and 1.25 (packed) VALU per operand instead of 4 and 4.
    //   rows touched: ry0 (and ry1), coefficients sr0 (and sr1); columns likewise
    const int nr = (xi == 0 || xi == 3) ? 1 : 2, nc = (nu == 0 || nu == 3) ? 1 : 2;
    const int ry0 = xi == 3 ? 1 : 0, cy0 = nu == 3 ? 1 : 0;
    const float sr0 = xi == 3 ? -1.f : 1.f, sr1 = xi == 2 ? -1.f : 1.f;       // (sr1 / sc1 only where two are touched)
    const float sc0 = nu == 3 ? -1.f : 1.f, sc1 = nu == 2 ? -1.f : 1.f;
    const float w00 = sgpr(sr0 * sc0), w01 = sgpr(sr0 * sc1), w10 = sgpr(sr1 * sc0), w11 = sgpr(sr1 * sc1);
    // the MFMAs accumulate V (w00 Z) -- w00 = +-1 leaves the first dY pixel without a multiply and goes onto the accumulators
    // once, at the store (exact: negation commutes with every rounding on the way)
    const f32x2 sa2 = {sa, sa}, sb2 = {sb, sb}, w01_2 = {w01 * w00, w01 * w00}, w10_2 = {w10 * w00, w10 * w00},
                w11_2 = {w11 * w00, w11 * w00};

    const int l31 = lane & 31, half = lane >> 5;
    const int ltr = G::lane_tr(half), ltc = G::lane_tc(half);
    // a lane owns the channel PAIR (2 l31, 2 l31 + 1): MFMA tile i of the pair's element i -- one 8-byte read per raw pixel
    auto xaddr = [&](int r, int c) { return ((r + 2 * ltr) * XPITCH + c + 2 * ltc) * 256 + l31 * 8; };
    auto yaddr = [&](int r, int c) { return G::XBYTES + ((r + 2 * ltr) * YPITCH + c + 2 * ltc) * 256 + l31 * 8; };
    const int x_aa = xaddr(ra, ca), x_ba = xaddr(rb, ca), x_ab = xaddr(ra, cb), x_bb = xaddr(rb, cb);
    const int y_00 = yaddr(ry0, cy0), y_01 = yaddr(ry0, 1), y_10 = yaddr(1, cy0), y_11 = yaddr(1, 1);   // [first / second row][first / second column]
    // the lane's eight LDS addresses as pointers, made opaque: the loop then reads them with immediate offsets (buffer, tile) -- left
    // to itself the compiler re-adds the (zero) LDS base to five of them in every stage, five VALU instructions of matrix time
    const unsigned lds0 = (unsigned)(uintptr_t)smem;        // (the low half of a generic LDS address is the LDS offset)
    unsigned p_aa = lds0 + x_aa, p_ba = lds0 + x_ba, p_ab = lds0 + x_ab, p_bb = lds0 + x_bb;
    unsigned q_00 = lds0 + y_00, q_01 = lds0 + y_01, q_10 = lds0 + y_10, q_11 = lds0 + y_11;
    asm volatile("" : "+v"(p_aa), "+v"(p_ba), "+v"(p_ab), "+v"(p_bb), "+v"(q_00), "+v"(q_01), "+v"(q_10), "+v"(q_11));

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

    struct Raw {
        f32x2 x[4], y[4];       // the channel pair's value at the four input / dY pixels of the position
    };
    auto run = [&](auto NRT, auto NCT) {
    constexpr int NR = decltype(NRT)::value, NC = decltype(NCT)::value;
    // (NC == 1 <=> nu = 0 or 3 <=> the position's two input columns are two apart = one tile step: the first column of the next
    //  tile in the row is the second column of this one -- two of the four input reads come from the previous step's registers)
    auto load_raw = [&](int sb, auto JT, Raw& r, const Raw& prev) {          // sb: byte offset of the stage buffer
        constexpr int j = decltype(JT)::value;
        constexpr bool share = NC == 1 && j > 0 && G::imm_tr(j) == G::imm_tr(j > 0 ? j - 1 : 0) &&
                               G::imm_tc(j) == G::imm_tc(j > 0 ? j - 1 : 0) + 1;
        const int ix = ((2 * G::imm_tr(j)) * XPITCH + 2 * G::imm_tc(j)) * 256;
        const int iy = ((2 * G::imm_tr(j)) * YPITCH + 2 * G::imm_tc(j)) * 256;
        f32x2 xa, xb;
        if constexpr (share) {
            xa = prev.x[2];
            xb = prev.x[3];
        } else {
            xa = lds_f32x2(p_aa + sb + ix);
            xb = lds_f32x2(p_ba + sb + ix);
        }
        const f32x2 xc = lds_f32x2(p_ab + sb + ix), xd = lds_f32x2(p_bb + sb + ix);
        f32x2 ya = lds_f32x2(q_00 + sb + iy), yb = ya, yc = ya, yd = ya;
        if (NC == 2) yb = lds_f32x2(q_01 + sb + iy);
        if (NR == 2) yc = lds_f32x2(q_10 + sb + iy);
        if (NR == 2 && NC == 2) yd = lds_f32x2(q_11 + sb + iy);
        r.x[0] = xa; r.x[1] = xb; r.x[2] = xc; r.x[3] = xd;
        r.y[0] = ya; r.y[1] = yb; r.y[2] = yc; r.y[3] = yd;
    };
    auto mfma_step = [&](const Raw& r) {
        // both channels of the pair per instruction (v_pk_fma_f32, the +-1 coefficient as an SGPR pair): the same fma chain per
        // element as the plain form, half the VALU instructions beside the MFMAs
        const f32x2 v = pk_fma_w(sb2, pk_fma(sa2, r.x[3], r.x[2]), pk_fma(sa2, r.x[1], r.x[0]));
        f32x2 z;
        if constexpr (NR == 1 && NC == 1) {
            z = r.y[0];                                        // straight from LDS
        } else if constexpr (NR == 1) {
            z = pk_fma_w(w01_2, r.y[1], r.y[0]);
        } else if constexpr (NC == 1) {
            z = pk_fma_w(w10_2, r.y[2], r.y[0]);
        } else {
            z = pk_fma_w(w11_2, r.y[3], pk_fma(w10_2, r.y[2], pk_fma(w01_2, r.y[1], r.y[0])));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i], z[jn], acc[i][jn], 0, 0, 0);
    };
    auto compute = [&](auto BUF) {
        constexpr int S = decltype(BUF)::value * G::STAGE;
        Raw r0, r1;
        load_raw(S, std::integral_constant<int, 0>{}, r0, r0);
        __builtin_amdgcn_sched_barrier(0);
        load_raw(S, std::integral_constant<int, 1>{}, r1, r0);
        mfma_step(r0);
        __builtin_amdgcn_sched_barrier(0);
        load_raw(S, std::integral_constant<int, 2>{}, r0, r1);
        mfma_step(r1);
        __builtin_amdgcn_sched_barrier(0);
        load_raw(S, std::integral_constant<int, 3>{}, r1, r0);
        mfma_step(r0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(r1);
    };
    auto stage_barrier = [&]() {          // behind the stage's MFMAs (see conv_wino.hip)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nunits = u_end - u_begin;
    if (nunits > 0) {
        issue(0);
        __syncthreads();
        for (int u = 0; u < nunits; u += 2) {
            if (u + 1 < nunits) issue(1);
            compute(std::integral_constant<int, 0>{});
            stage_barrier();
            if (u + 1 < nunits) {
                if (u + 2 < nunits) issue(0);
                compute(std::integral_constant<int, 1>{});
                stage_barrier();
            }
        }
    }
    };
    if (nr == 1 && nc == 1)
        run(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    else if (nr == 1)
        run(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
    else if (nc == 1)
        run(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
    else
        run(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});

    // ---- dU partial of this (split, position): rows = input channels, lanes = output channels --------------------
    // MFMA tile (i, jn), row r, lane l31 = input channel 2 row + i, output channel 2 l31 + jn: the two jn of a row are one 8-byte store
    float* out = a.part + ((size_t)(sp * 16 + xi * 4 + nu) * a.Cin + c0) * a.Cout + k0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = 2 * ((r & 3) + 8 * (r >> 2) + 4 * half) + i;
            // (written once, read once by the reduce: non-temporal, like the streams of bn_fused.hip)
            __builtin_nontemporal_store(f32x2{w00 * acc[i][0][r], w00 * acc[i][1][r]}, reinterpret_cast<f32x2*>(out + (size_t)c * a.Cout + 2 * l31));
        }
}

// dW[kh][kw][c][k] = sum_{xi,nu} GT[kh][xi] GT[kw][nu] (sum over splits, in order, of dU[split][xi*4+nu][c][k]),
// GT = [[1, 1/2, 1/2, 0], [0, 1/2, -1/2, 0], [0, 1/2, 1/2, 1]]
__global__ __launch_bounds__(256) void wgw_finish_kernel(const float* __restrict__ part, float* __restrict__ dw, int cc,
                                                         int splits) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cc) return;
    float u[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) u[p] = 0.f;
    for (int s = 0; s < splits; ++s)
#pragma unroll
        for (int p = 0; p < 16; ++p) u[p] += __builtin_nontemporal_load(part + ((size_t)s * 16 + p) * cc + idx);
    float rrow[3][4];           // G^T applied to the xi index
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
        const float h1 = 0.5f * u[4 + nu], h2 = 0.5f * u[8 + nu];
        rrow[0][nu] = u[nu] + (h1 + h2);
        rrow[1][nu] = h1 - h2;
        rrow[2][nu] = (h1 + h2) + u[12 + nu];
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const float h1 = 0.5f * rrow[kh][1], h2 = 0.5f * rrow[kh][2];
        dw[(size_t)(kh * 3 + 0) * cc + idx] = rrow[kh][0] + (h1 + h2);
        dw[(size_t)(kh * 3 + 1) * cc + idx] = h1 - h2;
        dw[(size_t)(kh * 3 + 2) * cc + idx] = (h1 + h2) + rrow[kh][3];
    }
}

struct WgwPlan {
    int uc, uy, ux, units, per_split, splits;
};

// force_uc != 0: the plan for that unit shape regardless of the L3_WGW_UC switch (scratch sizing)
WgwPlan wgw_plan(const ConvGeom& g, int n, int force_uc = 0) {
    const int TY = (g.H + 1) / 2, TX = (g.W + 1) / 2;
    WgwPlan p;
    size_t best = ~(size_t)0;
    p.uc = 8;
    for (int uc : {8, 4, 2}) {                    // least padded tile area; ties: the widest
        const int ur = 8 / uc;
        const size_t padded = (size_t)((TY + ur - 1) / ur * ur) * ((TX + uc - 1) / uc * uc);
        if (padded < best) {
            best = padded;
            p.uc = uc;
        }
    }
    const char* fenv = l3_knob("L3_WGW_UC");           // read per call: the tests switch it inside one process
    const int force = force_uc ? force_uc : fenv ? atoi(fenv) : 0;
    if (force == 8 || force == 4 || force == 2) p.uc = force;
    const int ur = 8 / p.uc;
    p.uy = (TY + ur - 1) / ur;
    p.ux = (TX + p.uc - 1) / p.uc;
    p.units = n * p.uy * p.ux;
    const int tiles = (g.Cin / 64) * (g.Cout / 64);
    static const int target = l3_knob("L3_WGW_BLOCKS") ? atoi(l3_knob("L3_WGW_BLOCKS")) : 256;
    int splits = (target + tiles - 1) / tiles;
    const int max_splits = (p.units + 31) / 32;   // >= 32 stages per block
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.per_split = (p.units + splits - 1) / splits;
    p.splits = (p.units + p.per_split - 1) / p.per_split;
    return p;
}

template <int UC>
void launch_wgw(const WgwArgs& a, hipStream_t s) {
    using G = WgwGeom<UC>;
    static std::once_flag once[L3_MAX_DEVICES];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & (L3_MAX_DEVICES - 1)], [] {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_wino_kernel<UC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)G::LDS_BYTES);
    });
    hipLaunchKernelGGL((conv_wgrad_wino_kernel<UC>), dim3(a.ctiles * a.ktiles * a.splits), dim3(1024), G::LDS_BYTES, s, a);
}

}  // namespace

bool conv_wgrad_wino_ok(const ConvGeom& g) {
    const char* env = l3_knob("L3_WG_WINO");             // read per call: the tests switch it inside one process
    return (env ? atoi(env) : 1) && g.KH == 3 && g.KW == 3 && g.padT == 1 && g.padL == 1 && g.Ho == g.H && g.Wo == g.W &&
           g.Cin % 64 == 0 && g.Cout % 64 == 0;
}

int conv_wgrad_wino_splits(const ConvGeom& g, int n) { return wgw_plan(g, n).splits; }

// The slice count a scratch buffer must hold: the unit shape is re-read per launch (the tests switch L3_WGW_UC inside
// one process, after the engine sized its scratch), so size for the largest of the three
int conv_wgrad_wino_max_splits(const ConvGeom& g, int n) {
    int m = wgw_plan(g, n).splits;
    for (int uc : {8, 4, 2}) {
        const int sp = wgw_plan(g, n, uc).splits;
        if (sp > m) m = sp;
    }
    return m;
}

double conv_wgrad_wino_executed_flops(const ConvGeom& g) {
    const WgwPlan p = wgw_plan(g, g.N);
    return 2.0 * 16.0 * (double)p.units * 8.0 * (double)g.Cin * (double)g.Cout;
}

// n samples starting at x / dy; `part` holds conv_wgrad_wino_splits(g, n) slices of 16 * Cin * Cout floats
void conv_wgrad_wino_launch(const float* x, const float* dy, float* part, const ConvGeom& g, int n, hipStream_t s) {
    const WgwPlan p = wgw_plan(g, n);
    WgwArgs a;
    a.x = x; a.dy = dy; a.part = part;
    a.N = n; a.H = g.H; a.W = g.W; a.Cin = g.Cin; a.Cout = g.Cout;
    a.uy = p.uy; a.ux = p.ux; a.units = p.units; a.per_split = p.per_split; a.splits = p.splits;
    a.ctiles = g.Cin / 64; a.ktiles = g.Cout / 64;
#ifdef L3_EXPERIMENTS
    const char* bx = l3_knob("L3_WG_BX6");               // read per call: the tests switch it inside one process
    if (bx != nullptr && atoi(bx) == 1) return conv_wgrad_bx6_launch(a, p.uc, s);      // split-bf16 operands (conv_wgrad_bx6.hip)
#endif
    if (p.uc == 8)
        launch_wgw<8>(a, s);
    else if (p.uc == 4)
        launch_wgw<4>(a, s);
    else
        launch_wgw<2>(a, s);
}

// dw (3, 3, Cin, Cout) from `splits` partial slices
void conv_wgrad_wino_finish(const float* part, float* dw, const ConvGeom& g, int splits, hipStream_t s) {
    const int cc = g.Cin * g.Cout;
    hipLaunchKernelGGL(wgw_finish_kernel, dim3((cc + 255) / 256), dim3(256), 0, s, part, dw, cc, splits);
}

}  // namespace l3
