// conv_wino.hip -- 3x3 / stride 1 / pad 1 convolution as Winograd F(2x2, 3x3) on fp32 MFMA.
//
// Replaces the Conv2D forward and data-gradient launches of the VGG blocks
// (l3embedding/audio_model.py:372-445, vision_model.py:126-205: every `Conv2D(n, (3, 3),
// padding='same')` except the first of each tower) -- 2.25x fewer multiplies than the direct
// implicit GEMM of conv.hip, still plain fp32 arithmetic:
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        per 2x2 output tile, summed over channels
//
// so the channel sum becomes 16 independent GEMMs (one per position of the 4x4 transformed
// tile): M_p[tile][k] = sum_c V_p[tile][c] * U_p[c][k].
//
// Mapping (gfx950):
//   block  = 64 tiles (BTY tile rows x BTX tile columns) x 64 output channels, 16 waves;
//   wave p = position p = (xi, nu) of the 4x4 transformed tile, all 64 tiles x 64 channels of it
//            (2x2 MFMA tiles of 32x32, 64 accumulator registers -> 4 waves per SIMD, so LDS
//            latency, the input transform's VALU work and the barrier hide behind other waves);
//   k loop = 8 input channels per stage: the raw 4-row input strips of the tile rows and the
//            16 x 8 x 64 slice of U go HBM -> LDS with buffer_load ... lds (scalar soffset per
//            stage, loop-invariant lane offsets, out-of-range == zero padding), double buffered;
//            a lane reads the 2x2 raw pixels its position needs (row pair of xi, column pair of
//            nu; 4 channels = the k-half of its MFMA lane), combines them with two wave-uniform
//            signs -- V = (d[ra][ca] + sa d[rb][ca]) + sb (d[ra][cb] + sa d[rb][cb]) -- and feeds
//            the result straight into the A operand of v_mfma_f32_32x32x2_f32: V never touches LDS;
//   output = the 16 positions of a (tile, channel) meet through LDS (the stage buffers, one
//            32x32 quarter at a time) where one thread applies A^T M A, adds the bias and stores.
//   LDS A image: [k-half][tile row][input row 0..3][pixel parity][pixel/2] x 16 B, so that the
//            lanes of one ds_read_b128 (consecutive tiles) read consecutive 16-B slots.
//   U layout in HBM: [pos 16][Cin/4][Cout][4] (conv_wino_transform_weights), one 1-KiB piece
//            = 64 output channels x 4 input channels of one position.
#include "kernels.h"
#include "device_common.h"
#include "wino_common.h"

#include <stdlib.h>

#include <atomic>
#include <mutex>

namespace l3 {

// conv_wino4.hip: Winograd F(4x4, 3x3) for the layers with many input channels
bool conv_wino4_selected(const ConvGeom& g);
double conv_wino4_executed_flops(const ConvGeom& g);
int conv_wino4_blocks(const ConvGeom& g, int n);
void conv_wino4_transform_weights(const float* w, float* u, const ConvGeom& g, bool from_fwd_for_dgrad, hipStream_t s);
void conv_wino4_launch(const float* x, const float* u, const float* bias, float* y, const ConvGeom& g, int n, hipStream_t s,
                       float* stat_part, int stat_mode, const BnBwdFuse* bn_bwd);

// conv_wino_bx6.hip: F(2x2, 3x3) with split-bf16 operands on the bf16 matrix pipe (g.f2x2 == 2) -- a measured experiment that loses to
// F(4x4,3x3) in the step (profiles/r05_bx6_ablations.txt); compiled only into an L3_BUILD_EXPERIMENTS=1 library (_build.py)
#ifdef L3_EXPERIMENTS
bool conv_wino_bx6_ok(const ConvGeom& g);
int conv_wino_bx6_blocks(const ConvGeom& g, int n);
double conv_wino_bx6_executed_flops(const ConvGeom& g);
void conv_wino_bx6_transform_weights(const float* w, float* u, const ConvGeom& g, bool from_fwd_for_dgrad, hipStream_t s);
void conv_wino_bx6_launch(const float* x, const float* u, const float* bias, float* y, const ConvGeom& g, int n, hipStream_t s,
                          float* stat_part, int stat_mode, const BnBwdFuse* bn_bwd);
#else
static inline bool conv_wino_bx6_ok(const ConvGeom&) { return false; }
static inline int conv_wino_bx6_blocks(const ConvGeom&, int) { return 0; }
static inline double conv_wino_bx6_executed_flops(const ConvGeom&) { return 0.0; }
static inline void conv_wino_bx6_transform_weights(const float*, float*, const ConvGeom&, bool, hipStream_t) {}
static inline void conv_wino_bx6_launch(const float*, const float*, const float*, float*, const ConvGeom&, int, hipStream_t, float*, int,
                                        const BnBwdFuse*) {}
#endif

namespace {


template <int BTX>
struct WinoGeom {
    static constexpr int BTY = 64 / BTX;
    static constexpr int PXH = BTX + 1;                  // 16-B slots per (input row, parity)
    static constexpr int ROWSLOTS = 2 * PXH;
    // tile-row stride in 16-B slots.  Measured with SQ_LDS_BANK_CONFLICT: a ds_read_b128 is conflict
    // free when lanes 0-7 / 24-31 and 8-15 / 16-23 land on different halves of a 256-B window, so the
    // stride is padded to 8 (mod 16) slots when a tile row holds 8 tiles, 0 (mod 16) when it holds
    // 16, and 12 (mod 16) when it holds 4 (four tile rows per 16 lanes).
    static constexpr int RSTRIDE = BTX == 4 ? 44 : BTX == 16 ? 144 : 4 * ROWSLOTS;
    static constexpr int HALF_SLOTS = BTY * RSTRIDE;
    static constexpr int A_SLOTS = 2 * HALF_SLOTS;
    static constexpr int A_PIECES = (A_SLOTS + 63) / 64;  // 1-KiB pieces (18 / 18 / 22)
    static constexpr int A_FLOATS = A_PIECES * 256;
    static constexpr int B_FLOATS = 32 * 256;            // 16 pos x 2 channel quads x 64 couts x 4
    static constexpr int STAGE = A_FLOATS + B_FLOATS;
    static constexpr int E_FLOATS = 16 * 32 * 64;        // the output transform's exchange area: one 32-tile x 64-channel half
    static constexpr size_t LDS_BYTES = (size_t)(2 * STAGE > E_FLOATS ? 2 * STAGE : E_FLOATS) * sizeof(float);
    static_assert(A_PIECES <= 32, "two A pieces per wave at most");
};

template <int BTX, int SM>
__global__ __launch_bounds__(1024) void conv_wino_kernel(WinoArgs a) {
    constexpr bool STATS = SM != 0;
    using G = WinoGeom<BTX>;
    constexpr int BTY = G::BTY, PXH = G::PXH, ROWSLOTS = G::ROWSLOTS, RSTRIDE = G::RSTRIDE;
    constexpr int HALF_SLOTS = G::HALF_SLOTS, A_SLOTS = G::A_SLOTS, A_PIECES = G::A_PIECES;
    constexpr int A_FLOATS = G::A_FLOATS, STAGE = G::STAGE;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane0 = t & 63;   // wave == position
    // Persistent grid (launch_wino2): one block per CU walks the tile blocks b, b + gridDim.x, ... -- a training step is
    // 129 k tile blocks of 16 waves, and starting waves is what the dispatcher does at a finite rate
    // (profiles/r02_wino_pipe_ab.txt).  gridDim.x is a multiple of 8 or the whole grid: a block stays on the contiguous
    // range xcd_remap gives its XCD.
    // Dynamic assignment (device_common.h wq_*; a.work): thread 0 asks for the NEXT tile block at the top of the one it is in and
    // parks the answer in LDS at its end -- a workgroup that starts late, beside a collective that holds CUs, costs its share only.
    const int total_tiles = a.mblocks * a.nblocks;
    const bool dyn = a.work != nullptr;
    const int xcd = (int)blockIdx.x & 7;
    int* const Qw = reinterpret_cast<int*>(smem + G::LDS_BYTES / sizeof(float));
    if (dyn) {
        if (t == 0) Qw[0] = wq_claim(a.work, xcd, 0, 0, total_tiles);
    }
    for (int it = 0;; ++it) {
    if (it != 0 || dyn) __syncthreads();                 // the previous tile block's last LDS reads are done (and Qw is published)
    const int lt = dyn ? __builtin_amdgcn_readfirstlane(Qw[it & 1]) : (int)blockIdx.x + it * (int)gridDim.x;
    if (lt < 0 || lt >= total_tiles) break;
    int claim_k = 0;
    if (dyn && t == 0) claim_k = atomicAdd(a.work + xcd, 1);
    // opaque copy of the lane index: otherwise every lane-derived address is hoisted out of this loop, stays live across
    // the stage loop and the body no longer compiles as it does without the loop (spills, +3 % time)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int logical = xcd_remap(lt, total_tiles);
    const int nb = logical % a.nblocks, mb = logical / a.nblocks;
    const int rb = mb / a.txb, cb = mb - rb * a.txb;
    const int R0 = rb * BTY, tx0 = cb * BTX, n0 = nb * 64;

    // ---- staging descriptors: A pieces `wave` and `16 + wave`, B pieces of this wave's position ----
    unsigned avoff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int s = (wave + 16 * q) * 64 + lane;
        unsigned vo = 0x80000000u;
        if (s < A_SLOTS) {
            const int half = s / HALF_SLOTS, rem = s - half * HALF_SLOTS;
            const int r = rem / RSTRIDE, rem2 = rem - r * RSTRIDE;
            const int i = rem2 / ROWSLOTS, rem3 = rem2 - i * ROWSLOTS;
            const int par = rem3 / PXH, pxh = rem3 - par * PXH;
            const int R = R0 + r;
            if (i < 4 && R < a.rows) {
                const int n = (int)(((float)R + 0.5f) * a.inv_ty), ty = R - n * a.TY;
                const int yy = 2 * ty - 1 + i, xx = 2 * tx0 - 1 + 2 * pxh + par;
                if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W)
                    vo = (unsigned)(((n * a.H + yy) * a.W + xx) * a.Cin * 4 + half * 16);
            }
        }
        avoff[q] = vo;
    }
    const bool second_a = wave + 16 < A_PIECES;
    const unsigned bvoff = (unsigned)((n0 + lane) * 16);
    const __amdgpu_buffer_rsrc_t xsrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)a.N * a.H * a.W * a.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t usrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, (int)((size_t)16 * a.Cin * a.Cout * 4), 0x00020000);
    const int cq_total = a.Cin >> 2;

    auto issue = [&](int buf, int chunk) {
        float* As = smem + buf * STAGE;
        float* Bs = As + A_FLOATS;
        const int asoff = chunk * 32;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (__attribute__((address_space(3))) void*)(As + wave * 256), 16,
                                                 (int)avoff[0], asoff, 0, 0);
        if (second_a)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                xsrd, (__attribute__((address_space(3))) void*)(As + (wave + 16) * 256), 16, (int)avoff[1], asoff, 0, 0);
#pragma unroll
        for (int cq = 0; cq < 2; ++cq) {
            const int bsoff = ((wave * cq_total + chunk * 2 + cq) * a.Cout) * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                usrd, (__attribute__((address_space(3))) void*)(Bs + (wave * 2 + cq) * 256), 16, (int)bvoff, bsoff, 0, 0);
        }
    };

    // ---- this wave's position: V = (d[ra][ca] + sa d[rb][ca]) + sb (d[ra][cb] + sa d[rb][cb]) ----
    //   xi/nu = 0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3        (rows of B^T)
    const int xi = wave >> 2, nu = wave & 3;
    const int ra = xi == 0 ? 0 : xi == 2 ? 2 : 1, rbw = xi == 0 ? 2 : xi == 1 ? 2 : xi == 2 ? 1 : 3;
    const int ca = nu == 0 ? 0 : nu == 2 ? 2 : 1, cbw = nu == 0 ? 2 : nu == 1 ? 2 : nu == 2 ? 1 : 3;
    const int l31 = lane & 31, half = lane >> 5;
    const int tr = l31 / BTX, tcol = l31 - tr * BTX;       // tile (of the first 32) -> (tile row, column)
    constexpr int TG = (32 / BTX) * RSTRIDE * 4;           // float offset of the second 32 tiles
    const int lane_a = (half * HALF_SLOTS + tr * RSTRIDE + tcol) * 4;
    auto slot = [&](int i, int j) { return (i * ROWSLOTS + (j & 1) * PXH + (j >> 1)) * 4; };
    const int o_aa = lane_a + slot(ra, ca), o_ba = lane_a + slot(rbw, ca);
    const int o_ab = lane_a + slot(ra, cbw), o_bb = lane_a + slot(rbw, cbw);
    const int lane_b = A_FLOATS + wave * 512 + (half * 64 + l31) * 4;

    f32x16 acc1[1][2][2];
    f32x16 (&acc)[2][2] = acc1[0];

    // The two signs of the position's transform are compile-time (four copies of the stage loop, picked
    // by a wave-uniform branch; every copy executes the same barriers): adds / subtracts instead of
    // multiplies by +-1 held in registers.  (A v_pk_add_f32 neg_lo/neg_hi inline-asm subtract was 2 %
    // faster but returned wrong data next to the MFMAs -- the hazard recogniser does not see into
    // inline asm -- so the subtractions stay scalar.)
    // `first` (compile-time): the accumulators start from the MFMA's inline-constant zero C operand
    // instead of 64 register clears.
    auto stage_loop = [&](auto SA, auto SB) {
        constexpr bool PA = decltype(SA)::value, PB = decltype(SB)::value;
        auto compute = [&](int buf, auto first) {
            const float* S = smem + buf * STAGE;
            f32x4 v[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 daa = *reinterpret_cast<const f32x4*>(S + o_aa + i * TG);
                const f32x4 dba = *reinterpret_cast<const f32x4*>(S + o_ba + i * TG);
                const f32x4 dab = *reinterpret_cast<const f32x4*>(S + o_ab + i * TG);
                const f32x4 dbb = *reinterpret_cast<const f32x4*>(S + o_bb + i * TG);
                const f32x4 t0 = PA ? daa + dba : daa - dba, t1 = PA ? dab + dbb : dab - dbb;
                v[i] = PB ? t0 + t1 : t0 - t1;
            }
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                b[jn] = *reinterpret_cast<const f32x4*>(S + lane_b + jn * 128);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) {
                        if constexpr (decltype(first)::value) {
                            if (j == 0) {
                                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                                                     0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i][j], b[jn][j], zero, 0, 0, 0);
                                continue;
                            }
                        }
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i][j], b[jn][j], acc[i][jn], 0, 0, 0);
                    }
        };
        // The barrier (and the vmcnt(0) in front of it) must stay BEHIND the stage's MFMAs: they touch
        // no memory, so the scheduler would otherwise hoist the barrier to right after the LDS reads and
        // every stage would wait out the full latency of the loads it has just issued.
        auto stage_barrier = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        };
        if (a.nchunks > 1) issue(1, 1);
        compute(0, TrueT{});
        stage_barrier();
        for (int c = 1; c < a.nchunks; c += 2) {          // odd stages live in buffer 1
            if (c + 1 < a.nchunks) issue(0, c + 1);
            compute(1, FalseT{});
            stage_barrier();
            if (c + 1 < a.nchunks) {
                if (c + 2 < a.nchunks) issue(1, c + 2);
                compute(0, FalseT{});
                stage_barrier();
            }
        }
    };
    issue(0, 0);
    __syncthreads();
    if (xi == 1) {
        if (nu == 1) stage_loop(TrueT{}, TrueT{}); else stage_loop(TrueT{}, FalseT{});
    } else {
        if (nu == 1) stage_loop(FalseT{}, TrueT{}); else stage_loop(FalseT{}, FalseT{});
    }

    wino_output<BTX, SM, 16>(a, acc1, smem, t, wave, lane, R0, tx0, n0, mb);
    if (dyn && t == 0) {
        int nx = wq_base(0, xcd) + 8 * claim_k;
        if (nx >= total_tiles) nx = wq_claim(a.work, xcd, 1, 0, total_tiles);
        Qw[(it + 1) & 1] = nx;
    }
    }   // tile-block loop
    if (dyn && t == 0) wq_leave(a.work, (int)gridDim.x);
}

// U[pos][c/4][k][c%4] = (G g G^T)[pos] for every (input channel c, output channel k).
// from_fwd_for_dgrad: g[kh][kw][c][k] = w[2-kh][2-kw][k][c] with w the FORWARD filter
// (Cin_fwd = Cout here, Cout_fwd = Cin here) -- the data-gradient filter without a flip pass.
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin,
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
    // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    float gg[4][3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        gg[0][kw] = g[0][kw];
        gg[1][kw] = 0.5f * (g[0][kw] + g[1][kw] + g[2][kw]);
        gg[2][kw] = 0.5f * (g[0][kw] - g[1][kw] + g[2][kw]);
        gg[3][kw] = g[2][kw];
    }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
        const float o0 = gg[xi][0];
        const float o1 = 0.5f * (gg[xi][0] + gg[xi][1] + gg[xi][2]);
        const float o2 = 0.5f * (gg[xi][0] - gg[xi][1] + gg[xi][2]);
        const float o3 = gg[xi][2];
        const float o[4] = {o0, o1, o2, o3};
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
            u[(((size_t)(xi * 4 + nu) * (Cin >> 2) + (c >> 2)) * Cout + k) * 4 + (c & 3)] = o[nu];
    }
}

template <int BTX, int SM>
void launch_wino2(const WinoArgs& a, hipStream_t s) {
    using G = WinoGeom<BTX>;
    // the opt-in for > 64 KiB of dynamic LDS is per device: once per (kernel instantiation, device)
    static std::once_flag once[L3_MAX_DEVICES];
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & (L3_MAX_DEVICES - 1)], [] {
        (void)hipFuncSetAttribute((const void*)conv_wino_kernel<BTX, SM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES + 16);
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
    // L3_W4_NCU (debug knob, read per call): a small chip emulated, as in conv_wino4_launch
    const int ncu_k = l3_knob("L3_W4_NCU") ? atoi(l3_knob("L3_W4_NCU")) / 8 * 8 : 0;
    const int ncu_eff = ncu_k > 0 && ncu_k < ncu ? ncu_k : ncu;
    const int dyn_on = l3_knob("L3_W4_DYNAMIC") ? atoi(l3_knob("L3_W4_DYNAMIC")) : a.dynamic;       // (read per call: the tests switch it)
    WinoArgs m = a;
    m.work = persist && total > ncu_eff && dyn_on ? persistent_work_counters(s) : nullptr;      // ConvGeom::dynamic: tile blocks through work counters
    hipLaunchKernelGGL((conv_wino_kernel<BTX, SM>), dim3(persist && total > ncu_eff ? ncu_eff : total), dim3(1024), G::LDS_BYTES + 16, s, m);
}
template <int BTX>
void launch_wino(const WinoArgs& a, hipStream_t s) {
    if (a.stat_part != nullptr && a.bb.x != nullptr)
        launch_wino2<BTX, 2>(a, s);
    else if (a.stat_part != nullptr)
        launch_wino2<BTX, 1>(a, s);
    else
        launch_wino2<BTX, 0>(a, s);
}

// tile-column block width with the least padding (ties: the widest), and the block counts it implies
struct WinoPlan {
    int btx, txb, mblocks;
};
WinoPlan wino_plan(const ConvGeom& g) {
    const int TY = (g.H + 1) / 2, TX = (g.W + 1) / 2;
    int best = 16, waste = ((TX + 15) / 16) * 16;
    for (int btx : {8, 4}) {
        const int wst = ((TX + btx - 1) / btx) * btx;
        if (wst < waste) {
            waste = wst;
            best = btx;
        }
    }
    static const int force = l3_knob("L3_WINO_BTX") ? atoi(l3_knob("L3_WINO_BTX")) : 0;
    if (force == 4 || force == 8 || force == 16) best = force;
    WinoPlan p;
    p.btx = best;
    p.txb = (TX + best - 1) / best;
    const int bty = 64 / best;
    p.mblocks = ((g.N * TY + bty - 1) / bty) * p.txb;
    return p;
}

}  // namespace

bool conv_wino_ok(const ConvGeom& g) {
    static const int enabled = l3_knob("L3_WINOGRAD") ? atoi(l3_knob("L3_WINOGRAD")) : 1;
    return enabled && g.KH == 3 && g.KW == 3 && g.padT == 1 && g.padL == 1 && g.Ho == g.H && g.Wo == g.W &&
           g.Cin % 8 == 0 && g.Cout % 64 == 0 && (size_t)16 * g.Cin * g.Cout * 4 < (1ull << 31) &&
           (size_t)g.H * g.W * (g.Cin > g.Cout ? g.Cin : g.Cout) * 4 < (1ull << 31);
}

// The kernels address a tensor through 32-bit buffer offsets (< 2 GiB).  Bigger batches -- 288 GB of
// HBM invite them -- are cut into sample ranges that fit; samples are independent in a convolution.
static int wino_chunk_samples(const ConvGeom& g) {
    const size_t per_sample = (size_t)g.H * g.W * (g.Cin > g.Cout ? g.Cin : g.Cout) * 4;
    size_t nc = ((1ull << 31) - 1) / per_sample;
    const size_t row_cap = ((1u << 22) - 1) / (size_t)((g.H + 1) / 2);
    if (nc > row_cap) nc = row_cap;
    if (nc > (size_t)g.N) nc = (size_t)g.N;
    return nc < 1 ? 1 : (int)nc;
}

static bool use_bx6(const ConvGeom& g) { return g.f2x2 == 2 && conv_wino_ok(g) && conv_wino_bx6_ok(g); }
static bool use_wino4(const ConvGeom& g) { return !use_bx6(g) && conv_wino_ok(g) && conv_wino4_selected(g); }

bool conv_wino_is_bx6(const ConvGeom& g) { return use_bx6(g); }

double conv_wino_executed_flops(const ConvGeom& g) {
    if (use_bx6(g)) return conv_wino_bx6_executed_flops(g);
    if (use_wino4(g)) return conv_wino4_executed_flops(g);
    return 2.0 * 16.0 * (double)g.N * ((g.H + 1) / 2) * ((g.W + 1) / 2) * (double)g.Cin * (double)g.Cout;
}

// sized for either algorithm (36 positions of F(4x4,3x3), 16 of F(2x2,3x3)): which one runs is decided per launch
size_t conv_wino_floats(const ConvGeom& g) { return conv_wino_ok(g) ? (size_t)36 * g.Cin * g.Cout : 0; }

void conv_wino_transform_weights(const float* w, float* u, const ConvGeom& g, bool from_fwd_for_dgrad, hipStream_t s) {
    if (use_bx6(g)) return conv_wino_bx6_transform_weights(w, u, g, from_fwd_for_dgrad, s);
    if (use_wino4(g)) return conv_wino4_transform_weights(w, u, g, from_fwd_for_dgrad, s);
    const int total = g.Cin * g.Cout;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, u, g.Cin, g.Cout,
                       from_fwd_for_dgrad ? 1 : 0);
}

static int stat_blocks_of(const ConvGeom& g, int algo) {       // 0 F(2x2) fp32, 1 F(4x4), 2 F(2x2) split-bf16
    const int nc = wino_chunk_samples(g);
    int blocks = 0;
    for (int n0 = 0; n0 < g.N; n0 += nc) {
        ConvGeom gc = g;
        gc.N = g.N - n0 < nc ? g.N - n0 : nc;
        blocks += algo == 1 ? conv_wino4_blocks(g, gc.N) : algo == 2 ? conv_wino_bx6_blocks(g, gc.N) : wino_plan(gc).mblocks;
    }
    return blocks;
}

int conv_wino_stat_blocks(const ConvGeom& g) { return conv_wino_ok(g) ? stat_blocks_of(g, use_bx6(g) ? 2 : use_wino4(g) ? 1 : 0) : 0; }

int conv_wino_stat_blocks_max(const ConvGeom& g) {
    if (!conv_wino_ok(g)) return 0;
    int m = stat_blocks_of(g, 0);
    const int b = stat_blocks_of(g, 1);
    if (b > m) m = b;
    if (g.Cin % 16 == 0 && conv_wino_bx6_ok(g)) {
        const int c = stat_blocks_of(g, 2);
        if (c > m) m = c;
    }
    return m;
}

void conv_wino_fwd(const float* x, const float* u, const float* bias, float* y, const ConvGeom& g, hipStream_t s,
                   float* stat_part, int stat_mode, const BnBwdFuse* bn_bwd) {
    const int nc = wino_chunk_samples(g);
    if (use_bx6(g)) {
        for (int n0 = 0; n0 < g.N; n0 += nc) {
            const int n = g.N - n0 < nc ? g.N - n0 : nc;
            BnBwdFuse bb;
            if (bn_bwd != nullptr) {
                bb = *bn_bwd;
                bb.x = bn_bwd->x + (size_t)n0 * g.H * g.W * g.Cout;      // the sample range of this launch
            }
            conv_wino_bx6_launch(x + (size_t)n0 * g.H * g.W * g.Cin, u, bias, y + (size_t)n0 * g.H * g.W * g.Cout, g, n, s, stat_part,
                                 stat_mode, bn_bwd != nullptr ? &bb : nullptr);
            if (stat_part != nullptr) stat_part += (size_t)conv_wino_bx6_blocks(g, n) * 2 * g.Cout;
        }
        return;
    }
    if (use_wino4(g)) {
        for (int n0 = 0; n0 < g.N; n0 += nc) {
            const int n = g.N - n0 < nc ? g.N - n0 : nc;
            BnBwdFuse bb;
            if (bn_bwd != nullptr) {
                bb = *bn_bwd;
                bb.x = bn_bwd->x + (size_t)n0 * g.H * g.W * g.Cout;      // the sample range of this launch
            }
            conv_wino4_launch(x + (size_t)n0 * g.H * g.W * g.Cin, u, bias, y + (size_t)n0 * g.H * g.W * g.Cout, g, n, s,
                              stat_part, stat_mode, bn_bwd != nullptr ? &bb : nullptr);
            if (stat_part != nullptr) stat_part += (size_t)conv_wino4_blocks(g, n) * 2 * g.Cout;
        }
        return;
    }
    for (int n0 = 0; n0 < g.N; n0 += nc) {
        ConvGeom gc = g;
        gc.N = g.N - n0 < nc ? g.N - n0 : nc;
        WinoArgs a;
        a.dynamic = g.dynamic;
        a.x = x + (size_t)n0 * g.H * g.W * g.Cin;
        a.u = u; a.bias = bias;
        a.y = y + (size_t)n0 * g.H * g.W * g.Cout;
        a.N = gc.N; a.H = g.H; a.W = g.W; a.Cin = g.Cin; a.Cout = g.Cout;
        a.TY = (g.H + 1) / 2;
        a.TX = (g.W + 1) / 2;
        a.rows = gc.N * a.TY;
        a.nblocks = g.Cout / 64;
        a.nchunks = g.Cin / 8;
        a.inv_ty = 1.0f / (float)a.TY;
        a.stat_part = stat_part;
        a.stat_mode = stat_mode;
        a.bb = BnBwdFuse{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0};
        if (bn_bwd != nullptr && stat_part != nullptr) {
            a.bb = *bn_bwd;
            a.bb.x = bn_bwd->x + (size_t)n0 * g.H * g.W * g.Cout;      // the sample range of this launch
        }
        const WinoPlan p = wino_plan(gc);
        a.txb = p.txb;
        a.mblocks = p.mblocks;
        if (p.btx == 16)
            launch_wino<16>(a, s);
        else if (p.btx == 8)
            launch_wino<8>(a, s);
        else
            launch_wino<4>(a, s);
        if (stat_part != nullptr) stat_part += (size_t)p.mblocks * 2 * g.Cout;
    }
}

}  // namespace l3
