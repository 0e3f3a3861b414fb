"""Developer probe: writes an instrumented copy of conv_wino4.hip (s_memtime stamps by wave W4_TWAVE of every tile block:
top barrier / set-up / first loads landed / first transform / stage loop / output-transform rounds) for
scripts/probes/w4_timing.py.  Optional timing-only ablations (results invalid): W4_ABL = comma list of nodma, noprep, nostore, noprefetch.
usage: [W4_ABL=...] python scripts/probes/w4_instrument.py <out.hip>
       hipcc ... -DW4_TWAVE=<wave> -c <out.hip>, linked in place of conv_wino4.o into a copy of libl3hip.so"""
import sys, os
src = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'l3embedding_amd', 'csrc', 'conv_wino4.hip')
s = open(src).read()
def rep(old, new, count=1):
    global s
    assert old in s, old[:70]
    s = s.replace(old, new, count)
rep("template <int SM, bool PART = false>\n__global__ __launch_bounds__(W4_THREADS) void conv_wino4_kernel(Wino4Args a) {",
'''__device__ unsigned long long g_w4_t[16];
#define W4_T(i) do { if (wave == W4_TWAVE) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; } } while (0)
template <int SM, bool PART = false>
__global__ __launch_bounds__(W4_THREADS) void conv_wino4_kernel(Wino4Args a) {''')
rep("    for (int lt = lt_begin; lt < a.lt_end; lt += lt_step) {\n",
'''    unsigned long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    for (int lt = lt_begin; lt < a.lt_end; lt += lt_step) {
''')
rep("    if (lt != lt_begin) lds_barrier();                   // the previous tile block's last LDS reads are done\n",
    "    if (lt != (int)blockIdx.x) lds_barrier();            // the previous tile block's last LDS reads are done\n    W4_T(9);\n")
rep("    // prologue: patches of stage 0 and both filter halves of stage 0", "    W4_T(10);\n    // prologue: patches of stage 0 and both filter halves of stage 0")
rep("    if (!have0) __syncthreads();\n", "    if (!have0) __syncthreads();\n    W4_T(0);\n")
rep("        default: stage_loop(IntT<5>{}, IntT<1>{}); break;\n    }\n", "        default: stage_loop(IntT<5>{}, IntT<1>{}); break;\n    }\n    W4_T(1);\n")
rep("        __syncthreads();                                   // patches of stage 1 landed; slot 0 read by every wave\n        SB();\n",
    "        __syncthreads();                                   // patches of stage 1 landed; slot 0 read by every wave\n        SB();\n        W4_T(11);\n")
rep("            lds_barrier();\n            if (!worker && has_next) {\n                if (jn == 0 && th == 0) prefetch_chunk",
    "            lds_barrier();\n            W4_T(2);\n            if (!worker && has_next) {\n                if (jn == 0 && th == 0) prefetch_chunk")
rep("            if (worker) {\n                const float bz = smem[W4_BIAS_BASE + jn * 32 + o_c32];", "            W4_T(4);\n            if (worker) {\n                const float bz = smem[W4_BIAS_BASE + jn * 32 + o_c32];")
rep("                        (okq >> k) & 1 ? (int)qbase : (int)0x80000000u, k * so_x, 0);\n            }\n",
    "                        (okq >> k) & 1 ? (int)qbase : (int)0x80000000u, k * so_x, 0);\n            }\n            W4_T(3);\n")
rep("    have0 = has_next;\n", "    have0 = has_next;\n    W4_T(6);\n    if (wave == W4_TWAVE) { tacc[7] += 1; tacc[8] += a.nchunks; }\n")
rep("    }   // tile-block loop\n", "    }   // tile-block loop\n    if (t == W4_TWAVE * 64) for (int i = 0; i < 16; ++i) atomicAdd(&g_w4_t[i], tacc[i]);\n")
rep("}  // namespace\n", "}  // namespace\nextern \"C\" int l3_debug_w4_timing(unsigned long long* out, int reset) {\n    unsigned long long z[16] = {0};\n    (void)hipDeviceSynchronize();\n    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w4_t), sizeof(z));\n    if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_w4_t), z, sizeof(z));\n    return 0;\n}\n")
abl = os.environ.get('W4_ABL', '').split(',')
if 'nodma' in abl:          # no LDS-DMA inside the stage loop
    rep("        auto u_piece = [&](int sub, int pp, int c8) {       // the 1-KiB filter slice (position, half-stage) of stage c8\n",
        "        auto u_piece = [&](int sub, int pp, int c8) {\n            return;\n")
    rep("        auto a_piece = [&](int slot, int c8, int q) {       // patch piece q of this wave\n",
        "        auto a_piece = [&](int slot, int c8, int q) {\n            return;\n")
if 'nostore' in abl:        # no output stores
    rep("                        (okq >> k) & 1 ? (int)qbase : (int)0x80000000u, k * so_x, 0);\n            }\n            W4_T(3);",
        "                        yv[k] == 12345.678f ? (int)qbase : (int)0x80000000u, k * so_x, 0);\n            }\n            W4_T(3);")
if 'noprefetch' in abl:     # the next tile block's first stage is not requested from the epilogue
    rep("    const bool has_next = !PART && lt + lt_step < a.lt_end;", "    const bool has_next = false;")
if 'noprep' in abl:         # no patch reads / input transform inside the stage loop
    rep("        auto col_read = [&](const float* SA, auto JJT, f32x4 (&d)[5]) {\n", "        auto col_read = [&](const float* SA, auto JJT, f32x4 (&d)[5]) {\n            return;\n")
    rep("        auto col_comb = [&](auto JJT, const f32x4 (&d)[5], f32x4 (&vn)[3]) {\n", "        auto col_comb = [&](auto JJT, const f32x4 (&d)[5], f32x4 (&vn)[3]) {\n            vn[0] = vn[1] = vn[2] = f32x4{1.f, 1.f, 1.f, 1.f};\n            return;\n")
open(sys.argv[1], 'w').write(s)
