"""Developer probe: timing-only ablations of conv_wgrad_wino.hip (results invalid).  Writes an ablated copy of the source.
usage: WGW_ABL=<nolds|novalu|nodma|nobarrier>[,..] python scripts/probes/wgw_ablate.py <out.hip>"""
import sys, os
src = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'l3embedding_amd', 'csrc', 'conv_wgrad_wino.hip')
s = open(src).read()
def rep(old, new):
    global s
    assert old in s, old[:70]
    s = s.replace(old, new, 1)
abl = os.environ.get('WGW_ABL', '').split(',')
if 'nolds' in abl:       # no LDS reads: the raw values are the lane index
    rep("__device__ __forceinline__ f32x2 lds_f32x2(const char* p) {\n    const f32x2 v = *reinterpret_cast<const f32x2*>(p);",
        "__device__ __forceinline__ f32x2 lds_f32x2(const char* p) {\n    const f32x2 v = {(float)(size_t)p, 1.f};")
if 'novalu' in abl:      # operands straight from the first raw value
    rep("            v[i] = __builtin_fmaf(sb, __builtin_fmaf(sa, r.x[i][3], r.x[i][2]), __builtin_fmaf(sa, r.x[i][1], r.x[i][0]));",
        "            v[i] = r.x[i][0];")
    rep("            float zz = w00 * r.y[i][0];\n            if (NC == 2) zz = __builtin_fmaf(w01, r.y[i][1], zz);\n            if (NR == 2) zz = __builtin_fmaf(w10, r.y[i][2], zz);\n            if (NR == 2 && NC == 2) zz = __builtin_fmaf(w11, r.y[i][3], zz);",
        "            float zz = r.y[i][0];")
if 'nodma' in abl:       # counters advance, nothing is loaded
    rep("            if (pc_isx[q])\n                __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd,", "            if (false && pc_isx[q])\n                __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd,")
    rep("            else\n                __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd,", "            else if (false)\n                __builtin_amdgcn_raw_ptr_buffer_load_lds(ysrd,")
if 'nobarrier' in abl:
    rep("    auto stage_barrier = [&]() {          // behind the stage's MFMAs (see conv_wino.hip)\n        __builtin_amdgcn_sched_barrier(0);\n        __syncthreads();",
        "    auto stage_barrier = [&]() {\n        __builtin_amdgcn_sched_barrier(0);")
open(sys.argv[1], 'w').write(s)
