// halo_bench -- times libl3hip.so's mixed-precision convolution launches (conv_bf16_halo_launch, conv_wgrad_bf16_tr_launch) on
// device-resident random bf16 tensors, one line per geometry: the per-layer numbers of DESIGN 4b without a profiler, and sweeps
// over Cin at fixed Cout (time = fixed cost per block + slope x Cin) that say what a block costs outside its MFMA loop.
//   build:  hipcc -O2 -std=c++17 scripts/probes/halo_bench.cpp -Il3embedding_amd/csrc -Ll3embedding_amd/lib -ll3hip -Wl,-rpath,'$ORIGIN/../../l3embedding_amd/lib' -o scripts/probes/halo_bench
//   run:    L3_DEBUG_KNOBS=1 [L3_HALO_FLAT=0 ...] scripts/probes/halo_bench <set> [batch]      set = layers | sweep | wgrad
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlfcn.h>

#include <algorithm>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale, int relu) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        float v = ((int)(h & 0xffff) - 32768) * (scale / 32768.f);
        if (relu && v < 0.f) v = 0.f;
        p[i] = (unsigned short)(__float_as_uint(v) >> 16);
    }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((int)(h & 0xffff) - 32768) * (scale / 32768.f);
    }
}

struct Layer { const char* name; int H, W, Cin, Cout; };

static double time_launches(int reps, hipStream_t s, const std::function<void()>& f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

// HB_STAMPS=1 with a library built by scripts/probes/build_halo_stamps.sh (-DL3_HALO_STAMPS): mean phase times of a block
// (wave 0's wall-clock stamps, 10-ns ticks) and how many blocks were alive on average
static void print_stamps(const char* what, int nblocks, hipStream_t s) {
    typedef int (*fn_t)(unsigned long long*, int);
    static fn_t fn = (fn_t)dlsym(RTLD_DEFAULT, "l3_dbg_halo_stamps");
    if (fn == nullptr) { printf("   (library has no stamps)\n"); return; }
    CK(hipStreamSynchronize(s));
    if (nblocks > (1 << 16)) nblocks = 1 << 16;
    std::vector<unsigned long long> st((size_t)nblocks * 8);
    if (fn(st.data(), nblocks) != 0) { printf("   (stamp copy failed)\n"); return; }
    double ph[6] = {0, 0, 0, 0, 0, 0}, life = 0;
    unsigned long long t0 = ~0ull, t1 = 0;
    std::vector<double> lifes;
    for (int b = 0; b < nblocks; ++b) {
        const unsigned long long* q = &st[(size_t)b * 8];
        for (int k = 0; k < 6; ++k) ph[k] += (double)(long long)(q[k + 1] - q[k]);
        life += (double)(q[6] - q[0]);
        lifes.push_back((double)(q[6] - q[0]));
        t0 = std::min(t0, q[0]); t1 = std::max(t1, q[6]);
    }
    std::sort(lifes.begin(), lifes.end());
    const double k = 0.01 / nblocks;   // ticks -> us, mean
    printf("   %s: %d blocks, span %.1f us, alive on average %.1f blocks = %.2f per CU | block life mean %.2f us (median %.2f, p90 %.2f)\n", what, nblocks,
           (t1 - t0) * 0.01, life / (double)(t1 - t0), life / (double)(t1 - t0) / 256.0, life * k, lifes[nblocks / 2] * 0.01, lifes[nblocks * 9 / 10] * 0.01);
    // per CU (HW_ID bits 8.. + XCC_ID): how many blocks are in their tap phases ([1,2] and [3,4]) at the same time -- independent blocks
    // give a binomial picture, blocks that run in phase (all load, all multiply, all store together) a 0-or-all one
    {
        std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
        for (int b = 0; b < nblocks; ++b) {
            const unsigned long long* q = &st[(size_t)b * 8];
            const unsigned long long cu = ((q[7] >> 32) << 32) | ((q[7] >> 8) & 0xffull);     // XCC_ID | SE_ID, SH_ID, CU_ID (HW_ID bits 15:8)
            auto& e = ev[cu];
            e.push_back({q[1], 1}); e.push_back({q[2], -1});
            if (q[3] > q[2] && q[4] >= q[3]) { e.push_back({q[3], 1}); e.push_back({q[4], -1}); }
        }
        double lvl[9] = {0}, tot = 0;
        for (auto& kv : ev) {
            auto& e = kv.second;
            std::sort(e.begin(), e.end());
            int c = 0;
            for (size_t i = 0; i + 1 < e.size(); ++i) {
                c += e[i].second;
                const double d = (double)(e[i + 1].first - e[i].first);
                lvl[c < 0 ? 0 : c > 8 ? 8 : c] += d; tot += d;
            }
        }
        printf("      %zu CUs seen; share of a CU's time with 0 / 1 / 2 / 3 / 4+ blocks in their tap phases: %.2f %.2f %.2f %.2f %.2f\n", ev.size(), lvl[0] / tot, lvl[1] / tot,
               lvl[2] / tot, lvl[3] / tot, (lvl[4] + lvl[5] + lvl[6] + lvl[7] + lvl[8]) / tot);
    }
    printf("      entry->loaded %.2f | chunk-0 taps %.2f | refill %.2f | remaining taps %.2f | tile out %.2f | partials + exit %.2f us\n", ph[0] * k, ph[1] * k,
           ph[2] * k, ph[3] * k, ph[4] * k, ph[5] * k);
}

int main(int argc, char** argv) {
    const char* set = argc > 1 ? argv[1] : "layers";
    const int N = argc > 2 ? atoi(argv[2]) : 128;
    const int reps = argc > 3 ? atoi(argv[3]) : 10;
    std::vector<Layer> L;
    if (!strcmp(set, "layers") || !strcmp(set, "wgrad")) {
        L = {{"A.conv1b", 256, 199, 64, 64},  {"A.conv2a", 128, 99, 64, 128},  {"A.conv2b", 128, 99, 128, 128}, {"A.conv3a", 64, 49, 128, 256},
             {"A.conv3b", 64, 49, 256, 256},  {"A.conv4a", 32, 24, 256, 512},  {"A.conv4b", 32, 24, 512, 512},  {"V.conv1b", 224, 224, 64, 64},
             {"V.conv2a", 112, 112, 64, 128}, {"V.conv2b", 112, 112, 128, 128}, {"V.conv3a", 56, 56, 128, 256}, {"V.conv3b", 56, 56, 256, 256},
             {"V.conv4a", 28, 28, 256, 512},  {"V.conv4b", 28, 28, 512, 512}};
    } else {   // sweep: fixed cost per block vs slope in Cin
        for (int co : {64, 128})
            for (int ci : {64, 128, 256, 512}) {
                L.push_back({"sweep224", 224, 224, ci, co});
                L.push_back({"sweep56", 56, 56, ci, co});
            }
    }
    if (const char* only = getenv("HB_ONLY")) {          // HB_ONLY=conv1b,conv2a: layers whose name contains one of the substrings
        std::vector<Layer> keep;
        for (auto& l : L) {
            std::string o(only);
            for (size_t p = 0; p <= o.size();) {
                const size_t q = o.find(',', p) == std::string::npos ? o.size() : o.find(',', p);
                if (q > p && strstr(l.name, o.substr(p, q - p).c_str())) { keep.push_back(l); break; }
                p = q + 1;
            }
        }
        L = keep;
    }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    size_t maxe = 0;
    for (auto& l : L) {
        size_t e = (size_t)N * l.H * l.W * (l.Cin > l.Cout ? l.Cin : l.Cout);
        if (e > maxe) maxe = e;
    }
    unsigned short *x, *y, *bx;
    float *w32, *bias, *stat, *bnp, *part;
    void* wn;
    CK(hipMalloc(&x, maxe * 2)); CK(hipMalloc(&y, maxe * 2)); CK(hipMalloc(&bx, maxe * 2));
    CK(hipMalloc(&w32, 9 * 512 * 512 * 4)); CK(hipMalloc(&wn, 2 * 9 * 512 * 512 * 2)); CK(hipMalloc(&bias, 512 * 4));
    CK(hipMalloc(&bnp, 4 * 512 * 4));
    CK(hipMalloc(&stat, (size_t)64 << 20));
    CK(hipMalloc(&part, (size_t)512 << 20));
    // HB_DATA=zero: all-zero activations and filters (same instruction stream, nothing toggles): what the data costs in clock
    const float dscale = getenv("HB_DATA") && !strcmp(getenv("HB_DATA"), "zero") ? 0.f : 1.f;
    fill_bf16<<<4096, 256, 0, s>>>(x, maxe, 1u, dscale, 1);
    fill_bf16<<<4096, 256, 0, s>>>(y, maxe, 11u, dscale, 0);
    fill_bf16<<<4096, 256, 0, s>>>(bx, maxe, 7u, dscale, 0);
    fill_f32<<<256, 256, 0, s>>>(w32, 9 * 512 * 512, 3u, 0.05f * dscale);
    fill_f32<<<2, 256, 0, s>>>(bias, 512, 5u, 0.1f);
    fill_f32<<<8, 256, 0, s>>>(bnp, 4 * 512, 9u, 1.f);
    CK(hipStreamSynchronize(s));
    printf("# set %s  batch %d  reps %d\n", set, N, reps);
    for (auto& l : L) {
        l3::ConvGeom g{};
        g.N = N; g.H = l.H; g.W = l.W; g.Cin = l.Cin; g.Ho = l.H; g.Wo = l.W; g.Cout = l.Cout; g.KH = g.KW = 3; g.padT = g.padL = 1;
        const double gf = 2.0 * 9 * l.Cin * l.Cout * (double)N * l.H * l.W * 1e-9;
        if (!strcmp(set, "wgrad")) {
            const int tiles = (l.Cin / 64) * (l.Cout / 64);
            int splits = 512 / tiles; if (splits < 1) splits = 1;
            if (getenv("WG_SPLITS")) splits = atoi(getenv("WG_SPLITS")) / tiles > 0 ? atoi(getenv("WG_SPLITS")) / tiles : 1;
            const double us = time_launches(reps, s, [&] { l3::conv_wgrad_bf16_tr_launch(x, y, part, g, N, splits, s); });
            printf("%-9s %3dx%-3d %3d->%-3d wgrad (%d splits) %8.1f us  %7.1f TF/s  %.3f of 2.5 PF\n", l.name, l.H, l.W, l.Cin, l.Cout, splits, us, gf / us * 1e3,
                   gf / us / 2.5);
            continue;
        }
        l3::conv_weights_bf16(w32, wn, 3, 3, l.Cin, l.Cout, true, s);
        // forward: statistics partials + bf16 output
        const double usf = time_launches(reps, s, [&] { l3::conv_bf16_halo_launch(x, wn, bias, y, g, N, s, stat, 1, true, nullptr); });
        // data gradient (the transposed layer): fused BatchNorm-backward partials
        l3::ConvGeom d = g;
        d.Cin = l.Cout; d.Cout = l.Cin;
        l3::BnBwdFuse bb{reinterpret_cast<const float*>(bx), bnp, bnp + 512, bnp + 1024, bnp + 1536, 1e-3f, 1};
        double usd = -1.0;
        if (d.Cout % 64 == 0) {
            l3::conv_weights_bf16(w32, wn, 3, 3, d.Cin, d.Cout, true, s);
            usd = time_launches(reps, s, [&] { l3::conv_bf16_halo_launch(x, wn, nullptr, y, d, N, s, stat, 1, true, &bb); });
        }
        printf("%-9s %3dx%-3d %3d->%-3d fwd %8.1f us %7.1f TF/s %.3f | dgrad %8.1f us %7.1f TF/s %.3f\n", l.name, l.H, l.W, l.Cin, l.Cout, usf, gf / usf * 1e3,
               gf / usf / 2.5, usd, gf / usd * 1e3, gf / usd / 2.5);
        if (getenv("HB_STAMPS")) {
            l3::conv_weights_bf16(w32, wn, 3, 3, l.Cin, l.Cout, true, s);
            l3::conv_bf16_halo_launch(x, wn, bias, y, g, N, s, stat, 1, true, nullptr);
            print_stamps("fwd  ", l3::conv_bf16_halo_patches(g, N) * (l.Cout / (l.Cout % 128 == 0 ? 128 : 64)), s);
            if (d.Cout % 64 == 0) {
                l3::conv_weights_bf16(w32, wn, 3, 3, d.Cin, d.Cout, true, s);
                l3::conv_bf16_halo_launch(x, wn, nullptr, y, d, N, s, stat, 1, true, &bb);
                print_stamps("dgrad", l3::conv_bf16_halo_patches(d, N) * (d.Cout / (d.Cout % 128 == 0 ? 128 : 64)), s);
            }
        }
    }
    return 0;
}
