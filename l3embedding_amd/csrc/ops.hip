// ops.hip -- stand-alone operator entry points of the C ABI (host buffers in/out).
// They run exactly the kernels the engine uses and exist for the op-level parity
// tests in tests/ (each op replaces a TF op instantiated by a Keras/kapre layer,
// SURVEY.md section 2.3).
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/l3hip.h"
#include "kernels.h"
#include "knobs.h"

namespace {
using namespace l3;

struct Scope {
    std::vector<void*> bufs;
    hipStream_t s = nullptr;
    bool ok = true;
    explicit Scope(int device) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= device || hipSetDevice(device) != hipSuccess) ok = false;
    }
    ~Scope() {
        (void)hipDeviceSynchronize();
        for (void* p : bufs) (void)hipFree(p);
    }
    template <class T>
    T* alloc(size_t count) {
        void* p = nullptr;
        if (hipMalloc(&p, (count ? count : 4) * sizeof(T)) != hipSuccess) {
            ok = false;
            return nullptr;
        }
        bufs.push_back(p);
        return static_cast<T*>(p);
    }
    template <class T>
    T* put(const T* host, size_t count) {
        T* d = alloc<T>(count);
        if (d && host && hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) ok = false;
        return d;
    }
    template <class T>
    void get(T* host, const T* dev, size_t count) {
        if (!host) return;
        if (hipDeviceSynchronize() != hipSuccess) ok = false;
        if (hipMemcpy(host, dev, count * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) ok = false;
    }
    int status() {
        if (hipDeviceSynchronize() != hipSuccess) ok = false;
        if (hipGetLastError() != hipSuccess) ok = false;
        return ok ? L3_OK : L3_EHIP;
    }
};

void tf_same(int n, int k, int s, int* out, int* before) {
    *out = (n + s - 1) / s;
    int total = (*out - 1) * s + k - n;
    if (total < 0) total = 0;
    *before = total / 2;
}

ConvGeom make_geom(int n, int h, int w, int cin, int cout, int kh, int kw, int same) {
    ConvGeom g{n, h, w, cin, 0, 0, cout, kh, kw, 0, 0};
    if (same) {
        tf_same(h, kh, 1, &g.Ho, &g.padT);
        tf_same(w, kw, 1, &g.Wo, &g.padL);
    } else {
        g.Ho = h - kh + 1;
        g.Wo = w - kw + 1;
    }
    g.solo = 1;          // an operator on its own: nothing queued beside it
    const char* algo = l3_knob("L3_FP32_CONV");          // the tests run every l3_config.fp32_conv value through the operators
    if (algo != nullptr) g.f2x2 = strcmp(algo, "f2x2_bf16x6") == 0 ? 2 : strcmp(algo, "f2x2") == 0 ? 1 : 0;
    return g;
}

PoolGeom make_pool(int n, int h, int w, int c, int ph, int pw, int sh, int sw, int same) {
    PoolGeom g{n, h, w, c, 0, 0, ph, pw, sh, sw, 0, 0, 0};
    if (same) {
        tf_same(h, ph, sh, &g.Ho, &g.padT);
        tf_same(w, pw, sw, &g.Wo, &g.padL);
    } else {
        g.Ho = (h - ph) / sh + 1;
        g.Wo = (w - pw) / sw + 1;
    }
    g.out_batch_stride = (int64_t)g.Ho * g.Wo * c;
    return g;
}
}  // namespace

extern "C" {

int l3_op_conv2d_fwd_dt(int device, int dtype, const float* x, const float* w, const float* b, float* y, int n, int h,
                        int wd, int cin, int cout, int kh, int kw, int same) {
    Scope sc(device);
    if (!sc.ok) return L3_EHIP;
    const ConvGeom g = make_geom(n, h, wd, cin, cout, kh, kw, same);
    float* dx = sc.put(x, (size_t)n * h * wd * cin);
    float* dw = sc.put(w, (size_t)kh * kw * cin * cout);
    float* db = b ? sc.put(b, (size_t)cout) : nullptr;
    float* dy = sc.alloc<float>((size_t)n * g.Ho * g.Wo * cout);
    if (!sc.ok) return L3_ENOMEM;
    if ((dtype == L3_OP_BF16_STORED || dtype == L3_OP_BF16_STORED_OUT) && conv_bf16_ok(g)) {
        // the engine's mixed-precision layers: activation and filter live in HBM as bfloat16
        const size_t nx = (size_t)n * h * wd * cin, nw = (size_t)kh * kw * cin * cout, ny = (size_t)n * g.Ho * g.Wo * cout;
        uint16_t* xb = sc.alloc<uint16_t>(nx);
        uint16_t* wb = sc.alloc<uint16_t>(2 * nw);      // both layouts of conv_weights_bf16
        if (!sc.ok) return L3_ENOMEM;
        cast_bf16(dx, xb, (int64_t)nx, sc.s);
        conv_weights_bf16(dw, wb, kh, kw, cin, cout, true, sc.s);
        if (dtype == L3_OP_BF16_STORED_OUT) {
            // ... and so does the output (read back widened: every value returned is a bfloat16)
            uint16_t* yb = sc.alloc<uint16_t>(ny);
            if (!sc.ok) return L3_ENOMEM;
            conv_bf16_fwd(reinterpret_cast<const float*>(xb), reinterpret_cast<const float*>(wb), db,
                          reinterpret_cast<float*>(yb), g, sc.s, true, nullptr, 0, true);
            std::vector<uint16_t> hy(ny);
            sc.get(hy.data(), yb, ny);
            for (size_t i = 0; i < ny; ++i) {
                const uint32_t u = (uint32_t)hy[i] << 16;
                memcpy(y + i, &u, 4);
            }
            return sc.status();
        }
        conv_bf16_fwd(reinterpret_cast<const float*>(xb), reinterpret_cast<const float*>(wb), db, dy, g, sc.s, true);
    } else if (dtype != L3_DTYPE_F32 && conv_bf16_ok(g)) {
        float* dwn = sc.alloc<float>((size_t)kh * kw * cin * cout);
        if (!sc.ok) return L3_ENOMEM;
        conv_flip_weights(dw, dwn, kh, kw, cin, cout, sc.s);
        conv_bf16_fwd(dx, dwn, db, dy, g, sc.s);
    } else if (conv_first_ok(g)) {
        if (dtype == L3_OP_BF16_STORED_OUT) {                 // first conv of a tower in a bf16 engine: fp32 math, bf16 store
            const size_t ny = (size_t)n * g.Ho * g.Wo * cout;
            uint16_t* yb = sc.alloc<uint16_t>(ny);
            if (!sc.ok) return L3_ENOMEM;
            conv_first_fwd(dx, dw, db, yb, g, sc.s, nullptr, 0, true);
            std::vector<uint16_t> hy(ny);
            sc.get(hy.data(), yb, ny);
            for (size_t i = 0; i < ny; ++i) {
                const uint32_t u = (uint32_t)hy[i] << 16;
                memcpy(y + i, &u, 4);
            }
            return sc.status();
        }
        conv_first_fwd(dx, dw, db, dy, g, sc.s);
    } else {
        float* du = nullptr;
        if (conv_wino_floats(g)) {
            du = sc.alloc<float>(conv_wino_floats(g));
            if (!sc.ok) return L3_ENOMEM;
            conv_wino_transform_weights(dw, du, g, false, sc.s);
        }
        conv_fwd(dx, dw, db, dy, g, sc.s, du);
    }
    sc.get(y, dy, (size_t)n * g.Ho * g.Wo * cout);
    return sc.status();
}

int l3_op_conv2d_fwd(int device, const float* x, const float* w, const float* b, float* y, int n, int h, int wd,
                     int cin, int cout, int kh, int kw, int same) {
    return l3_op_conv2d_fwd_dt(device, L3_DTYPE_F32, x, w, b, y, n, h, wd, cin, cout, kh, kw, same);
}

int l3_op_conv2d_bwd_dt(int device, int dtype, const float* x, const float* w, const float* dy, float* dx, float* dw,
                        float* db, int n, int h, int wd, int cin, int cout, int kh, int kw, int same) {
    Scope sc(device);
    if (!sc.ok) return L3_EHIP;
    const ConvGeom g = make_geom(n, h, wd, cin, cout, kh, kw, same);
    const size_t nx = (size_t)n * h * wd * cin, ny = (size_t)n * g.Ho * g.Wo * cout, nw = (size_t)kh * kw * cin * cout;
    float* d_x = sc.put(x, nx);
    float* d_w = sc.put(w, nw);
    float* d_dy = sc.put(dy, ny);
    float* d_dx = sc.alloc<float>(nx);
    float* d_dw = sc.alloc<float>(nw);
    float* d_db = sc.alloc<float>((size_t)cout);
    float* d_wf = sc.alloc<float>(nw);
    float* d_part = sc.alloc<float>(conv_wgrad_scratch_floats(g));
    float* d_red = sc.alloc<float>(colreduce_scratch_floats((int64_t)n * g.Ho * g.Wo, cout));
    if (!sc.ok) return L3_ENOMEM;
    const bool mp = dtype != L3_DTYPE_F32;
    ConvGeom dg{n, g.Ho, g.Wo, cout, h, wd, cin, kh, kw, kh - 1 - g.padT, kw - 1 - g.padL};
    dg.solo = 1;
    dg.f2x2 = g.f2x2;
    if ((dtype == L3_OP_BF16_STORED || dtype == L3_OP_BF16_STORED_OUT) && conv_wgrad_bf16_ok(g) && conv_bf16_ok(dg)) {
        // bfloat16-stored operands, as the engine keeps them for its mixed-precision layers; the bias
        // gradient stays a plain fp32 column sum of the unrounded dy
        uint16_t* xb = sc.alloc<uint16_t>(nx);
        uint16_t* gb = sc.alloc<uint16_t>(ny);
        uint16_t* wb = sc.alloc<uint16_t>(2 * nw);      // both layouts of conv_weights_bf16
        if (!sc.ok) return L3_ENOMEM;
        cast_bf16(d_x, xb, (int64_t)nx, sc.s);
        cast_bf16(d_dy, gb, (int64_t)ny, sc.s);
        conv_wgrad(reinterpret_cast<const float*>(xb), reinterpret_cast<const float*>(gb), d_dw, d_part, g, sc.s, true, true);
        colsum(d_dy, d_db, d_red, (int64_t)n * g.Ho * g.Wo, cout, sc.s);
        conv_weights_bf16(d_w, wb, kh, kw, cin, cout, false, sc.s);
        if (dtype == L3_OP_BF16_STORED_OUT) {      // the data gradient is stored as bfloat16 too (returned widened)
            uint16_t* dxb = sc.alloc<uint16_t>(nx);
            if (!sc.ok) return L3_ENOMEM;
            conv_bf16_fwd(reinterpret_cast<const float*>(gb), reinterpret_cast<const float*>(wb), nullptr,
                          reinterpret_cast<float*>(dxb), dg, sc.s, true, nullptr, 0, true);
            std::vector<uint16_t> hx(nx);
            sc.get(hx.data(), dxb, nx);
            for (size_t i = 0; i < nx; ++i) {
                const uint32_t u = (uint32_t)hx[i] << 16;
                memcpy(dx + i, &u, 4);
            }
        } else {
            conv_bf16_fwd(reinterpret_cast<const float*>(gb), reinterpret_cast<const float*>(wb), nullptr, d_dx, dg, sc.s, true);
            sc.get(dx, d_dx, nx);
        }
        sc.get(dw, d_dw, nw);
        sc.get(db, d_db, (size_t)cout);
        return sc.status();
    }
    conv_wgrad(d_x, d_dy, d_dw, d_part, g, sc.s, mp && conv_wgrad_bf16_ok(g));
    colsum(d_dy, d_db, d_red, (int64_t)n * g.Ho * g.Wo, cout, sc.s);
    if (!conv_dgrad_small(d_dy, d_w, d_dx, g, sc.s)) {
        if (mp && conv_bf16_ok(dg)) {
            conv_bf16_fwd(d_dy, d_w, nullptr, d_dx, dg, sc.s);      // the forward filter is the dgrad's [flip][n][k]
        } else {
            float* d_u = nullptr;
            if (conv_wino_floats(dg)) {
                d_u = sc.alloc<float>(conv_wino_floats(dg));
                if (!sc.ok) return L3_ENOMEM;
                conv_wino_transform_weights(d_w, d_u, dg, true, sc.s);
            }
            conv_flip_weights(d_w, d_wf, kh, kw, cin, cout, sc.s);
            conv_fwd(d_dy, d_wf, nullptr, d_dx, dg, sc.s, d_u);
        }
    }
    sc.get(dx, d_dx, nx);
    sc.get(dw, d_dw, nw);
    sc.get(db, d_db, (size_t)cout);
    return sc.status();
}

int l3_op_conv2d_bwd(int device, const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                     int n, int h, int wd, int cin, int cout, int kh, int kw, int same) {
    return l3_op_conv2d_bwd_dt(device, L3_DTYPE_F32, x, w, dy, dx, dw, db, n, h, wd, cin, cout, kh, kw, same);
}

int l3_op_bn_relu_fwd(int device, const float* x, const float* gamma, const float* beta, float* y, float* mean,
                      float* var, int64_t rows, int c, int relu, int x_bf16) {
    Scope sc(device);
    if (!sc.ok) return L3_EHIP;
    if ((x_bf16 & 1) && !bn_fast_ok(c)) return L3_EINVAL;
    const size_t n = (size_t)rows * c, cp = (size_t)(c + 3) / 4 * 4;
    float* d_x = sc.put(x, n);
    float* d_g = sc.alloc<float>(cp);
    float* d_b = sc.alloc<float>(cp);
    float *d_m = sc.alloc<float>(cp), *d_v = sc.alloc<float>(cp), *d_sc = sc.alloc<float>(cp), *d_sh = sc.alloc<float>(cp);
    float* d_y = sc.alloc<float>(n);
    float* d_red = sc.alloc<float>(colreduce_scratch_floats(rows, c));
    if (!sc.ok) return L3_ENOMEM;
    (void)hipMemcpy(d_g, gamma, c * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_b, beta, c * 4, hipMemcpyHostToDevice);
    if (x_bf16 & 1) {  // x as a mixed-precision conv leaves it: bfloat16 in HBM
        uint16_t* xb = sc.alloc<uint16_t>(n);
        if (!sc.ok) return L3_ENOMEM;
        cast_bf16(d_x, xb, (int64_t)n, sc.s);
        bn_stats_fast(reinterpret_cast<const float*>(xb), d_g, d_b, d_m, d_v, d_sc, d_sh, d_red, rows, c, 1e-3f, 0, sc.s, 1);
        bn_apply_fast(reinterpret_cast<const float*>(xb), d_sc, d_sh, d_y, rows, c, relu, sc.s, 0, 1);
    } else {
        bn_stats(d_x, d_g, d_b, d_m, d_v, d_sc, d_sh, d_red, rows, c, 1e-3f, sc.s);
        bn_apply(d_x, d_sc, d_sh, d_y, rows, c, relu, sc.s);
    }
    sc.get(y, d_y, n);
    sc.get(mean, d_m, (size_t)c);
    sc.get(var, d_v, (size_t)c);
    return sc.status();
}

int l3_op_bn_relu_bwd(int device, const float* x, const float* y, const float* dy, const float* gamma,
                      const float* beta, const float* mean, const float* var, float* dx, float* dgamma, float* dbeta,
                      int64_t rows, int c, int relu, int x_bf16) {
    Scope sc(device);
    if (!sc.ok) return L3_EHIP;
    const size_t n = (size_t)rows * c, cp = (size_t)(c + 3) / 4 * 4;
    float* d_x = sc.put(x, n);
    float* d_y = sc.put(y, n);
    float* d_dy = sc.put(dy, n);
    float *d_g = sc.alloc<float>(cp), *d_m = sc.alloc<float>(cp), *d_v = sc.alloc<float>(cp);
    float *d_dg = sc.alloc<float>(cp), *d_db = sc.alloc<float>(cp);
    float* d_dx = sc.alloc<float>(n);
    const bool fast = beta != nullptr && bn_fast_ok(c) && rows < (int64_t)1 << 30;
    if (x_bf16 && !fast) return L3_EINVAL;
    size_t red = colreduce_scratch_floats(rows, c);
    if (fast && bn_fast_scratch_floats(c) > red) red = bn_fast_scratch_floats(c);
    float* d_red = sc.alloc<float>(red);
    if (!sc.ok) return L3_ENOMEM;
    (void)hipMemcpy(d_g, gamma, c * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_m, mean, c * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_v, var, c * 4, hipMemcpyHostToDevice);
    if (fast) {
        // the engine's dispatch for power-of-two channel counts: ReLU mask recomputed from x*scale+shift
        float *d_b = sc.put(beta, (size_t)c), *d_sc = sc.alloc<float>(cp), *d_sh = sc.alloc<float>(cp);
        if (!sc.ok) return L3_ENOMEM;
        bn_scale_shift(d_g, d_b, d_m, d_v, d_sc, d_sh, c, 1e-3f, sc.s);
        const float* xin = d_x;
        const float* dyin = d_dy;
        if (x_bf16 & 1) {
            uint16_t* xb = sc.alloc<uint16_t>(n);
            if (!sc.ok) return L3_ENOMEM;
            cast_bf16(d_x, xb, (int64_t)n, sc.s);
            xin = reinterpret_cast<const float*>(xb);
        }
        if (x_bf16 & 2) {       // dy as the mixed-precision data-gradient kernel leaves it: bfloat16 in HBM
            uint16_t* gb = sc.alloc<uint16_t>(n);
            if (!sc.ok) return L3_ENOMEM;
            cast_bf16(d_dy, gb, (int64_t)n, sc.s);
            dyin = reinterpret_cast<const float*>(gb);
        }
        bn_bwd_fast(xin, d_sc, d_sh, d_m, d_v, d_g, dyin, 0, 1, 1, (int)rows, c, 1, (int)rows, (int64_t)rows * c, d_dx, d_dg,
                    d_db, nullptr, d_red, 1e-3f, relu, 1, sc.s, 0, (x_bf16 & 1) ? 1 : 0, (x_bf16 & 2) ? 1 : 0);
    } else {
        bn_bwd(d_x, d_y, d_dy, d_g, d_m, d_v, d_dx, d_dg, d_db, d_red, rows, c, 1e-3f, relu, 1, sc.s);
    }
    sc.get(dx, d_dx, n);
    sc.get(dgamma, d_dg, (size_t)c);
    sc.get(dbeta, d_db, (size_t)c);
    return sc.status();
}

static int pool2_common(int device, const float* x, const float* gamma, const float* beta, const float* dp, float* p,
                        float* mean, float* var, float* dx, float* dgamma, float* dbeta, float* dbias, int n, int h,
                        int wd, int c, int same, int mode, int x_bf16) {
    if (!bn_fast_ok(c) || (mode != 1 && mode != 2)) return L3_EINVAL;
    Scope sc(device);
    if (!sc.ok) return L3_EHIP;
    const PoolGeom g = make_pool(n, h, wd, c, 2, 2, 2, 2, same);
    const size_t nx = (size_t)n * h * wd * c, np_ = (size_t)n * g.Ho * g.Wo * c;
    float* d_x = sc.put(x, nx);
    float* d_g = sc.put(gamma, (size_t)c);
    float* d_b = sc.put(beta, (size_t)c);
    float *d_m = sc.alloc<float>(c), *d_v = sc.alloc<float>(c), *d_sc = sc.alloc<float>(c), *d_sh = sc.alloc<float>(c);
    float* d_p = sc.alloc<float>(np_);
    float* d_red = sc.alloc<float>(colreduce_scratch_floats((int64_t)n * h * wd, c));
    if (!sc.ok) return L3_ENOMEM;
    if (x_bf16 & 1) {  // x as a mixed-precision conv leaves it: bfloat16 in HBM
        uint16_t* xb = sc.alloc<uint16_t>(nx);
        if (!sc.ok) return L3_ENOMEM;
        cast_bf16(d_x, xb, (int64_t)nx, sc.s);
        d_x = reinterpret_cast<float*>(xb);
    }
    if (mode == 2 || (x_bf16 & 1))      // mode 2 = ReLU -> BN (vision_model.py:138-139): moments of relu(x)
        bn_stats_fast(d_x, d_g, d_b, d_m, d_v, d_sc, d_sh, d_red, (int64_t)n * h * wd, c, 1e-3f, mode == 2 ? 1 : 0, sc.s, x_bf16 & 1);
    else
        bn_stats(d_x, d_g, d_b, d_m, d_v, d_sc, d_sh, d_red, (int64_t)n * h * wd, c, 1e-3f, sc.s);
    bn_relu_pool2_fwd(d_x, d_sc, d_sh, d_p, n, h, wd, c, g.Ho, g.Wo, g.out_batch_stride, mode, sc.s, 0, x_bf16 & 1);
    sc.get(p, d_p, np_);
    sc.get(mean, d_m, (size_t)c);
    sc.get(var, d_v, (size_t)c);
    if (dp) {
        float* d_dp = sc.put(dp, np_);
        if (x_bf16 & 2) {   // the pooled gradient as the mixed-precision data-gradient kernel leaves it: bfloat16 in HBM
            uint16_t* gb = sc.alloc<uint16_t>(np_);
            if (!sc.ok) return L3_ENOMEM;
            cast_bf16(d_dp, gb, (int64_t)np_, sc.s);
            d_dp = reinterpret_cast<float*>(gb);
        }
        float* d_dx = sc.alloc<float>(nx);
        float *d_dg = sc.alloc<float>(c), *d_db = sc.alloc<float>(c), *d_dbias = sc.alloc<float>(c);
        if (!sc.ok) return L3_ENOMEM;
        bn_bwd_fast(d_x, d_sc, d_sh, d_m, d_v, d_g, d_dp, 1, n, h, wd, c, g.Ho, g.Wo, g.out_batch_stride, d_dx, d_dg,
                    d_db, d_dbias, d_red, 1e-3f, mode, 1, sc.s, 0, x_bf16 & 1, (x_bf16 & 2) ? 1 : 0);
        sc.get(dx, d_dx, nx);
        sc.get(dgamma, d_dg, (size_t)c);
        sc.get(dbeta, d_db, (size_t)c);
        sc.get(dbias, d_dbias, (size_t)c);
    }
    return sc.status();
}

int l3_op_bn_relu_pool2_fwd(int device, const float* x, const float* gamma, const float* beta, float* p, float* mean,
                            float* var, int n, int h, int wd, int c, int same, int relu_mode, int x_bf16) {
    return pool2_common(device, x, gamma, beta, nullptr, p, mean, var, nullptr, nullptr, nullptr, nullptr, n, h, wd, c,
                        same, relu_mode, x_bf16);
}

int l3_op_bn_relu_pool2_bwd(int device, const float* x, const float* gamma, const float* beta, const float* dp,
                            float* dx, float* dgamma, float* dbeta, float* dbias, int n, int h, int wd, int c, int same,
                            int relu_mode, int x_bf16) {
    return pool2_common(device, x, gamma, beta, dp, nullptr, nullptr, nullptr, dx, dgamma, dbeta, dbias, n, h, wd, c,
                        same, relu_mode, x_bf16);
}

int l3_op_maxpool_fwd(int device, const float* x, float* y, int n, int h, int wd, int c, int ph, int pw, int sh,
                      int sw, int same) {
    Scope sc(device);
    if (!sc.ok) return L3_EHIP;
    const PoolGeom g = make_pool(n, h, wd, c, ph, pw, sh, sw, same);
    float* d_x = sc.put(x, (size_t)n * h * wd * c);
    float* d_y = sc.alloc<float>((size_t)n * g.Ho * g.Wo * c);
    if (!sc.ok) return L3_ENOMEM;
    maxpool_fwd(d_x, d_y, g, sc.s);
    sc.get(y, d_y, (size_t)n * g.Ho * g.Wo * c);
    return sc.status();
}

int l3_op_maxpool_bwd(int device, const float* x, const float* dy, float* dx, int n, int h, int wd, int c, int ph,
                      int pw, int sh, int sw, int same) {
    Scope sc(device);
    if (!sc.ok) return L3_EHIP;
    if (sh < ph || sw < pw) return L3_EINVAL;
    const PoolGeom g = make_pool(n, h, wd, c, ph, pw, sh, sw, same);
    float* d_x = sc.put(x, (size_t)n * h * wd * c);
    float* d_dy = sc.put(dy, (size_t)n * g.Ho * g.Wo * c);
    float* d_dx = sc.alloc<float>((size_t)n * h * wd * c);
    if (!sc.ok) return L3_ENOMEM;
    maxpool_bwd(d_x, d_dy, d_dx, g, sc.s);
    sc.get(dx, d_dx, (size_t)n * h * wd * c);
    return sc.status();
}

// BatchNormalization batch moments from the partial sums a convolution epilogue leaves (`nblk` rows of [sum, sum of squares][c]
// about `pivot`): the engine's stage 2 (bn_fused.hip launch_fast_final, its fp64 pre-reduction above 2048 rows included) on a
// buffer of EXACTLY nblk rows followed by a guard region, which must come back untouched (L3_EINVAL otherwise).
int l3_op_bn_stats_from_partials(int device, const float* part, int nblk, int c, const float* pivot, int64_t rows, float eps,
                                 float* mean, float* var) {
    Scope sc(device);
    if (!sc.ok) return L3_EHIP;
    if (nblk < 1 || c < 4 || c % 4) return L3_EINVAL;
    const size_t n = (size_t)nblk * 2 * c, guard = (size_t)8 * c;
    std::vector<float> host(n + guard);
    memcpy(host.data(), part, n * sizeof(float));
    for (size_t i = 0; i < guard; ++i) host[n + i] = -12345.f;
    float* d_part = sc.put(host.data(), n + guard);
    float* d_pivot = sc.put(pivot, (size_t)c);
    std::vector<float> ones((size_t)c, 1.f), zeros((size_t)c, 0.f);
    float* d_gamma = sc.put(ones.data(), (size_t)c);
    float* d_beta = sc.put(zeros.data(), (size_t)c);
    float* d_mean = sc.alloc<float>((size_t)c);
    float* d_var = sc.alloc<float>((size_t)c);
    float* d_scale = sc.alloc<float>((size_t)c);
    float* d_shift = sc.alloc<float>((size_t)c);
    if (!sc.ok) return L3_ENOMEM;
    bn_stats_from_partials(d_part, nblk, d_pivot, d_gamma, d_beta, d_mean, d_var, d_scale, d_shift, rows, c, eps, 0, sc.s);
    sc.get(mean, d_mean, (size_t)c);
    sc.get(var, d_var, (size_t)c);
    sc.get(host.data() + n, d_part + n, guard);
    for (size_t i = 0; i < guard; ++i)
        if (host[n + i] != -12345.f) return L3_EINVAL;
    return sc.status();
}

int l3_op_preprocess(int device, const uint8_t* video_u8, int64_t nv, float* video, const int16_t* audio_i16,
                     int64_t na, float* audio) {
    Scope sc(device);
    if (!sc.ok) return L3_EHIP;
    if (video_u8 && nv > 0) {
        uint8_t* d_in = sc.put(video_u8, (size_t)nv);
        float* d_out = sc.alloc<float>((size_t)nv);
        if (!sc.ok) return L3_ENOMEM;
        preprocess_video(d_in, d_out, nv, sc.s);
        sc.get(video, d_out, (size_t)nv);
    }
    if (audio_i16 && na > 0) {
        int16_t* d_in = sc.put(audio_i16, (size_t)na);
        float* d_out = sc.alloc<float>((size_t)na);
        if (!sc.ok) return L3_ENOMEM;
        preprocess_audio(d_in, d_out, na, sc.s);
        sc.get(audio, d_out, (size_t)na);
    }
    return sc.status();
}

int l3_op_frontend(int device, int model_type, const float* audio, int n, int db_max_scope, float* out) {
    // runs the engine's own front-end path on a throw-away engine of batch n
    l3_config cfg{};
    cfg.struct_size = (int32_t)sizeof(l3_config);
    cfg.model_type = model_type;
    cfg.batch = n;
    cfg.device = device;
    cfg.db_max_scope = db_max_scope;
    cfg.bn_zero_debias = 1;
    l3_engine* e = nullptr;
    int rc = l3_create(&cfg, 1, &e);
    if (rc) return rc;
    rc = l3_upload_batch(e, nullptr, audio, nullptr);
    if (!rc) rc = l3_step_forward(e, 0);
    int64_t numel = 0;
    if (!rc) rc = l3_activation_numel(e, "audio_model/frontend", &numel);
    if (!rc) rc = l3_get_activation(e, "audio_model/frontend", out, numel);
    l3_destroy(e);
    return rc;
}

}  // extern "C"
