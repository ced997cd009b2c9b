// The reference's in-tree AESRGAN - RRDB trunk with full-image self-attention blocks, the network of AESRGANFaceRestorer - as ONE
// engine behind the C-ABI: weights, workspace arena and launch sequencing (reference: processors/aesrgan_face.py: AttentionBlock
// :142-168, ResidualDenseBlock / RRDB :171-204, AESRGAN :205-269 - no pixel-unshuffle front end, an AttentionBlock behind every
// (num_block / num_attention)-th RRDB, conv_up2 only for scale >= 4).  Round 1 sequenced the ~360 convolutions and the attention
// blocks from Python over the building-block entries; this file issues the same launches from C++ under one mutex:
//
//   every 3x3 convolution on conv3x3_mfma.hip (chunk-planar typed activations in a six-plane concat buffer, the residual streams in
//   fp32, `x5 * 0.2 + x` and the RRDB's second residual in the conv5 epilogue); the 1x1 query / key / value projections on the
//   pointwise GEMM; softmax(q^T k) over all pixels as a row kernel (fw_attn_softmax_rows); the product with v as a GEMM over the
//   pixel axis (fw_pack_pointwise_transposed + the pointwise GEMM with the gamma residual epilogue).
// Parity: oracle/rrdbnet_ref.py aesrgan_forward, pinned on vectors the reference's own module produced
// (tests/golden/aesrgan_attention.npz).  Face crops are small: no pair fusion, no split trunk here (csrc/rrdbnet.hip has those for
// the 1080p Real-ESRGAN path).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>
#include "fw_internal.h"
#include "../../include/framewright_hip.h"

using namespace fw;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct Conv {
    DevBuf w, b;
    int ct = 0, chunks = 0;
};
struct Attn {
    DevBuf w[3], b[3], gamma;   // query, key, value
    int tiles[3] = {0, 0, 0};
};

struct Arena {
    char* base = nullptr;
    size_t top = 0, peak = 0;
    bool plan = false;
    void* take(size_t bytes) {
        const size_t at = top;
        top += (bytes + 255) / 256 * 256;
        if (top > peak) peak = top;
        return base + at;
    }
};

}  // namespace

struct fw_aesrgan {
    int device = 0;
    fw::StreamOrder order;   // device-side ordering of forwards enqueued on different streams (fw_internal.h)
    DType dt = DT_F16;
    int num_block = 23, scale = 2, num_attention = 4;
    std::mutex mu;
    std::set<int> attn_after;
    std::map<std::string, size_t> want;
    std::map<std::string, std::vector<float>> host;
    std::map<std::string, Conv> convs;
    std::map<int, Attn> attn;
    bool built = false;
    DevBuf ws;
};

namespace {

int fail(int code, const std::string& m) {
    fw::last_error_ref() = m;
    return code;
}
template <typename F>
int guarded(F&& f) {
    try {
        f();
        return FW_OK;
    } catch (const fw::Error& e) {
        return fail(e.code, e.what());
    } catch (const std::bad_alloc&) {
        return fail(FW_ERR_OOM, "host out of memory");
    } catch (const std::exception& e) {
        return fail(FW_ERR_INTERNAL, e.what());
    }
}
void upload(DevBuf& b, const void* src, size_t bytes) {
    b.release();
    FW_HIP_CHECK(hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    FW_HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
}
void chk(int status) {
    if (status != FW_OK) throw Error(status, fw::last_error_ref());
}

std::vector<std::string> conv_names(const fw_aesrgan* n) {
    std::vector<std::string> v = {"conv_first", "conv_body", "conv_up1", "conv_hr", "conv_last"};
    if (n->scale >= 4) v.push_back("conv_up2");
    for (int i = 0; i < n->num_block; ++i)
        for (int r = 1; r <= 3; ++r)
            for (int c = 1; c <= 5; ++c) v.push_back("body." + std::to_string(i) + ".rdb" + std::to_string(r) + ".conv" + std::to_string(c));
    return v;
}
void conv_shape(const std::string& name, int* cout, int* cin) {
    if (name == "conv_first") { *cout = 64; *cin = 3; return; }
    if (name == "conv_last") { *cout = 3; *cin = 64; return; }
    if (name.rfind("body.", 0) == 0) {
        const int c = name.back() - '0';
        *cin = 64 + 32 * (c - 1);
        *cout = c == 5 ? 64 : 32;
        return;
    }
    *cout = 64;
    *cin = 64;
}

// x: fp32 [H*W][3] RGB in [0, 1]; out: fp32 [sH*sW][3], un-clamped (what AESRGAN.forward returns, NHWC)
void forward(fw_aesrgan* n, Arena& A, const float* x, int H, int W, float* out, hipStream_t st_) {
    void* st = (void*)st_;
    const int dt = (int)n->dt;
    const bool run = !A.plan;
    const long M = (long)H * W, PL = M * 32;
    auto f32 = [&](size_t elems) { return (float*)A.take(elems * 4); };
    auto typ = [&](size_t elems) { return (char*)A.take(elems * 2); };
#define RUN(expr)            \
    do {                     \
        if (run) chk(expr);  \
    } while (0)
    auto conv = [&](const std::string& name, const void* src, long src_pstride, int h, int w, void* o, long out_pstride, int act, int ups,
                    const float* res1, float s1, const float* res2, float s2, float* o32) {
        const Conv& cv = n->convs.at(name);
        RUN(fw_conv3x3_nhwc(dt, src, 32, src_pstride, cv.chunks, h, w, cv.w.p, (const float*)cv.b.p, cv.ct, act, ups, res1, s1, res2, s2, o, 32,
                            out_pstride, 0, o32, st));
    };

    char* img = typ((size_t)M * 32);
    if (run) launch_rgb_f32_to_nhwc(n->dt, x, M, img, st_);
    char* cat[2] = {typ((size_t)6 * PL), typ((size_t)6 * PL)};
    float* feat = f32((size_t)M * 64);
    float* pool[4] = {f32((size_t)M * 64), f32((size_t)M * 64), f32((size_t)M * 64), f32((size_t)M * 64)};
    auto free_buf = [&](const float* a, const float* b) {
        for (float* t : pool)
            if (t != a && t != b) return t;
        return pool[0];
    };

    conv("conv_first", img, 32, H, W, cat[0], PL, 0, 0, nullptr, 1.f, nullptr, 1.f, feat);
    const float* cur_f = feat;
    int cur = 0;
    for (int i = 0; i < n->num_block; ++i) {
        const float* rrdb_in = cur_f;
        const float* x_f = cur_f;
        for (int r = 1; r <= 3; ++r) {
            const std::string pre = "body." + std::to_string(i) + ".rdb" + std::to_string(r) + ".conv";
            char* X = cat[cur];
            for (int c = 1; c <= 4; ++c)   // x1..x4 -> planes 2..5 (aesrgan_face.py:184-187)
                conv(pre + std::to_string(c), X, PL, H, W, X + (size_t)(1 + c) * PL * 2, 0, 1, 0, nullptr, 1.f, nullptr, 1.f, nullptr);
            float* y_f = free_buf(x_f, rrdb_in);
            if (r < 3)                     // x5 * 0.2 + x  (:188-189)
                conv(pre + "5", X, PL, H, W, cat[1 - cur], PL, 0, 0, x_f, 0.2f, nullptr, 1.f, y_f);
            else                           // ... and the RRDB's own residual: * 0.2 + rrdb_in (:204)
                conv(pre + "5", X, PL, H, W, cat[1 - cur], PL, 0, 0, x_f, 0.2f, rrdb_in, 0.2f, y_f);
            x_f = y_f;
            cur = 1 - cur;
        }
        cur_f = x_f;
        if (n->attn_after.count(i)) {     // AttentionBlock behind this RRDB (:229-233): gamma * (v softmax(q^T k)^T) + x
            const Attn& a = n->attn.at(i);
            const size_t mark = A.top;
            char* qkv[3];
            for (int k = 0; k < 3; ++k) {
                qkv[k] = typ((size_t)M * 32 * a.tiles[k]);
                RUN(fw_pointwise_nhwc(dt, cur_f, 1, 64, M, 64, a.w[k].p, (const float*)a.b[k].p, a.tiles[k], qkv[k], 32 * a.tiles[k], nullptr, 0, nullptr,
                                      nullptr, st));
            }
            const long kp = (M + 31) / 32 * 32;
            char* P = typ((size_t)M * kp);
            RUN(fw_attn_softmax_rows(dt, qkv[0], 32, qkv[1], 32, M, 8, P, kp, st));
            char* vt = typ(fw_pack_pointwise(dt, nullptr, 64, (int)kp, nullptr));
            RUN(fw_pack_pointwise_transposed(dt, qkv[2], 64, M, 64, (int)kp, vt, st));
            float* y = free_buf(cur_f, nullptr);
            RUN(fw_pointwise_nhwc(dt, P, 0, kp, M, (int)kp, vt, nullptr, 2, nullptr, 0, y, 64, cur_f, (const float*)a.gamma.p, st));
            cur_f = y;
            RUN(fw_f32_to_planar(dt, cur_f, M, 64, cat[cur], st));
            A.top = mark;
        }
    }
    // feat + conv_body(body)  (:256-257)
    char* body = typ((size_t)2 * PL);
    conv("conv_body", cat[cur], PL, H, W, body, PL, 0, 0, feat, 1.f, nullptr, 1.f, nullptr);
    const int h2 = 2 * H, w2 = 2 * W;
    char* u1 = typ((size_t)2 * h2 * w2 * 32);
    conv("conv_up1", body, PL, h2, w2, u1, (long)h2 * w2 * 32, 1, 1, nullptr, 1.f, nullptr, 1.f, nullptr);   // nearest x2 + conv + lrelu
    char* top = u1;
    int ht = h2, wt = w2;
    if (n->scale >= 4) {
        const int h4 = 2 * h2, w4 = 2 * w2;
        char* u2 = typ((size_t)2 * h4 * w4 * 32);
        conv("conv_up2", u1, (long)h2 * w2 * 32, h4, w4, u2, (long)h4 * w4 * 32, 1, 1, nullptr, 1.f, nullptr, 1.f, nullptr);
        top = u2;
        ht = h4;
        wt = w4;
    }
    char* hr = typ((size_t)2 * ht * wt * 32);
    conv("conv_hr", top, (long)ht * wt * 32, ht, wt, hr, (long)ht * wt * 32, 1, 0, nullptr, 1.f, nullptr, 1.f, nullptr);
    float* last = f32((size_t)ht * wt * 32);
    conv("conv_last", hr, (long)ht * wt * 32, ht, wt, nullptr, 0, 0, 0, nullptr, 1.f, nullptr, 1.f, last);
    if (run) launch_take_rgb_f32(last, 32, (long)ht * wt, out, st_);
#undef RUN
}

}  // namespace

extern "C" {

int fw_aesrgan_create(int device_id, int num_block, int scale, int num_attention, int dtype, fw_aesrgan** out) {
    if (!out) return fail(FW_ERR_INVALID, "fw_aesrgan_create: NULL argument");
    *out = nullptr;
    if ((scale != 2 && scale != 4) || num_block < 1 || num_block > 64 || num_attention < 1 || num_attention > num_block)
        return fail(FW_ERR_INVALID, "fw_aesrgan_create: scale 2 or 4, 1 <= num_attention <= num_block <= 64");
    if (dtype != FW_DTYPE_BF16 && dtype != FW_DTYPE_F16) return fail(FW_ERR_INVALID, "fw_aesrgan_create: bad dtype");
    return guarded([&] {
        int nd = 0;
        FW_HIP_CHECK(hipGetDeviceCount(&nd));
        if (device_id < 0 || device_id >= nd) throw Error(FW_ERR_INVALID, "fw_aesrgan_create: no such device");
        auto n = std::make_unique<fw_aesrgan>();
        n->device = device_id;
        n->dt = (DType)dtype;
        n->num_block = num_block;
        n->scale = scale;
        n->num_attention = num_attention;
        for (int i = 0; i < num_block; i += num_block / num_attention) n->attn_after.insert(i);   // aesrgan_face.py:229
        for (const auto& name : conv_names(n.get())) {
            int cout, cin;
            conv_shape(name, &cout, &cin);
            n->want[name + ".weight"] = (size_t)cout * cin * 9;
            n->want[name + ".bias"] = (size_t)cout;
        }
        for (int i : n->attn_after) {
            const std::string p = "attn." + std::to_string(i) + ".";
            n->want[p + "query.weight"] = 8 * 64;
            n->want[p + "query.bias"] = 8;
            n->want[p + "key.weight"] = 8 * 64;
            n->want[p + "key.bias"] = 8;
            n->want[p + "value.weight"] = 64 * 64;
            n->want[p + "value.bias"] = 64;
            n->want[p + "gamma"] = 1;
        }
        *out = n.release();
    });
}

// keys: BasicSR's names for the trunk / tail (conv_first, body.{i}.rdb{1,2,3}.conv{1..5}, conv_body, conv_up1 [, conv_up2], conv_hr,
// conv_last: .weight [cout][cin][3][3], .bias) and attn.{i}.query|key|value.weight ([8|8|64][64], the 1x1 convs) / .bias, attn.{i}.gamma
int fw_aesrgan_set_tensor(fw_aesrgan* n, const char* key_c, const float* data, size_t numel) {
    if (!n || !key_c || !data) return fail(FW_ERR_INVALID, "fw_aesrgan_set_tensor: NULL argument");
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        const std::string key(key_c);
        auto it = n->want.find(key);
        if (it == n->want.end()) throw Error(FW_ERR_INVALID, "fw_aesrgan_set_tensor: unknown tensor '" + key + "'");
        if (numel != it->second)
            throw Error(FW_ERR_INVALID, "fw_aesrgan_set_tensor: '" + key + "' has " + std::to_string(numel) + " elements, expected " + std::to_string(it->second));
        n->host[key].assign(data, data + numel);
        n->built = false;
    });
}

int fw_aesrgan_finalize(fw_aesrgan* n) {
    if (!n) return fail(FW_ERR_INVALID, "fw_aesrgan_finalize: NULL");
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        if (n->built) return;
        for (const auto& kv : n->want)
            if (!n->host.count(kv.first)) throw Error(FW_ERR_INVALID, "fw_aesrgan_finalize: missing " + kv.first);
        DevGuard dg(n->device);
        FW_HIP_CHECK(hipDeviceSynchronize());
        const DType dt = n->dt;
        for (const auto& name : conv_names(n)) {
            int cout, cin;
            conv_shape(name, &cout, &cin);
            Conv& cv = n->convs[name];
            cv.ct = (cout + 31) / 32;
            cv.chunks = (cin + 31) / 32;
            std::vector<uint16_t> pk(pack_conv3x3_weights(dt, nullptr, cout, cin, cv.ct, cv.chunks, nullptr));
            pack_conv3x3_weights(dt, n->host.at(name + ".weight").data(), cout, cin, cv.ct, cv.chunks, pk.data());
            upload(cv.w, pk.data(), pk.size() * 2);
            std::vector<float> bp((size_t)32 * cv.ct, 0.f);
            const auto& b = n->host.at(name + ".bias");
            for (int i = 0; i < cout; ++i) bp[i] = b[i];
            upload(cv.b, bp.data(), bp.size() * 4);
        }
        const char* names[3] = {"query", "key", "value"};
        for (int i : n->attn_after) {
            Attn& a = n->attn[i];
            const std::string p = "attn." + std::to_string(i) + ".";
            for (int k = 0; k < 3; ++k) {
                const auto& w = n->host.at(p + names[k] + ".weight");
                const auto& b = n->host.at(p + names[k] + ".bias");
                const int cout = (int)b.size(), cp = (cout + 31) / 32 * 32;
                std::vector<float> wp((size_t)cp * 64, 0.f), bp(cp, 0.f);
                for (int co = 0; co < cout; ++co) {
                    for (int ci = 0; ci < 64; ++ci) wp[(size_t)co * 64 + ci] = w[(size_t)co * 64 + ci];
                    bp[co] = b[co];
                }
                std::vector<uint16_t> pk(fw_pack_pointwise(dt, nullptr, cp, 64, nullptr));
                if (fw_pack_pointwise(dt, wp.data(), cp, 64, pk.data()) != pk.size()) throw Error(FW_ERR_INTERNAL, "fw_pack_pointwise failed");
                upload(a.w[k], pk.data(), pk.size() * 2);
                upload(a.b[k], bp.data(), bp.size() * 4);
                a.tiles[k] = cp / 32;
            }
            std::vector<float> g(64, n->host.at(p + "gamma")[0]);
            upload(a.gamma, g.data(), g.size() * 4);
        }
        n->host.clear();
        n->built = true;
    });
}

size_t fw_aesrgan_workspace_bytes(fw_aesrgan* n, int H, int W) {
    if (!n || H < 1 || W < 1 || !n->built) return 0;
    Arena A;
    A.plan = true;
    try {
        forward(n, A, nullptr, H, W, nullptr, nullptr);
    } catch (...) {
        return 0;
    }
    return A.peak;
}

// x_rgb: fp32 [H][W][3] RGB in [0, 1] on the engine's device; out_rgb: fp32 [sH][sW][3], un-clamped - AESRGAN.forward on one image
int fw_aesrgan_forward_rgb(fw_aesrgan* n, const float* x_rgb, int H, int W, float* out_rgb, void* stream) {
    if (!n || !x_rgb || !out_rgb) return fail(FW_ERR_INVALID, "fw_aesrgan_forward_rgb: NULL argument");
    if (H < 1 || W < 1 || (long)H * W > (1L << 18)) return fail(FW_ERR_INVALID, "fw_aesrgan_forward_rgb: bad crop size (the attention matrix is pixels x pixels)");
    int rc = fw_aesrgan_finalize(n);
    if (rc != FW_OK) return rc;
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        DevGuard dg(n->device);
        Arena P;
        P.plan = true;
        forward(n, P, nullptr, H, W, nullptr, nullptr);
        if (n->ws.bytes < P.peak) {
            FW_HIP_CHECK(hipDeviceSynchronize());
            n->ws.release();
            FW_HIP_CHECK(hipMalloc(&n->ws.p, P.peak));
            n->ws.bytes = P.peak;
        }
        Arena A;
        A.base = (char*)n->ws.p;
        StreamOrder::Scope in_order(n->order, (hipStream_t)stream);
        forward(n, A, x_rgb, H, W, out_rgb, (hipStream_t)stream);
    });
}

int fw_aesrgan_destroy(fw_aesrgan* n) {
    if (!n) return FW_OK;
    { std::lock_guard<std::mutex> lk(n->mu); }   // a call in flight on another thread finishes first
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(n->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : n->convs) {
        kv.second.w.release();
        kv.second.b.release();
    }
    for (auto& kv : n->attn) {
        for (int k = 0; k < 3; ++k) {
            kv.second.w[k].release();
            kv.second.b[k].release();
        }
        kv.second.gamma.release();
    }
    n->ws.release();
    if (prev >= 0) (void)hipSetDevice(prev);
    n->order.destroy();
    delete n;
    return FW_OK;
}

}  // extern "C"
