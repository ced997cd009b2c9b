// SRVGGNetCompact (the Real-ESRGAN checkpoints realesr-animevideov3 / realesr-general-x4v3) as ONE engine behind the C-ABI:
// weights, workspace and launch sequencing.  The reference lists both checkpoints in its model table
// (processors/pytorch_realesrgan.py:119-128) and builds an RRDBNet for them, which cannot load the published weights; SURVEY.md
// section 8(f) item 4 asks for the intended behaviour.  Round 1 sequenced the convolutions from Python over the building-block
// entries (fw_conv3x3_nhwc_ex, fw_pixel_shuffle_add_u8); a non-Python binder of the header now gets the operator:
//
//   body.0: conv 3 -> 64 + PReLU; body.2 .. body.2 n: conv 64 -> 64 + PReLU (num_conv of them); last: conv 64 -> 3 s^2;
//   out = PixelShuffle(s)(last) + nearest-upsampled input -> clamp, x255, rint, uint8 BGR.
//
// Every conv is the 64-output-channel instantiation of conv3x3_mfma.hip on chunk-planar typed activations (PReLU in the epilogue),
// ping-ponging between two buffers; the last conv leaves fp32 for the tail kernel (frame_ops.hip).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
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

struct Layer {
    DevBuf w, b, slopes;
    int chunks = 1;
    std::vector<float> hw, hb, hs;   // as set, until finalize()
    bool have_w = false, have_b = false, have_s = false;
};

}  // namespace

struct fw_srvgg {
    int device = 0;
    fw::StreamOrder order;   // device-side ordering of forwards enqueued on different streams (fw_internal.h)
    DType dt = DT_F16;
    int num_feat = 64, num_conv = 16, scale = 4;
    std::mutex mu;
    std::vector<Layer> layers;   // num_conv + 2
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

size_t up256(size_t v) { return (v + 255) / 256 * 256; }

struct Plan {
    size_t in_u8, out_u8, x0, buf0, buf1, last, total;
};
Plan make_plan(const fw_srvgg* n, int H, int W) {
    Plan p{};
    const size_t M = (size_t)H * W;
    size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += up256(b); return at; };
    p.in_u8 = take(M * 3 * 2);                               // staging for host buffers, 8- or 16-bit samples
    p.out_u8 = take(M * n->scale * n->scale * 3 * 2);
    p.x0 = take(M * 32 * 2);
    p.buf0 = take(M * 64 * 2);
    p.buf1 = take(M * 64 * 2);
    p.last = take(M * 64 * 4);
    p.total = o;
    return p;
}

void forward(fw_srvgg* n, const void* d_in, int bits, int H, int W, void* d_out, float* d_rgb, hipStream_t st) {
    const Plan pl = make_plan(n, H, W);
    char* ws = (char*)n->ws.p;
    const long PL = (long)H * W * 32;    // elements per 32-channel plane
    launch_frame_to_nhwc(n->dt, d_in, bits, H, W, ws + pl.x0, 32, 1, st);
    const void* cur = ws + pl.x0;
    for (size_t i = 0; i < n->layers.size(); ++i) {
        const Layer& L = n->layers[i];
        const bool final = i + 1 == n->layers.size();
        void* dst = final ? nullptr : ws + ((i & 1) ? pl.buf1 : pl.buf0);
        ConvParams p{};
        p.in = cur;
        p.in_cstride = 32;
        p.in_pstride = L.chunks > 1 ? PL : 32;
        p.cin_chunks = L.chunks;
        p.H = H;
        p.W = W;
        p.wpk = L.w.p;
        p.bias = (const float*)L.b.p;
        p.out = dst;
        p.out_cstride = 32;
        p.out_pstride = PL;
        p.out_f32 = final ? (float*)(ws + pl.last) : nullptr;
        p.s1 = p.s2 = 1.f;
        p.act = final ? 0 : 2;                       // PReLU with per-channel slopes
        p.chan_scale = final ? nullptr : (const float*)L.slopes.p;
        launch_conv3x3(n->dt, 2, EPI_STORE, p, st);
        cur = dst;
    }
    launch_pixel_shuffle_add_bits((const float*)(ws + pl.last), 64, d_in, bits, H, W, n->scale, d_out, d_rgb, st);
}

}  // namespace

extern "C" {

int fw_srvgg_create(int device_id, int num_feat, int num_conv, int upscale, int dtype, fw_srvgg** out) {
    if (!out) return fail(FW_ERR_INVALID, "fw_srvgg_create: NULL argument");
    *out = nullptr;
    if (num_feat != 64) return fail(FW_ERR_INVALID, "fw_srvgg_create: num_feat must be 64 (the published checkpoints)");
    if (num_conv < 1 || num_conv > 256 || upscale < 1 || upscale > 4) return fail(FW_ERR_INVALID, "fw_srvgg_create: bad num_conv / upscale");
    if (dtype != FW_DTYPE_BF16 && dtype != FW_DTYPE_F16) return fail(FW_ERR_INVALID, "fw_srvgg_create: bad dtype");
    return guarded([&] {
        int nd = 0;
        FW_HIP_CHECK(hipGetDeviceCount(&nd));
        if (device_id < 0 || device_id >= nd) throw Error(FW_ERR_INVALID, "fw_srvgg_create: no such device");
        auto n = std::make_unique<fw_srvgg>();
        n->device = device_id;
        n->dt = (DType)dtype;
        n->num_feat = num_feat;
        n->num_conv = num_conv;
        n->scale = upscale;
        n->layers.resize((size_t)num_conv + 2);
        *out = n.release();
    });
}

// keys: body.{2 i}.weight [cout][cin][3][3], body.{2 i}.bias [cout], body.{2 i + 1}.weight [64] (PReLU slopes; not for the last conv)
int fw_srvgg_set_tensor(fw_srvgg* n, const char* key_c, const float* data, size_t numel) {
    if (!n || !key_c || !data) return fail(FW_ERR_INVALID, "fw_srvgg_set_tensor: NULL argument");
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        int idx = -1, used = 0;
        const std::string key(key_c);
        if (sscanf(key_c, "body.%d.%n", &idx, &used) != 1 || used <= 0 || idx < 0 || idx > 2 * (n->num_conv + 1))
            throw Error(FW_ERR_INVALID, "fw_srvgg_set_tensor: unknown tensor '" + key + "'");
        const std::string rest = key.substr(used);
        const int i = idx / 2;
        Layer& L = n->layers[(size_t)i];
        const int cin = i == 0 ? 3 : 64, cout = i == n->num_conv + 1 ? 3 * n->scale * n->scale : 64;
        auto need = [&](size_t want) {
            if (numel != want)
                throw Error(FW_ERR_INVALID, "fw_srvgg_set_tensor: '" + key + "' has " + std::to_string(numel) + " elements, expected " + std::to_string(want));
        };
        if ((idx & 1) == 0 && rest == "weight") { need((size_t)cout * cin * 9); L.hw.assign(data, data + numel); L.have_w = true; }
        else if ((idx & 1) == 0 && rest == "bias") { need((size_t)cout); L.hb.assign(data, data + numel); L.have_b = true; }
        else if ((idx & 1) == 1 && rest == "weight" && i <= n->num_conv) { need(64); L.hs.assign(data, data + numel); L.have_s = true; }
        else throw Error(FW_ERR_INVALID, "fw_srvgg_set_tensor: unknown tensor '" + key + "'");
        n->built = false;
    });
}

int fw_srvgg_finalize(fw_srvgg* n) {
    if (!n) return fail(FW_ERR_INVALID, "fw_srvgg_finalize: NULL");
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        if (n->built) return;
        for (size_t i = 0; i < n->layers.size(); ++i) {
            const Layer& L = n->layers[i];
            const bool last = i + 1 == n->layers.size();
            if (!L.have_w || !L.have_b || (!last && !L.have_s))
                throw Error(FW_ERR_INVALID, "fw_srvgg_finalize: missing tensors of body." + std::to_string(2 * i));
        }
        DevGuard dg(n->device);
        FW_HIP_CHECK(hipDeviceSynchronize());
        for (size_t i = 0; i < n->layers.size(); ++i) {
            Layer& L = n->layers[i];
            const int cin = i == 0 ? 3 : 64, cout = i + 1 == n->layers.size() ? 3 * n->scale * n->scale : 64;
            L.chunks = (cin + 31) / 32;
            const int cinp = 32 * L.chunks;
            std::vector<float> wp((size_t)64 * cinp * 9, 0.f), bp(64, 0.f);
            for (int co = 0; co < cout; ++co) {
                for (int ci = 0; ci < cin; ++ci)
                    for (int t = 0; t < 9; ++t) wp[((size_t)co * cinp + ci) * 9 + t] = L.hw[((size_t)co * cin + ci) * 9 + t];
                bp[co] = L.hb[co];
            }
            std::vector<uint16_t> pk(pack_conv3x3_weights(n->dt, nullptr, 64, cinp, 2, L.chunks, nullptr));
            pack_conv3x3_weights(n->dt, wp.data(), 64, cinp, 2, L.chunks, pk.data());
            upload(L.w, pk.data(), pk.size() * 2);
            upload(L.b, bp.data(), bp.size() * 4);
            if (L.have_s) upload(L.slopes, L.hs.data(), 64 * 4);
        }
        n->built = true;
    });
}

size_t fw_srvgg_workspace_bytes(const fw_srvgg* n, int H, int W) {
    if (!n || H < 1 || W < 1) return 0;
    return make_plan(n, H, W).total;
}

double fw_srvgg_flops(const fw_srvgg* n, int H, int W) {
    if (!n || H < 1 || W < 1) return 0.0;
    const double mac = 9.0 * (3.0 * 64 + (double)n->num_conv * 64 * 64 + 64.0 * 3 * n->scale * n->scale);
    return 2.0 * mac * H * W;
}

namespace {
int upscale_any(fw_srvgg* n, const void* in_bgr, int bits, int in_loc, int H, int W, void* out_bgr, int out_loc, float* out_rgb_f32, void* stream,
                const char* who) {
    if (!n || !in_bgr) return fail(FW_ERR_INVALID, std::string(who) + ": NULL argument");
    if (!out_bgr && !out_rgb_f32) return fail(FW_ERR_INVALID, std::string(who) + ": no output requested");
    if (H < 1 || W < 1 || H > 16384 || W > 16384) return fail(FW_ERR_INVALID, std::string(who) + ": bad frame size");
    if ((in_loc != FW_HOST && in_loc != FW_DEVICE) || (out_loc != FW_HOST && out_loc != FW_DEVICE))
        return fail(FW_ERR_INVALID, std::string(who) + ": bad buffer location");
    int rc = fw_srvgg_finalize(n);
    if (rc != FW_OK) return rc;
    return guarded([&] {
        std::lock_guard<std::mutex> lk(n->mu);
        DevGuard dg(n->device);
        hipStream_t st = (hipStream_t)stream;
        StreamOrder::Scope in_order(n->order, st);
        const Plan pl = make_plan(n, H, W);
        if (n->ws.bytes < pl.total) {
            FW_HIP_CHECK(hipDeviceSynchronize());
            n->ws.release();
            FW_HIP_CHECK(hipMalloc(&n->ws.p, pl.total));
            n->ws.bytes = pl.total;
        }
        char* ws = (char*)n->ws.p;
        const size_t in_bytes = (size_t)H * W * 3 * (bits / 8), out_bytes = in_bytes * n->scale * n->scale;
        const void* d_in = in_bgr;
        if (in_loc == FW_HOST) {
            FW_HIP_CHECK(hipMemcpyAsync(ws + pl.in_u8, in_bgr, in_bytes, hipMemcpyHostToDevice, st));
            d_in = ws + pl.in_u8;
        }
        void* d_out = out_bgr;
        if (out_bgr && out_loc == FW_HOST) d_out = ws + pl.out_u8;
        forward(n, d_in, bits, H, W, d_out, out_rgb_f32, st);
        if (out_bgr && out_loc == FW_HOST) {
            FW_HIP_CHECK(hipMemcpyAsync(out_bgr, d_out, out_bytes, hipMemcpyDeviceToHost, st));
            FW_HIP_CHECK(hipStreamSynchronize(st));
        }
    });
}
}  // namespace

int fw_srvgg_upscale_u8(fw_srvgg* n, const uint8_t* in_bgr, int in_loc, int H, int W, uint8_t* out_bgr, int out_loc, float* out_rgb_f32,
                        void* stream) {
    return upscale_any(n, in_bgr, 8, in_loc, H, W, out_bgr, out_loc, out_rgb_f32, stream, "fw_srvgg_upscale_u8");
}

int fw_srvgg_upscale_u16(fw_srvgg* n, const uint16_t* in_bgr, int in_loc, int H, int W, uint16_t* out_bgr, int out_loc, float* out_rgb_f32,
                         void* stream) {
    return upscale_any(n, in_bgr, 16, in_loc, H, W, out_bgr, out_loc, out_rgb_f32, stream, "fw_srvgg_upscale_u16");
}

int fw_srvgg_destroy(fw_srvgg* n) {
    if (!n) return FW_OK;
    { std::lock_guard<std::mutex> lk(n->mu); }   // a call in flight on another thread finishes first
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(n->device);
    (void)hipDeviceSynchronize();
    for (auto& L : n->layers) {
        L.w.release();
        L.b.release();
        L.slopes.release();
    }
    n->ws.release();
    if (prev >= 0) (void)hipSetDevice(prev);
    n->order.destroy();
    delete n;
    return FW_OK;
}

}  // extern "C"
