// abi.hip -- the extern "C" surface declared in include/bitnet_hip.h.
//
// Argument validation mirrors the reference function each symbol replaces (cited in
// the header); kernels live in kernels_*.hip.  No CPU fallback exists here: every
// compute path needs a HIP device.
#include <atomic>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "qact.hpp"

namespace bitnet_hip {

thread_local std::string g_last_error;

namespace {

// The provider trait is Send + Sync (K/lib.rs:39) and a KernelManager is shared through an Arc per layer
// (crates/bitnet-inference/src/layers/quantized_linear.rs:165,208): every entry point may run concurrently with every
// other one, weights_free included.  Handles therefore map to REFERENCE-COUNTED matrices: a call holds its own
// reference for its whole duration, weights_free only drops the table's, and the device memory goes with the last one.
std::mutex g_mu;
bool g_inited = false;
int g_device = 0;
std::atomic<int> g_kernel{BITNET_HIP_KERNEL_AUTO};  // default kernel of the calls that do not name one
uint64_t g_next_handle = 1;
using WeightsRef = std::shared_ptr<Weights>;
std::unordered_map<uint64_t, WeightsRef> g_weights;

void free_weights(Weights *w) {
    if (!w) return;
    if (w->codes) (void)hipFree(w->codes);
    if (w->scales) (void)hipFree(w->scales);
    if (w->tiles) (void)hipFree(w->tiles);
    if (w->tiles4) (void)hipFree(w->tiles4);
    if (w->scale_tiles) (void)hipFree(w->scale_tiles);
    if (w->scale_tiles_h) (void)hipFree(w->scale_tiles_h);
    if (w->ln_g) (void)hipFree(w->ln_g);
    delete w;
}

WeightsRef lookup(bitnet_hip_weights_t h) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_weights.find(h);
    return it == g_weights.end() ? WeightsRef() : it->second;
}

int ensure_init() {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_inited) return BITNET_HIP_OK;
    }
    return bitnet_hip_init(-1);
}

int register_weights(Weights *w, bitnet_hip_weights_t *out) {
    WeightsRef ref(w, free_weights);
    std::lock_guard<std::mutex> lk(g_mu);
    uint64_t h = g_next_handle++;
    g_weights[h] = std::move(ref);
    *out = h;
    return BITNET_HIP_OK;
}

// RAII device scratch for the host-pointer drop-ins.
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T>
    T *as() {
        return static_cast<T *>(p);
    }
};

// Sizes are computed as products of caller-supplied dimensions: refuse anything whose products could
// wrap size_t or exceed the kernels' 32-bit row/column indices before multiplying.
bool dims_sane(size_t a, size_t b, size_t c = 1) {
    const size_t lim = size_t(1) << 31;
    if (a >= lim || b >= lim || c >= lim) return false;
    unsigned long long ab = 0, abc = 0;
    if (__builtin_mul_overflow((unsigned long long)a, (unsigned long long)b, &ab)) return false;
    if (__builtin_mul_overflow(ab, (unsigned long long)c, &abc)) return false;
    return abc < (1ull << 44);
}
#define BH_CHECK_DIMS(...)                                                                               \
    if (!dims_sane(__VA_ARGS__))                                                                         \
    return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "dimensions too large for this library (%s)", #__VA_ARGS__)

int run_gemv(Weights &w, const float *x_dev, float *y_dev, size_t m, const GemvFusion &fu, hipStream_t stream, int kernel = -1) {
    if (kernel < 0) kernel = g_kernel.load(std::memory_order_relaxed);
    if (kernel == BITNET_HIP_KERNEL_AUTO) kernel = BITNET_HIP_KERNEL_MFMA;
    if ((kernel == BITNET_HIP_KERNEL_MFMA || kernel == BITNET_HIP_KERNEL_MFMA_TILED) && !mfma_supported(w))
        kernel = BITNET_HIP_KERNEL_VALU;
    if (kernel == BITNET_HIP_KERNEL_VALU && !valu_supported(w)) kernel = BITNET_HIP_KERNEL_EXACT;
    if ((fu.ln_gamma || fu.residual || fu.silu_mul || fu.attn_rec) && kernel != BITNET_HIP_KERNEL_MFMA && kernel != BITNET_HIP_KERNEL_MFMA_TILED) {
        // A shape (or a kernel choice) the fused MFMA GEMV does not take: the same result as separate launches on the device --
        // LayerNorm rows, the product, silu * mul, residual add (the reference's own op order) -- through stream-ordered scratch.
        if (fu.attn_rec || fu.qout)
            return set_error(BITNET_HIP_ERR_UNSUPPORTED, "the merging / QAct forms need the MFMA GEMV (shape %zux%zu unsupported)", w.rows, w.cols);
        if (fu.silu_mul && (!w.paired || fu.residual)) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "FUSE_SILU_MUL needs an interleaved (gate, up) handle and no residual");
        float *xn = nullptr, *yt = nullptr;
        hipError_t e = hipSuccess;
        auto done = [&](int rc) {
            if (xn) (void)hipFreeAsync(xn, stream);
            if (yt) (void)hipFreeAsync(yt, stream);
            return rc;
        };
        const float *xin = x_dev;
        if (fu.ln_gamma) {
            e = hipMallocAsync((void **)&xn, m * w.cols * sizeof(float), stream);
            if (e == hipSuccess) e = launch_norm_rows(x_dev, fu.ln_gamma, xn, (int)m, (int)w.cols, fu.ln_eps, false, stream);
            if (e != hipSuccess) return done(set_error(BITNET_HIP_ERR_GPU, "LayerNorm ahead of the GEMV failed: %s", hipGetErrorString(e)));
            xin = xn;
        }
        float *yout = y_dev;
        if (fu.silu_mul) {
            e = hipMallocAsync((void **)&yt, m * w.rows * sizeof(float), stream);
            if (e != hipSuccess) return done(set_error(BITNET_HIP_ERR_GPU, "scratch allocation failed: %s", hipGetErrorString(e)));
            yout = yt;
        }
        const int rc = run_gemv(w, xin, yout, m, GemvFusion{}, stream, kernel);
        if (rc != BITNET_HIP_OK) return done(rc);
        if (fu.silu_mul) e = launch_silu_mul(yt, yt + 16, y_dev, m * (w.rows / 2), 16, stream);
        if (e == hipSuccess && fu.residual) e = launch_add(y_dev, fu.residual, y_dev, m * w.rows, stream);
        if (e != hipSuccess) return done(set_error(BITNET_HIP_ERR_GPU, "kernel launch failed: %s", hipGetErrorString(e)));
        return done(BITNET_HIP_OK);
    }
    hipError_t e;
    if (kernel == BITNET_HIP_KERNEL_MFMA_TILED || kernel == BITNET_HIP_KERNEL_MFMA) {
        e = build_tiles(w, stream);  // no-op after the first call (done at upload normally)
        if (e == hipSuccess) e = launch_gemv_mfma(w, x_dev, y_dev, m, fu, stream);
    } else if (kernel == BITNET_HIP_KERNEL_VALU) {
        ReferencePin pin(w, stream);  // the reference-layout copies are dropped at upload and rebuilt for these kernels
        e = pin.status;
        if (e == hipSuccess) e = launch_gemv_valu(w, x_dev, y_dev, m, stream);
    } else {
        ReferencePin pin(w, stream);
        e = pin.status;
        if (e == hipSuccess) e = launch_gemv_exact(w, x_dev, y_dev, m, stream);
    }
    if (e != hipSuccess)
        return set_error(BITNET_HIP_ERR_GPU, "kernel launch failed: %s", hipGetErrorString(e));
    return BITNET_HIP_OK;
}

}  // namespace
}  // namespace bitnet_hip

using namespace bitnet_hip;

#define BH_GUARD_BEGIN try {
#define BH_GUARD_END                                                                     \
    }                                                                                    \
    catch (const std::exception &e) {                                                    \
        return set_error(BITNET_HIP_ERR_EXECUTION, "%s", e.what());                      \
    }                                                                                    \
    catch (...) {                                                                        \
        return set_error(BITNET_HIP_ERR_EXECUTION, "unknown C++ exception");             \
    }

extern "C" {

int bitnet_hip_init(int device) {
    BH_GUARD_BEGIN
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return set_error(BITNET_HIP_ERR_GPU, "no HIP device available (hipGetDeviceCount: %s)",
                         hipGetErrorString(e));
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= count)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "device %d out of range (%d visible)", device,
                         count);
    BH_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    BH_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(BITNET_HIP_ERR_UNSUPPORTED,
                         "device %d is %s; this library carries gfx950 (MI355X) code objects only",
                         device, prop.gcnArchName);
    std::lock_guard<std::mutex> lk(g_mu);
    g_device = device;
    g_inited = true;
    g_last_error.clear();
    return BITNET_HIP_OK;
    BH_GUARD_END
}

void bitnet_hip_cleanup(void) {
    try {
        std::lock_guard<std::mutex> lk(g_mu);
        g_weights.clear();  // matrices still used by a running call go with that call's reference
        g_inited = false;
        g_last_error.clear();
    } catch (...) {
    }
}

int bitnet_hip_is_available(void) {
    int count = 0;
    return hipGetDeviceCount(&count) == hipSuccess && count > 0 ? 1 : 0;
}

int bitnet_hip_device_count(void) {
    int count = 0;
    return hipGetDeviceCount(&count) == hipSuccess ? count : 0;
}

const char *bitnet_hip_get_last_error(void) {
    return g_last_error.empty() ? nullptr : g_last_error.c_str();
}

int bitnet_hip_get_device_info(int device, bitnet_hip_device_info *out) {
    BH_GUARD_BEGIN
    if (!out) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to get_device_info");
    hipDeviceProp_t prop;
    BH_HIP_TRY(hipGetDeviceProperties(&prop, device));
    memset(out, 0, sizeof(*out));
    out->device_id = device;
    strncpy(out->name, prop.name, sizeof(out->name) - 1);
    strncpy(out->gcn_arch, prop.gcnArchName, sizeof(out->gcn_arch) - 1);
    out->total_memory = prop.totalGlobalMem;
    out->compute_unit_count = prop.multiProcessorCount;
    out->max_wavefront_size = prop.warpSize;
    out->max_shared_memory_per_workgroup = prop.sharedMemPerBlock;
    out->supports_fp16 = 1;
    out->supports_bf16 = 1;
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_set_kernel(int kernel) {
    if (kernel < BITNET_HIP_KERNEL_AUTO || kernel > BITNET_HIP_KERNEL_MFMA_TILED)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown kernel id %d", kernel);
    g_kernel.store(kernel, std::memory_order_relaxed);
    return BITNET_HIP_OK;
}

int bitnet_hip_get_kernel(void) { return g_kernel.load(std::memory_order_relaxed); }

/* ---------------------------------------------------------------- handles */

int bitnet_hip_weights_upload_qk256(const uint8_t *qs_data, size_t qs_len, size_t rows, size_t cols,
                                    size_t row_stride_bytes, bitnet_hip_weights_t *out) {
    BH_GUARD_BEGIN
    if (!qs_data || !out)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to weights_upload_qk256");
    if (rows == 0 || cols == 0)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "I2SQk256NoScale: rows and cols must be > 0");
    BH_CHECK_DIMS(rows, cols);
    const size_t stride = div_ceil(cols, 256) * 64;
    if (row_stride_bytes != stride)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT,
                         "I2S_QK256: row bytes mismatch: got %zu, expected %zu for %zu cols",
                         row_stride_bytes, stride, cols);
    const size_t expected = rows * stride;
    const size_t diff = qs_len > expected ? qs_len - expected : expected - qs_len;
    if (diff > 128)  // Q/i2s_qk256.rs:91-103
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT,
                         "I2SQk256NoScale: data size mismatch: got %zu bytes, expected %zu for %zux%zu "
                         "matrix. Check tensor orientation: QK256 requires [out_dim, in_dim] layout.",
                         qs_len, expected, rows, cols);
    if (qs_len < expected)  // gemv_qk256 would refuse it (Q/i2s_qk256.rs:308-311)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "I2S_QK256: data too short: %zu < %zu", qs_len,
                         expected);
    int rc = ensure_init();
    if (rc) return rc;
    Weights *w = new Weights();
    w->rows = rows;
    w->cols = cols;
    w->row_stride_bytes = stride;
    w->lut = LUT_QK256;
    w->algorithmic_bytes = expected;
    w->device = g_device;
    if (hipMalloc((void **)&w->codes, expected) != hipSuccess) {
        free_weights(w);
        return set_error(BITNET_HIP_ERR_GPU, "hipMalloc(%zu) failed for QK256 codes", expected);
    }
    if (hipMemcpy(w->codes, qs_data, expected, hipMemcpyHostToDevice) != hipSuccess) {
        free_weights(w);
        return set_error(BITNET_HIP_ERR_GPU, "hipMemcpy H2D failed for QK256 codes");
    }
    if (mfma_supported(*w) && (build_tiles(*w, nullptr) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) {
        free_weights(w);
        return set_error(BITNET_HIP_ERR_GPU, "re-tiling QK256 codes for the MFMA kernel failed");
    }
    if (mfma_supported(*w)) trim_reference(*w);  // one copy of the weights on the device; ensure_reference() rebuilds on demand
    return register_weights(w, out);
    BH_GUARD_END
}

}  // extern "C"

// 2-bit coded matrix [n, ceil(k/4)] + f32 scale per (row, block) with any 4-entry code map.
static int upload_coded(const uint8_t *weights_packed, size_t w_len, const float *scales, size_t scales_len, size_t n,
                        size_t k, size_t block_size, uint32_t lut, bitnet_hip_weights_t *out) {
    BH_GUARD_BEGIN
    if (!weights_packed || !scales || !out)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to weights_upload_i2s");
    if (block_size == 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "block_size must be > 0");
    if (n == 0 || k == 0)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "dimensions must be > 0: n=%zu, k=%zu", n, k);
    BH_CHECK_DIMS(n, k);
    const size_t packed_k = div_ceil(k, 4), nblk = div_ceil(k, block_size);
    if (w_len < packed_k * n)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "weights_packed too small: expected %zu, got %zu",
                         packed_k * n, w_len);
    if (scales_len < n * nblk)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "scales too small: expected %zu, got %zu",
                         n * nblk, scales_len);
    int rc = ensure_init();
    if (rc) return rc;
    Weights *w = new Weights();
    w->rows = n;
    w->cols = k;
    w->row_stride_bytes = packed_k;
    w->block_size = block_size;
    w->nblk = nblk;
    w->lut = lut;
    w->scaled = true;
    if (block_size == 32) {  // f16-exact scales (BitNet32-F16): keep them as f16 in the streaming layout
        bool exact = true, x2 = true;
        for (size_t i = 0; i < n * nblk && exact; ++i) {
            exact = (float)(_Float16)scales[i] == scales[i];
            x2 = x2 && fabsf(scales[i]) <= 32752.0f;
        }
        w->scales_f16 = exact;
        w->scales_f16_x2_finite = exact && x2;
    }
    w->algorithmic_bytes = packed_k * n + (w->scales_f16 ? 2 : 4) * n * nblk;
    w->device = g_device;
    if (hipMalloc((void **)&w->codes, packed_k * n + 16) != hipSuccess ||
        hipMalloc((void **)&w->scales, 4 * n * nblk) != hipSuccess) {
        free_weights(w);
        return set_error(BITNET_HIP_ERR_GPU, "hipMalloc failed for I2_S weights");
    }
    if (hipMemcpy(w->codes, weights_packed, packed_k * n, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(w->scales, scales, 4 * n * nblk, hipMemcpyHostToDevice) != hipSuccess) {
        free_weights(w);
        return set_error(BITNET_HIP_ERR_GPU, "hipMemcpy H2D failed for I2_S weights");
    }
    if (mfma_supported(*w) && (build_tiles(*w, nullptr) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) {
        free_weights(w);
        return set_error(BITNET_HIP_ERR_GPU, "re-tiling I2_S codes for the MFMA kernel failed");
    }
    if (mfma_supported(*w)) trim_reference(*w);
    return register_weights(w, out);
    BH_GUARD_END
}

extern "C" {

int bitnet_hip_weights_upload_i2s(const uint8_t *weights_packed, size_t w_len, const float *scales,
                                  size_t scales_len, size_t n, size_t k, size_t block_size,
                                  bitnet_hip_weights_t *out) {
    return upload_coded(weights_packed, w_len, scales, scales_len, n, k, block_size, LUT_TERNARY, out);
}

int bitnet_hip_weights_upload_coded(const uint8_t *weights_packed, size_t w_len, const float *scales,
                                    size_t scales_len, size_t n, size_t k, size_t block_size, const int8_t *code_map,
                                    bitnet_hip_weights_t *out) {
    if (!code_map) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to weights_upload_coded");
    return upload_coded(weights_packed, w_len, scales, scales_len, n, k, block_size,
                        pack_lut(code_map[0], code_map[1], code_map[2], code_map[3]), out);
}

int bitnet_hip_weights_upload_inline_f16(const uint8_t *blocks, size_t len, size_t n, size_t k, const int8_t *code_map,
                                         int scale_mode, bitnet_hip_weights_t *out) {
    BH_GUARD_BEGIN
    if (!blocks || !code_map || !out)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to weights_upload_inline_f16");
    if (n == 0 || k == 0 || k % 32 != 0)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "inline f16 blocks need k %% 32 == 0 and n, k > 0: n=%zu, k=%zu", n, k);
    BH_CHECK_DIMS(n, k);
    const size_t nb = n * (k / 32);
    if (len < nb * 10)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "blocks too small: expected %zu, got %zu", nb * 10, len);
    std::vector<uint8_t> codes(nb * 8);
    std::vector<float> scales(nb);
    for (size_t b = 0; b < nb; ++b) {
        memcpy(&codes[b * 8], blocks + b * 10, 8);
        uint16_t bits;
        memcpy(&bits, blocks + b * 10 + 8, 2);
        _Float16 h;
        memcpy(&h, &bits, 2);
        float s = (float)h;
        if (scale_mode == 1) {  // M/quant/i2s.rs:84-101
            s = fabsf(s);
            if (!std::isnan(s)) s = fminf(fmaxf(s, 1e-3f), 1e3f);  // f32::clamp keeps NaN
        }
        scales[b] = s;
    }
    return upload_coded(codes.data(), codes.size(), scales.data(), scales.size(), n, k, 32,
                        pack_lut(code_map[0], code_map[1], code_map[2], code_map[3]), out);
    BH_GUARD_END
}

int bitnet_hip_weights_free(bitnet_hip_weights_t h) {
    BH_GUARD_BEGIN
    WeightsRef w;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_weights.find(h);
        if (it == g_weights.end())
            return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown weights handle %llu",
                             (unsigned long long)h);
        w = std::move(it->second);
        g_weights.erase(it);
    }
    w.reset();  // the last reference frees the device memory: here, or when a concurrent call on this handle returns
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_weights_info(bitnet_hip_weights_t h, size_t *rows, size_t *cols, size_t *algorithmic_bytes) {
    const WeightsRef w = lookup(h);
    if (!w) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown weights handle %llu", (unsigned long long)h);
    if (rows) *rows = w->rows;
    if (cols) *cols = w->cols;
    if (algorithmic_bytes) *algorithmic_bytes = w->algorithmic_bytes;
    return BITNET_HIP_OK;
}

int bitnet_hip_gemv_dev(bitnet_hip_weights_t h, const float *x_dev, float *y_dev, void *stream) {
    return bitnet_hip_matmul_dev(h, x_dev, y_dev, 1, stream);
}

static int matmul_dev_kernel(bitnet_hip_weights_t h, const float *x_dev, float *y_dev, size_t m, int kernel, void *stream) {
    BH_GUARD_BEGIN
    const WeightsRef w = lookup(h);
    if (!w) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown weights handle %llu", (unsigned long long)h);
    if (!x_dev || !y_dev) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to matmul_dev");
    if (m == 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "dimensions must be > 0: m=0");
    if (kernel < 0) kernel = g_kernel.load(std::memory_order_relaxed);
    if (kernel < BITNET_HIP_KERNEL_AUTO || kernel > BITNET_HIP_KERNEL_MFMA_TILED)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown kernel id %d", kernel);
    const bool mfma_ok = kernel == BITNET_HIP_KERNEL_AUTO || kernel == BITNET_HIP_KERNEL_MFMA || kernel == BITNET_HIP_KERNEL_MFMA_TILED;
    if (m >= 16 && mfma_ok && gemm_supported(*w)) {
        // many rows: one tiled matmul instead of m GEMV launches.  The digit-plane workspace is allocated and
        // released IN STREAM ORDER (no host synchronisation, nothing that outlives the call on the host side);
        // callers that replay the call from a hipGraph pass their own workspace to bitnet_hip_matmul_fused_dev.
        const size_t wsb = gemm_workspace_bytes(m, w->cols, 4);
        std::unique_ptr<ReferencePin> pin;  // row-major block scales: held until the launch is enqueued
        if (gemm_needs_row_major_scales(*w)) {
            pin.reset(new ReferencePin(*w, (hipStream_t)stream, /*scales_only=*/true));
            if (pin->status != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "rebuilding the row-major block scales failed");
        }
        void *ws = nullptr;
        if (hipMallocAsync(&ws, wsb, (hipStream_t)stream) != hipSuccess)
            return set_error(BITNET_HIP_ERR_GPU, "hipMallocAsync failed for the matmul workspace (%zu bytes)", wsb);
        hipError_t e = launch_gemm_mfma(*w, x_dev, y_dev, m, GemvFusion(), 4, ws, wsb, (hipStream_t)stream);
        const hipError_t ef = hipFreeAsync(ws, (hipStream_t)stream);
        if (e == hipSuccess) e = ef;
        if (e != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "kernel launch failed: %s", hipGetErrorString(e));
        return BITNET_HIP_OK;
    }
    return run_gemv(*w, x_dev, y_dev, m, GemvFusion(), (hipStream_t)stream, kernel);
    BH_GUARD_END
}

int bitnet_hip_matmul_dev(bitnet_hip_weights_t h, const float *x_dev, float *y_dev, size_t m, void *stream) {
    return matmul_dev_kernel(h, x_dev, y_dev, m, -1, stream);
}

int bitnet_hip_matmul_kernel_dev(bitnet_hip_weights_t h, const float *x_dev, float *y_dev, size_t m, int kernel, void *stream) {
    if (kernel < 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown kernel id %d", kernel);
    return matmul_dev_kernel(h, x_dev, y_dev, m, kernel, stream);
}

size_t bitnet_hip_matmul_workspace_bytes(size_t m, size_t k, int digits) {
    return gemm_workspace_bytes(m, k, digits >= 2 && digits <= 4 ? digits : 4);
}

int bitnet_hip_matmul_fused_dev(bitnet_hip_weights_t h, const float *x_dev, float *y_dev, size_t m,
                                const float *ln_gamma_dev, float ln_eps, const float *residual_dev, int flags, int digits,
                                void *workspace_dev, size_t workspace_bytes, void *stream) {
    BH_GUARD_BEGIN
    const WeightsRef w = lookup(h);
    if (!w) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown weights handle %llu", (unsigned long long)h);
    if (!x_dev || !y_dev) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to matmul_fused_dev");
    if (m == 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "dimensions must be > 0: m=0");
    if (digits < 2 || digits > 4) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "digits must be 2, 3 or 4, got %d", digits);
    GemvFusion fu;
    fu.ln_gamma = ln_gamma_dev;
    fu.ln_eps = ln_eps;
    fu.residual = residual_dev;
    fu.silu_mul = (flags & BITNET_HIP_FUSE_SILU_MUL) != 0;
    fu.x_f16 = (flags & BITNET_HIP_FUSE_X_F16) != 0;
    fu.y_f16 = (flags & BITNET_HIP_FUSE_Y_F16) != 0;
    fu.int8_form = (flags & BITNET_HIP_FUSE_INT8_DIGITS) != 0;
    fu.fp6_form = (flags & BITNET_HIP_FUSE_FP6_DIGITS) != 0;
    fu.fp6_expand = (flags & BITNET_HIP_FUSE_FP6_EXPAND) != 0;
    if (fu.fp6_form && (fu.int8_form || digits != 2 || !gemm_fp6_supported(*w)))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "FUSE_FP6_DIGITS needs digits = 2, an unscaled matrix with a code map in -2..2, and no FUSE_INT8_DIGITS");
    if (fu.silu_mul && (!w->paired || residual_dev))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT,
                         "FUSE_SILU_MUL needs a handle from weights_concat(..., interleave16=1) and no residual");
    if ((fu.x_f16 || fu.y_f16) && (!gemm_supported(*w) || (fu.y_f16 && (!fu.silu_mul || (w->rows / 2) % 4 != 0)) || (fu.x_f16 && ln_gamma_dev)))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "FUSE_X_F16 (no LayerNorm) / FUSE_Y_F16 (with FUSE_SILU_MUL) need a matrix the tiled matmul takes");
    if (!gemm_supported(*w)) return run_gemv(*w, x_dev, y_dev, m, fu, (hipStream_t)stream);  // 32-element block scales: row by row
    const size_t need = gemm_workspace_bytes(m, w->cols, digits);
    if (!workspace_dev || workspace_bytes < need)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "workspace too small: expected %zu, got %zu", need, workspace_dev ? workspace_bytes : (size_t)0);
    std::unique_ptr<ReferencePin> pin;  // row-major block scales: held until the launch is enqueued
    if (gemm_needs_row_major_scales(*w)) {
        pin.reset(new ReferencePin(*w, (hipStream_t)stream, /*scales_only=*/true));
        if (pin->status != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "rebuilding the row-major block scales failed");
    }
    // the fp6 x fp4 form reads its weights from the resident fp4 image: built here on first use (an allocation + one retile launch:
    // hosts that capture or time the call build it ahead with bitnet_hip_weights_fp4_image)
    if (gemm_fp4_resident_enabled() && !fu.fp6_expand && !w->tiles4 && gemm_takes_fp6(*w, fu, digits)) {
        const hipError_t ei = ensure_fp4_image(*w, (hipStream_t)stream);
        if (ei != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "building the resident fp4 weight image failed: %s", hipGetErrorString(ei));
    }
    hipError_t e = launch_gemm_mfma(*w, x_dev, y_dev, m, fu, digits, workspace_dev, workspace_bytes, (hipStream_t)stream);
    if (e == hipErrorInvalidValue && (fu.x_f16 || fu.y_f16))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "FUSE_X_F16 / FUSE_Y_F16: this matrix runs on the f16 matrix cores at this digit count: use bitnet_hip_matmul_f16_dev");
    if (e != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "kernel launch failed: %s", hipGetErrorString(e));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_weights_fp4_image(bitnet_hip_weights_t h, int enable, void *stream) {
    BH_GUARD_BEGIN
    const WeightsRef w = lookup(h);
    if (!w) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown weights handle %llu", (unsigned long long)h);
    if (!enable) {
        drop_fp4_image(*w);
        return BITNET_HIP_OK;
    }
    if (!gemm_fp6_supported(*w))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "weights_fp4_image: needs an unscaled matrix whose code map lies in -2..2 (the fp6 x fp4 form's matrices)");
    const hipError_t e = ensure_fp4_image(*w, (hipStream_t)stream);
    if (e != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "building the resident fp4 weight image failed: %s", hipGetErrorString(e));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_matmul_f16_supported(bitnet_hip_weights_t h) {
    const WeightsRef w = lookup(h);
    return w && gemm_f16_chain_supported(*w) ? 1 : 0;
}

int bitnet_hip_rows_to_f16_dev(const float *x_dev, const float *gamma_dev, size_t m, size_t cols, void *xh_dev, float *stats_dev, void *stream) {
    BH_GUARD_BEGIN
    if (!x_dev || !xh_dev) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to rows_to_f16_dev");
    if (m == 0 || cols == 0 || cols % 4 != 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "rows_to_f16_dev: m > 0 and cols %% 4 == 0 expected: m=%zu, cols=%zu", m, cols);
    hipError_t e = launch_rows_to_f16(x_dev, gamma_dev, m, cols, xh_dev, stats_dev, (hipStream_t)stream);
    if (e != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "kernel launch failed: %s", hipGetErrorString(e));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_matmul_f16_dev(bitnet_hip_weights_t h, const void *xh_dev, size_t m, const float *stats_in_dev, size_t n_stats,
                              const float *ln_gamma_dev, float ln_eps, float *y_dev, const float *residual_dev, int flags, void *yh_dev,
                              const float *gamma_out_dev, float *stats_out_dev, void *stream) {
    BH_GUARD_BEGIN
    const WeightsRef w = lookup(h);
    if (!w) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown weights handle %llu", (unsigned long long)h);
    if (!xh_dev || (!y_dev && !yh_dev)) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to matmul_f16_dev");
    if (m == 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "dimensions must be > 0: m=0");
    if (!gemm_f16_chain_supported(*w))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "matmul_f16_dev: this matrix does not take the f16 chain (rows %% 256, cols %% 256, code map in -2..2, f16 block scales)");
    if ((ln_gamma_dev != nullptr) != (stats_in_dev != nullptr) || (ln_gamma_dev && (!w->ln_g || w->ln_gamma_bound != ln_gamma_dev || n_stats == 0)))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "matmul_f16_dev: LayerNorm needs the bound gamma (bitnet_hip_weights_bind_ln) and its statistics partials");
    GemmF16Io io;
    io.xh = xh_dev;
    io.stats_in = stats_in_dev;
    io.n_stats = (int)n_stats;
    io.ln_eps = ln_eps;
    io.y = y_dev;
    io.residual = residual_dev;
    io.silu_mul = (flags & BITNET_HIP_FUSE_SILU_MUL) != 0;
    io.yh = yh_dev;
    io.gamma_out = gamma_out_dev;
    io.stats_out = stats_out_dev;
    if (flags & BITNET_HIP_FUSE_YH_QB32) {  // yh_dev is a QB32 buffer (bitnet_hip_qb32_bytes(m, rows)): the next fp6-form matmul's input
        if (!yh_dev || io.silu_mul || w->rows % 256 != 0)
            return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "FUSE_YH_QB32 needs a QB32 buffer in yh_dev, rows %% 256 == 0 and no FUSE_SILU_MUL");
        io.qb_out = yh_dev;
        io.yh = nullptr;
    }
    if (io.silu_mul && (!w->paired || residual_dev))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "FUSE_SILU_MUL needs a handle from weights_concat(..., interleave16=1) and no residual");
    hipError_t e = launch_gemm_f16_chain(*w, io, m, (hipStream_t)stream);
    if (e == hipErrorInvalidValue && io.qb_out)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "FUSE_YH_QB32: this launch does not run 64-token tiles (too few rows for the QB32 hand-over)");
    if (e != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "kernel launch failed: %s", hipGetErrorString(e));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

unsigned long long bitnet_hip_f16_saturations(int reset) { return f16_saturations(reset != 0); }

size_t bitnet_hip_qb32_bytes(size_t m, size_t cols) { return qb32_bytes(m, cols); }

int bitnet_hip_rows_to_qb32_dev(const float *x_dev, const float *gamma_dev, size_t m, size_t cols, void *qb_dev, float *stats_dev, void *stream) {
    BH_GUARD_BEGIN
    if (!x_dev || !qb_dev) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to rows_to_qb32_dev");
    if (m == 0 || cols == 0 || cols % 256 != 0 || cols > 8192) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "rows_to_qb32_dev: m > 0 and cols a multiple of 256 up to 8192, got m=%zu cols=%zu", m, cols);
    hipError_t e = launch_rows_to_qb32(x_dev, gamma_dev, m, cols, qb_dev, stats_dev, (hipStream_t)stream);
    if (e != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "kernel launch failed: %s", hipGetErrorString(e));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_matmul_qb32_supported(bitnet_hip_weights_t h) {
    const WeightsRef w = lookup(h);
    return w && gemm_qb32_supported(*w) ? 1 : 0;
}

int bitnet_hip_matmul_qb32_dev(bitnet_hip_weights_t h, const void *qb_dev, size_t m, const float *stats_in_dev, size_t n_stats,
                               const float *ln_gamma_dev, float ln_eps, float *y_dev, const float *residual_dev, int flags, void *yh_dev,
                               const float *gamma_out_dev, float *stats_out_dev, void *stream) {
    BH_GUARD_BEGIN
    const WeightsRef w = lookup(h);
    if (!w) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown weights handle %llu", (unsigned long long)h);
    if (!qb_dev || (!y_dev && !yh_dev)) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to matmul_qb32_dev");
    if (m == 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "dimensions must be > 0: m=0");
    if (!gemm_qb32_supported(*w))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "matmul_qb32_dev: needs an unscaled matrix with a code map in -2..2, rows %% 256 == 0, cols %% 256 == 0");
    if ((ln_gamma_dev != nullptr) != (stats_in_dev != nullptr) || (ln_gamma_dev && (!w->ln_g || w->ln_gamma_bound != ln_gamma_dev || n_stats == 0)))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "matmul_qb32_dev: LayerNorm needs the bound gamma (bitnet_hip_weights_bind_ln) and its statistics partials");
    if (flags & ~BITNET_HIP_FUSE_SILU_MUL) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "matmul_qb32_dev: unknown flag bits 0x%x", flags);
    GemmF16Io io;
    io.xh = qb_dev;
    io.stats_in = stats_in_dev;
    io.n_stats = (int)n_stats;
    io.ln_eps = ln_eps;
    io.y = y_dev;
    io.residual = residual_dev;
    io.silu_mul = (flags & BITNET_HIP_FUSE_SILU_MUL) != 0;
    io.yh = yh_dev;
    io.gamma_out = gamma_out_dev;
    io.stats_out = stats_out_dev;
    if (io.silu_mul && (!w->paired || residual_dev))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "FUSE_SILU_MUL needs a handle from weights_concat(..., interleave16=1) and no residual");
    if (gemm_fp4_resident_enabled() && !w->tiles4) {
        const hipError_t ei = ensure_fp4_image(*w, (hipStream_t)stream);
        if (ei != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "building the resident fp4 weight image failed: %s", hipGetErrorString(ei));
    }
    hipError_t e = launch_gemm_qb32(*w, io, m, (hipStream_t)stream);
    if (e != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "kernel launch failed: %s", hipGetErrorString(e));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_matmul_last_tile(int *digits, int *wave_tokens, int *waves, int *scale_mode) {
    const GemmTileChoice &t = g_last_gemm_tile;
    if (digits) *digits = t.digits;
    if (wave_tokens) *wave_tokens = t.wave_tokens;
    if (waves) *waves = t.waves;
    if (scale_mode) *scale_mode = t.scale_mode;
    return t.digits ? BITNET_HIP_OK : set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "no tiled matmul has been launched on this thread");
}

int bitnet_hip_matmul_last_wave_rows(void) { return g_last_gemm_tile.digits ? g_last_gemm_tile.wave_rows : 0; }

int bitnet_hip_matmul_last_resident_fp4(void) { return g_last_gemm_tile.resident_fp4; }

int bitnet_hip_gemv_fused_dev(bitnet_hip_weights_t h, const float *x_dev, float *y_dev, size_t m,
                              const float *ln_gamma_dev, float ln_eps, const float *residual_dev, int flags,
                              void *stream) {
    BH_GUARD_BEGIN
    const WeightsRef w = lookup(h);
    if (!w) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown weights handle %llu", (unsigned long long)h);
    if (!x_dev || !y_dev) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to gemv_fused_dev");
    if (m == 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "dimensions must be > 0: m=0");
    GemvFusion fu;
    fu.ln_gamma = ln_gamma_dev;
    fu.ln_eps = ln_eps;
    fu.residual = residual_dev;
    fu.silu_mul = (flags & BITNET_HIP_FUSE_SILU_MUL) != 0;
    if (fu.silu_mul && (!w->paired || residual_dev))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT,
                         "FUSE_SILU_MUL needs a handle from weights_concat(..., interleave16=1) and no residual");
    return run_gemv(*w, x_dev, y_dev, m, fu, (hipStream_t)stream);
    BH_GUARD_END
}

int bitnet_hip_weights_bind_ln(bitnet_hip_weights_t h, const float *ln_gamma_dev, void *stream) {
    BH_GUARD_BEGIN
    const WeightsRef w = lookup(h);
    if (!w) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown weights handle %llu", (unsigned long long)h);
    if (!ln_gamma_dev) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to weights_bind_ln");
    if (!mfma_supported(*w)) return BITNET_HIP_OK;  // other kernels keep the prologue form
    if (!w->ln_g && hipMalloc((void **)&w->ln_g, w->rows * sizeof(float)) != hipSuccess)
        return set_error(BITNET_HIP_ERR_GPU, "hipMalloc failed in weights_bind_ln");
    w->ln_gamma_bound = nullptr;
    hipError_t e = build_tiles(*w, (hipStream_t)stream);
    if (e == hipSuccess) e = launch_gemv_mfma(*w, ln_gamma_dev, w->ln_g, 1, GemvFusion(), (hipStream_t)stream);  // g = W . gamma, stored row order
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "weights_bind_ln failed: %s", hipGetErrorString(e));
    w->ln_gamma_bound = ln_gamma_dev;
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_weights_concat(const bitnet_hip_weights_t *parts, size_t n_parts, int interleave16,
                              bitnet_hip_weights_t *out) {
    BH_GUARD_BEGIN
    if (!parts || !out || n_parts == 0)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to weights_concat");
    std::vector<WeightsRef> ws;
    size_t rows = 0;
    for (size_t i = 0; i < n_parts; ++i) {
        const WeightsRef w = lookup(parts[i]);
        if (!w) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown weights handle %llu", (unsigned long long)parts[i]);
        if (i > 0 && (w->cols != ws[0]->cols || w->row_stride_bytes != ws[0]->row_stride_bytes || w->lut != ws[0]->lut ||
                      w->block_size != ws[0]->block_size || w->scaled != ws[0]->scaled))
            return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "weights_concat: part %zu differs in cols / code map / scales", i);
        ws.push_back(w);
        rows += w->rows;
    }
    if (interleave16 && (n_parts != 2 || ws[0]->rows != ws[1]->rows || ws[0]->rows % 16 != 0))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "weights_concat: interleave16 needs two parts of equal rows %% 16 == 0");
    Weights *f = new Weights();
    *f = *ws[0];
    f->mu = std::make_shared<std::mutex>();
    f->ref_pins = std::make_shared<std::atomic<int>>(0);
    f->codes = nullptr;
    f->scales = nullptr;
    f->tiles = nullptr;
    f->scale_tiles = nullptr;
    f->scale_tiles_h = nullptr;
    f->ln_g = nullptr;
    f->ln_gamma_bound = nullptr;
    for (const WeightsRef &w : ws) f->scales_f16 = f->scales_f16 && w->scales_f16;
    for (const WeightsRef &w : ws) f->scales_f16_x2_finite = f->scales_f16_x2_finite && w->scales_f16_x2_finite;
    f->rows = rows;
    f->paired = interleave16 != 0;
    f->algorithmic_bytes = 0;
    for (const WeightsRef &w : ws) {
        f->algorithmic_bytes += w->algorithmic_bytes;
        if (w->scales_f16 && !f->scales_f16) f->algorithmic_bytes += 2 * w->rows * w->nblk;  // stored as f32 in the fused matrix
    }
    // Tile path: every part already lives as 16-row tiles (rows % 16 == 0, except that the last part of a plain
    // concatenation may be ragged) with one kind of scale tiles -> the fused matrix is the tile arrays laid end to end
    // (or alternating), no reference-layout copy is ever made.
    const size_t nblk256 = div_ceil(f->cols, 256);
    bool tile_path = true;
    for (size_t i = 0; i < ws.size(); ++i) {
        const Weights &w = *ws[i];
        if (!w.tiles || (w.rows % 16 != 0 && (interleave16 || i + 1 != ws.size()))) tile_path = false;
        if (w.scaled && w.block_size == 32 && !(f->scales_f16 ? w.scale_tiles_h != nullptr : w.scale_tiles != nullptr)) tile_path = false;
        if (w.scaled && w.block_size != 32) tile_path = false;
    }
    bool ok = true;
    if (tile_path) {
        const size_t trb = nblk256 * 1024, n_tiles = div_ceil(rows, 16);
        const bool st = f->scaled, sth = st && f->scales_f16;
        const size_t srb = nblk256 * 128 * (sth ? sizeof(uint16_t) : sizeof(float));
        ok = hipMalloc((void **)&f->tiles, n_tiles * trb) == hipSuccess;
        uint8_t *sdst = nullptr;
        if (ok && st) {
            ok = hipMalloc((void **)&sdst, n_tiles * srb) == hipSuccess;
            if (sth)
                f->scale_tiles_h = reinterpret_cast<uint16_t *>(sdst);
            else
                f->scale_tiles = reinterpret_cast<float *>(sdst);
        }
        auto ssrc = [&](const Weights &w) { return sth ? (const uint8_t *)w.scale_tiles_h : (const uint8_t *)w.scale_tiles; };
        if (ok && interleave16) {
            const size_t nt = ws[0]->rows / 16;
            for (int i = 0; i < 2 && ok; ++i) {
                ok = hipMemcpy2D(f->tiles + i * trb, 2 * trb, ws[i]->tiles, trb, trb, nt, hipMemcpyDeviceToDevice) == hipSuccess;
                if (ok && st) ok = hipMemcpy2D(sdst + i * srb, 2 * srb, ssrc(*ws[i]), srb, srb, nt, hipMemcpyDeviceToDevice) == hipSuccess;
            }
        } else if (ok) {
            size_t t0 = 0;
            for (const WeightsRef &w : ws) {
                const size_t nt = div_ceil(w->rows, 16);
                ok = ok && hipMemcpy(f->tiles + t0 * trb, w->tiles, nt * trb, hipMemcpyDeviceToDevice) == hipSuccess;
                if (ok && st) ok = hipMemcpy(sdst + t0 * srb, ssrc(*w), nt * srb, hipMemcpyDeviceToDevice) == hipSuccess;
                t0 += nt;
            }
        }
        f->n_row_tiles = n_tiles;
        f->n_kblocks = nblk256;
        if (!ok) {
            free_weights(f);
            return set_error(BITNET_HIP_ERR_GPU, "weights_concat: device allocation or copy failed");
        }
        return register_weights(f, out);
    }
    // Reference path (odd shapes): the parts' reference layouts are rebuilt if they were dropped
    std::vector<std::unique_ptr<ReferencePin>> pins;  // the copies below are synchronous: the pins go when this block is left
    for (const WeightsRef &w : ws) {
        pins.emplace_back(new ReferencePin(*w, nullptr));
        if (pins.back()->status != hipSuccess) ok = false;
    }
    const size_t stride = f->row_stride_bytes, sstride = f->nblk * sizeof(float);
    ok = ok && hipMalloc((void **)&f->codes, rows * stride + 16) == hipSuccess;
    if (ok && f->scaled) ok = hipMalloc((void **)&f->scales, rows * sstride) == hipSuccess;
    if (ok && interleave16) {
        const size_t nt = ws[0]->rows / 16;
        for (int i = 0; i < 2 && ok; ++i) {
            ok = hipMemcpy2D(f->codes + i * 16 * stride, 32 * stride, ws[i]->codes, 16 * stride, 16 * stride, nt,
                             hipMemcpyDeviceToDevice) == hipSuccess;
            if (ok && f->scales)
                ok = hipMemcpy2D((uint8_t *)f->scales + i * 16 * sstride, 32 * sstride, ws[i]->scales, 16 * sstride,
                                 16 * sstride, nt, hipMemcpyDeviceToDevice) == hipSuccess;
        }
    } else if (ok) {
        size_t r0 = 0;
        for (const WeightsRef &w : ws) {
            ok = ok && hipMemcpy(f->codes + r0 * stride, w->codes, w->rows * stride, hipMemcpyDeviceToDevice) == hipSuccess;
            if (ok && f->scales)
                ok = hipMemcpy((uint8_t *)f->scales + r0 * sstride, w->scales, w->rows * sstride, hipMemcpyDeviceToDevice) == hipSuccess;
            r0 += w->rows;
        }
    }
    pins.clear();
    for (const WeightsRef &w : ws)
        if (mfma_supported(*w)) trim_reference(*w);
    if (ok && mfma_supported(*f)) ok = build_tiles(*f, nullptr) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    if (!ok) {
        free_weights(f);
        return set_error(BITNET_HIP_ERR_GPU, "weights_concat: device allocation or copy failed");
    }
    if (mfma_supported(*f)) trim_reference(*f);
    return register_weights(f, out);
    BH_GUARD_END
}

size_t bitnet_hip_weights_device_bytes(bitnet_hip_weights_t h) {
    const WeightsRef w = lookup(h);
    if (!w) return 0;
    std::lock_guard<std::mutex> lk(*w->mu);
    return weights_device_bytes(*w);
}

int bitnet_hip_weights_trim(bitnet_hip_weights_t h) {
    const WeightsRef w = lookup(h);
    if (!w) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown weights handle %llu", (unsigned long long)h);
    trim_reference(*w);
    return BITNET_HIP_OK;
}

/* ------------------------------------------------ decode-step operators */

int bitnet_hip_norm_rows_dev(const float *x_dev, const float *gamma_dev, float *out_dev, size_t rows, size_t hidden,
                             float eps, int rms, void *stream) {
    BH_GUARD_BEGIN
    if (!x_dev || !gamma_dev || !out_dev) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to norm_rows_dev");
    if (hidden == 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "hidden_dim must be > 0");
    BH_HIP_TRY(launch_norm_rows(x_dev, gamma_dev, out_dev, (int)rows, (int)hidden, eps, rms != 0, (hipStream_t)stream));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_rmsnorm(const float *input, size_t in_len, const float *gamma, size_t gamma_len, float *output,
                       size_t out_len, size_t num_rows, size_t hidden_dim, float eps) {
    BH_GUARD_BEGIN
    if (!input || !gamma || !output) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to rmsnorm");
    if (hidden_dim == 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "hidden_dim must be > 0");
    BH_CHECK_DIMS(num_rows, hidden_dim);
    if (in_len < num_rows * hidden_dim || out_len < num_rows * hidden_dim || gamma_len < hidden_dim)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "rmsnorm: buffer too small for %zu rows x %zu", num_rows, hidden_dim);
    if (num_rows == 0) return BITNET_HIP_OK;
    int rc = ensure_init();
    if (rc) return rc;
    DevBuf xd, gd, od;
    const size_t n = num_rows * hidden_dim * 4;
    if (xd.alloc(n) != hipSuccess || gd.alloc(hidden_dim * 4) != hipSuccess || od.alloc(n) != hipSuccess)
        return set_error(BITNET_HIP_ERR_GPU, "hipMalloc failed in rmsnorm");
    BH_HIP_TRY(hipMemcpy(xd.p, input, n, hipMemcpyHostToDevice));
    BH_HIP_TRY(hipMemcpy(gd.p, gamma, hidden_dim * 4, hipMemcpyHostToDevice));
    BH_HIP_TRY(launch_norm_rows(xd.as<float>(), gd.as<float>(), od.as<float>(), (int)num_rows, (int)hidden_dim, eps, true, nullptr));
    BH_HIP_TRY(hipMemcpy(output, od.p, n, hipMemcpyDeviceToHost));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_advance_pos_dev(int32_t *pos_dev, void *stream) {
    BH_GUARD_BEGIN
    if (!pos_dev) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to advance_pos_dev");
    BH_HIP_TRY(launch_advance_pos(pos_dev, (hipStream_t)stream));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_add_dev(const float *a_dev, const float *b_dev, float *out_dev, size_t n, void *stream) {
    BH_GUARD_BEGIN
    if (!a_dev || !b_dev || !out_dev) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to add_dev");
    if (n == 0) return BITNET_HIP_OK;
    BH_HIP_TRY(launch_add(a_dev, b_dev, out_dev, n, (hipStream_t)stream));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_silu_mul_dev(const float *gate_dev, const float *up_dev, float *out_dev, size_t n, size_t tile, void *stream) {
    BH_GUARD_BEGIN
    if (!gate_dev || !up_dev || !out_dev) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to silu_mul_dev");
    if (n == 0) return BITNET_HIP_OK;
    if (tile != 0 && n % tile != 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "silu_mul: n %zu is not a multiple of the tile %zu", n, tile);
    BH_HIP_TRY(launch_silu_mul(gate_dev, up_dev, out_dev, n, tile, (hipStream_t)stream));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_embed_f16_dev(const void *table, const int32_t *tokens_dev, const int32_t *offset_dev, size_t n,
                             size_t hidden, size_t vocab, float *out_dev, void *stream) {
    BH_GUARD_BEGIN
    if (!table || !tokens_dev || !out_dev) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to embed_f16_dev");
    if (hidden % 8 != 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "embed: hidden %zu must be a multiple of 8", hidden);
    BH_HIP_TRY(launch_embed_f16(table, tokens_dev, offset_dev, (int)n, (int)hidden, (int)vocab, out_dev, (hipStream_t)stream));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_attention_decode_dev(const float *qkv, const float *rope_sin, const float *rope_cos, float *kcache,
                                    float *vcache, size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_pos,
                                    const int32_t *pos_dev, float *scratch, float *out, void *stream) {
    BH_GUARD_BEGIN
    if (!qkv || !rope_sin || !rope_cos || !kcache || !vcache || !pos_dev || !scratch || !out)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to attention_decode_dev");
    if (n_kv_heads == 0 || n_heads % n_kv_heads != 0)  // T:215-220
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "num_heads %zu must be divisible by num_key_value_heads %zu", n_heads, n_kv_heads);
    if (head_dim != 128 || n_heads / n_kv_heads > 4)
        return set_error(BITNET_HIP_ERR_UNSUPPORTED, "attention_decode: head_dim %zu / group %zu unsupported (head_dim 128, group <= 4)",
                         head_dim, n_heads / n_kv_heads);
    BH_HIP_TRY(launch_attn_decode(qkv, rope_sin, rope_cos, kcache, vcache, (int)n_heads, (int)n_kv_heads, (int)head_dim,
                                  (int)max_pos, pos_dev, scratch, out, (hipStream_t)stream));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_attention_decode_wide_dev(const float *qkv, const float *rope_sin, const float *rope_cos, float *kcache,
                                         float *vcache, size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_pos,
                                         const int32_t *pos_dev, float *scratch, float *out, void *stream) {
    BH_GUARD_BEGIN
    if (!qkv || !rope_sin || !rope_cos || !kcache || !vcache || !pos_dev || !scratch || !out)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to attention_decode_wide_dev");
    if (n_kv_heads == 0 || n_heads % n_kv_heads != 0)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "num_heads %zu must be divisible by num_key_value_heads %zu", n_heads, n_kv_heads);
    if (head_dim != 128 || n_heads / n_kv_heads > 4)
        return set_error(BITNET_HIP_ERR_UNSUPPORTED, "attention_decode: head_dim %zu / group %zu unsupported (head_dim 128, group <= 4)",
                         head_dim, n_heads / n_kv_heads);
    BH_HIP_TRY(launch_attn_decode(qkv, rope_sin, rope_cos, kcache, vcache, (int)n_heads, (int)n_kv_heads, (int)head_dim,
                                  (int)max_pos, pos_dev, scratch, out, (hipStream_t)stream, true, 2));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_attention_decode_partial_dev(const float *qkv, const float *rope_sin, const float *rope_cos, float *kcache,
                                            float *vcache, size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_pos,
                                            const int32_t *pos_dev, float *scratch, void *stream) {
    BH_GUARD_BEGIN
    if (!qkv || !rope_sin || !rope_cos || !kcache || !vcache || !pos_dev || !scratch)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to attention_decode_partial_dev");
    if (n_kv_heads == 0 || n_heads % n_kv_heads != 0)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "num_heads %zu must be divisible by num_key_value_heads %zu", n_heads, n_kv_heads);
    if (head_dim != 128 || n_heads / n_kv_heads > 4)
        return set_error(BITNET_HIP_ERR_UNSUPPORTED, "attention_decode: head_dim %zu / group %zu unsupported (head_dim 128, group <= 4)",
                         head_dim, n_heads / n_kv_heads);
    BH_HIP_TRY(launch_attn_decode(qkv, rope_sin, rope_cos, kcache, vcache, (int)n_heads, (int)n_kv_heads, (int)head_dim,
                                  (int)max_pos, pos_dev, scratch, nullptr, (hipStream_t)stream, false));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

size_t bitnet_hip_attention_merge_max_keys(void) { return (size_t)4 * 64; }

static int gemv_attn_merge(bitnet_hip_weights_t h, const float *attn_scratch_dev, size_t n_heads, size_t n_kv_heads, size_t max_pos,
                           const int32_t *pos_dev, float *y_dev, const float *residual_dev, void *qact_out, const float *gamma_out_dev,
                           double *stats_out, void *stream, int chunk_log2 = 6) {
    BH_GUARD_BEGIN
    const WeightsRef w = lookup(h);
    if (!w) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown weights handle %llu", (unsigned long long)h);
    if (!attn_scratch_dev || !pos_dev || !y_dev) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to gemv_attn_merge_dev");
    const size_t group = n_kv_heads ? n_heads / n_kv_heads : 0;
    if (n_kv_heads == 0 || n_heads % n_kv_heads != 0 || (group != 1 && group != 2 && group != 4) || w->cols != n_heads * 128)
        return set_error(BITNET_HIP_ERR_UNSUPPORTED, "gemv_attn_merge_dev: needs cols == n_heads * 128 and a query group of 1, 2 or 4 (got %zu heads / %zu)",
                         n_heads, n_kv_heads);
    if (!mfma_supported(*w) || div_ceil(div_ceil(w->cols, 256), (size_t)8) > 2)
        return set_error(BITNET_HIP_ERR_UNSUPPORTED, "gemv_attn_merge_dev: matrix shape %zux%zu not supported", w->rows, w->cols);
    if (qact_out && w->rows % 16 != 0)
        return set_error(BITNET_HIP_ERR_UNSUPPORTED, "gemv_attn_merge_q_dev: rows %zu must be a multiple of 16 for a QAct output", w->rows);
    if (qact_out && gemvq_supported(*w) && w->cols <= 4096) {
        // QAct consumer form: the workgroup merges the records and quantises them once, into its LDS image (kernels_gemvq.hip MRG)
        GemvQIo io;
        io.residual = residual_dev;
        io.y = y_dev;
        io.qout = qact_out;
        io.gamma_out = gamma_out_dev;
        io.stats_out = stats_out;
        io.attn_rec = attn_scratch_dev;
        io.attn_pos = pos_dev;
        io.attn_chunk_log2 = chunk_log2;
        io.attn_chunks_max = (int)div_ceil(max_pos, (size_t)1 << chunk_log2);
        io.attn_group_log2 = group == 4 ? 2 : group == 2 ? 1 : 0;
        hipError_t e = launch_gemv_q(*w, io, (hipStream_t)stream);
        if (e != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "kernel launch failed: %s", hipGetErrorString(e));
        return BITNET_HIP_OK;
    }
    GemvFusion fu;
    fu.residual = residual_dev;
    fu.attn_rec = attn_scratch_dev;
    fu.attn_pos = pos_dev;
    fu.attn_chunk_log2 = chunk_log2;
    fu.attn_chunks_max = (int)div_ceil(max_pos, (size_t)1 << fu.attn_chunk_log2);
    fu.attn_group_log2 = group == 4 ? 2 : group == 2 ? 1 : 0;
    fu.qout = qact_out;
    fu.gamma_out = gamma_out_dev;
    fu.stats_out = stats_out;
    return run_gemv(*w, attn_scratch_dev, y_dev, 1, fu, (hipStream_t)stream, BITNET_HIP_KERNEL_MFMA);
    BH_GUARD_END
}

int bitnet_hip_gemv_attn_merge_dev(bitnet_hip_weights_t h, const float *attn_scratch_dev, size_t n_heads, size_t n_kv_heads,
                                   size_t max_pos, const int32_t *pos_dev, float *y_dev, const float *residual_dev, void *stream) {
    return gemv_attn_merge(h, attn_scratch_dev, n_heads, n_kv_heads, max_pos, pos_dev, y_dev, residual_dev, nullptr, nullptr, nullptr, stream);
}

int bitnet_hip_gemv_attn_merge_q_dev(bitnet_hip_weights_t h, const float *attn_scratch_dev, size_t n_heads, size_t n_kv_heads,
                                     size_t max_pos, const int32_t *pos_dev, float *y_dev, const float *residual_dev, void *qact_out,
                                     const float *gamma_out_dev, double *stats_out, void *stream) {
    if (!qact_out) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to gemv_attn_merge_q_dev");
    return gemv_attn_merge(h, attn_scratch_dev, n_heads, n_kv_heads, max_pos, pos_dev, y_dev, residual_dev, qact_out, gamma_out_dev, stats_out, stream);
}

/* ---- QAct: activations quantised by their producer (csrc/qact.hpp) ---- */

size_t bitnet_hip_qact_bytes(size_t cols) { return qact_bytes(cols); }
size_t bitnet_hip_qact_stats_bytes(size_t cols) { return div_ceil(cols, 16) * 2 * sizeof(double); }

int bitnet_hip_quantize_act_dev(const float *x_dev, const float *gamma_dev, size_t cols, void *qact_out, double *stats_out, void *stream) {
    BH_GUARD_BEGIN
    if (!x_dev || !qact_out) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to quantize_act_dev");
    if (cols == 0 || cols % 16 != 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "quantize_act: cols %zu must be a positive multiple of 16", cols);
    BH_HIP_TRY(launch_quant_act(x_dev, gamma_dev, cols, qact_out, stats_out, (hipStream_t)stream));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_embed_q_dev(const void *table, const int32_t *tokens_dev, const int32_t *offset_dev, size_t hidden, size_t vocab,
                           float *x_out_dev, const float *gamma_dev, void *qact_out, double *stats_out, void *stream) {
    BH_GUARD_BEGIN
    if (!table || !tokens_dev || !x_out_dev || !qact_out) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to embed_q_dev");
    if (hidden == 0 || hidden % 16 != 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "embed_q: hidden %zu must be a positive multiple of 16", hidden);
    BH_HIP_TRY(launch_embed_q(table, tokens_dev, offset_dev, (int)hidden, (int)vocab, x_out_dev, gamma_dev, qact_out, stats_out, (hipStream_t)stream));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_gemv_q_supported(bitnet_hip_weights_t h) {
    const WeightsRef w = lookup(h);
    return w && gemvq_supported(*w) ? 1 : 0;
}

int bitnet_hip_gemv_q_dev(bitnet_hip_weights_t h, const void *qact_in, const double *stats_in, const float *ln_gamma_dev, float ln_eps,
                          const float *residual_dev, int flags, float *y_dev, void *qact_out, const float *gamma_out_dev, double *stats_out,
                          void *stream) {
    BH_GUARD_BEGIN
    const WeightsRef w = lookup(h);
    if (!w) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "unknown weights handle %llu", (unsigned long long)h);
    if (!qact_in || (!y_dev && !qact_out)) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to gemv_q_dev");
    if (!gemvq_supported(*w))
        return set_error(BITNET_HIP_ERR_UNSUPPORTED, "gemv_q_dev: matrix %zux%zu (block %zu) is not on the QAct path (cols %% 256 == 0, rows %% 16 == 0, no scales or 32-element blocks)",
                         w->rows, w->cols, w->block_size);
    GemvQIo io;
    io.qin = qact_in;
    io.stats_in = stats_in;
    io.ln_gamma = ln_gamma_dev;
    io.ln_eps = ln_eps;
    io.residual = residual_dev;
    io.silu_mul = (flags & BITNET_HIP_FUSE_SILU_MUL) != 0;
    io.y = y_dev;
    io.qout = qact_out;
    io.gamma_out = gamma_out_dev;
    io.stats_out = stats_out;
    if (io.silu_mul && (!w->paired || residual_dev || w->rows % 32 != 0))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "FUSE_SILU_MUL needs a handle from weights_concat(..., interleave16=1) and no residual");
    if (ln_gamma_dev && (!stats_in || !w->ln_g || w->ln_gamma_bound != ln_gamma_dev))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "gemv_q_dev: LayerNorm needs stats_in and the gamma bound with bitnet_hip_weights_bind_ln");
    if (ln_gamma_dev && w->cols > 4096)
        return set_error(BITNET_HIP_ERR_UNSUPPORTED, "gemv_q_dev: LayerNorm input of %zu columns (<= 4096)", w->cols);
    hipError_t e = launch_gemv_q(*w, io, (hipStream_t)stream);
    if (e != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "kernel launch failed: %s", hipGetErrorString(e));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_attention_decode_q_dev(const float *qkv, const float *rope_sin, const float *rope_cos, void *kcache, void *vcache,
                                      size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_pos, const int32_t *pos_dev,
                                      float *scratch, int flags, float *out, void *qact_out, void *stream) {
    const int wide = (flags & BITNET_HIP_ATTN_WIDE) != 0, kv16 = (flags & BITNET_HIP_ATTN_KV_F16) != 0, partial = (flags & BITNET_HIP_ATTN_PARTIAL) != 0;
    BH_GUARD_BEGIN
    if (!qkv || !rope_sin || !rope_cos || !kcache || !vcache || !pos_dev || !scratch || (!out && !qact_out && !partial))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to attention_decode_q_dev");
    if (n_kv_heads == 0 || n_heads % n_kv_heads != 0)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "num_heads %zu must be divisible by num_key_value_heads %zu", n_heads, n_kv_heads);
    if (head_dim != 128 || n_heads / n_kv_heads > 4)
        return set_error(BITNET_HIP_ERR_UNSUPPORTED, "attention_decode: head_dim %zu / group %zu unsupported (head_dim 128, group <= 4)",
                         head_dim, n_heads / n_kv_heads);
    BH_HIP_TRY(launch_attn_decode(qkv, rope_sin, rope_cos, static_cast<float *>(kcache), static_cast<float *>(vcache), (int)n_heads, (int)n_kv_heads,
                                  (int)head_dim, (int)max_pos, pos_dev, scratch, out, (hipStream_t)stream, !partial, wide ? 2 : 1, qact_out, kv16));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

size_t bitnet_hip_attention_prefill_workspace_bytes(size_t n_heads, size_t n_kv_heads, size_t seq_len) {
    return attn_prefill_workspace_bytes((int)n_heads, (int)n_kv_heads, (int)seq_len, (int)seq_len);
}

size_t bitnet_hip_attention_prefill_sharded_workspace_bytes(size_t n_heads, size_t n_kv_heads, size_t n_q, size_t n_ctx) {
    return attn_prefill_workspace_bytes((int)n_heads, (int)n_kv_heads, (int)n_q, (int)n_ctx);
}

static int check_prefill_args(const void *a, const void *b, const void *rs, const void *rc, const void *kc, const void *vc,
                              const void *ws, const void *out, size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_pos,
                              size_t n_ctx) {
    if (!a || !b || !rs || !rc || !kc || !vc || !ws || !out)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to attention_prefill");
    if (n_kv_heads == 0 || n_heads % n_kv_heads != 0)  // T:215-220
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "num_heads %zu must be divisible by num_key_value_heads %zu", n_heads, n_kv_heads);
    if (head_dim != 128) return set_error(BITNET_HIP_ERR_UNSUPPORTED, "attention_prefill: head_dim %zu unsupported (128)", head_dim);
    if (n_ctx == 0 || n_ctx > max_pos)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "KV cache overflow: seq_len %zu, max_pos %zu", n_ctx, max_pos);  // T:1190-1194
    return BITNET_HIP_OK;
}

int bitnet_hip_attention_prefill_dev(const float *qkv, const float *rope_sin, const float *rope_cos, float *kcache,
                                     float *vcache, size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_pos,
                                     size_t seq_len, void *workspace, size_t workspace_bytes, float *out, void *stream) {
    BH_GUARD_BEGIN
    int rc = check_prefill_args(qkv, qkv, rope_sin, rope_cos, kcache, vcache, workspace, out, n_heads, n_kv_heads, head_dim, max_pos, seq_len);
    if (rc) return rc;
    const size_t need = attn_prefill_workspace_bytes((int)n_heads, (int)n_kv_heads, (int)seq_len, (int)seq_len);
    if (workspace_bytes < need)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "workspace too small: expected %zu, got %zu", need, workspace_bytes);
    const int ld = (int)((n_heads + 2 * n_kv_heads) * head_dim);
    BH_HIP_TRY(launch_attn_prefill(qkv, ld, nullptr, (int)seq_len, qkv + n_heads * head_dim, ld, (int)seq_len, rope_sin, rope_cos, kcache,
                                   vcache, (int)n_heads, (int)n_kv_heads, (int)head_dim, (int)max_pos, workspace, workspace_bytes, out,
                                   (hipStream_t)stream));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_attention_prefill_sharded_dev(const float *q, size_t ld_q, const int32_t *q_block_pos, size_t n_q, const float *kv,
                                             size_t ld_kv, size_t n_ctx, const float *rope_sin, const float *rope_cos, float *kcache,
                                             float *vcache, size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_pos,
                                             void *workspace, size_t workspace_bytes, float *out, void *stream) {
    BH_GUARD_BEGIN
    int rc = check_prefill_args(q, kv, rope_sin, rope_cos, kcache, vcache, workspace, out, n_heads, n_kv_heads, head_dim, max_pos, n_ctx);
    if (rc) return rc;
    if (n_q == 0 || !q_block_pos) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "attention_prefill_sharded: n_q and q_block_pos must be given");
    if (ld_q < n_heads * head_dim || ld_kv < 2 * n_kv_heads * head_dim)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "attention_prefill_sharded: row strides too small (ld_q %zu, ld_kv %zu)", ld_q, ld_kv);
    const size_t need = attn_prefill_workspace_bytes((int)n_heads, (int)n_kv_heads, (int)n_q, (int)n_ctx);
    if (workspace_bytes < need)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "workspace too small: expected %zu, got %zu", need, workspace_bytes);
    BH_HIP_TRY(launch_attn_prefill(q, (int)ld_q, q_block_pos, (int)n_q, kv, (int)ld_kv, (int)n_ctx, rope_sin, rope_cos, kcache, vcache,
                                   (int)n_heads, (int)n_kv_heads, (int)head_dim, (int)max_pos, workspace, workspace_bytes, out,
                                   (hipStream_t)stream));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_attention_prefill_kv16_dev(const float *qkv, const float *rope_sin, const float *rope_cos, void *kcache_f16, void *vcache_f16,
                                          size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_pos, size_t seq_len, void *workspace,
                                          size_t workspace_bytes, float *out, void *stream) {
    BH_GUARD_BEGIN
    int rc = check_prefill_args(qkv, qkv, rope_sin, rope_cos, kcache_f16, vcache_f16, workspace, out, n_heads, n_kv_heads, head_dim, max_pos, seq_len);
    if (rc) return rc;
    const size_t need = attn_prefill_workspace_bytes((int)n_heads, (int)n_kv_heads, (int)seq_len, (int)seq_len);
    if (workspace_bytes < need)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "workspace too small: expected %zu, got %zu", need, workspace_bytes);
    const int ld = (int)((n_heads + 2 * n_kv_heads) * head_dim);
    BH_HIP_TRY(launch_attn_prefill(qkv, ld, nullptr, (int)seq_len, qkv + n_heads * head_dim, ld, (int)seq_len, rope_sin, rope_cos, static_cast<float *>(kcache_f16),
                                   static_cast<float *>(vcache_f16), (int)n_heads, (int)n_kv_heads, (int)head_dim, (int)max_pos, workspace, workspace_bytes, out,
                                   (hipStream_t)stream, 0, 0, 1));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_attention_prefill_flags_dev(const float *qkv, const float *rope_sin, const float *rope_cos, void *kcache, void *vcache, size_t n_heads,
                                           size_t n_kv_heads, size_t head_dim, size_t max_pos, size_t seq_len, void *workspace, size_t workspace_bytes,
                                           void *out, int flags, void *stream) {
    BH_GUARD_BEGIN
    int rc = check_prefill_args(qkv, qkv, rope_sin, rope_cos, kcache, vcache, workspace, out, n_heads, n_kv_heads, head_dim, max_pos, seq_len);
    if (rc) return rc;
    if (flags & ~(BITNET_HIP_ATTN_CACHE_F16 | BITNET_HIP_ATTN_OUT_F16)) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "attention_prefill_flags_dev: unknown flag bits 0x%x", flags);
    const size_t need = attn_prefill_workspace_bytes((int)n_heads, (int)n_kv_heads, (int)seq_len, (int)seq_len);
    if (workspace_bytes < need)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "workspace too small: expected %zu, got %zu", need, workspace_bytes);
    const int ld = (int)((n_heads + 2 * n_kv_heads) * head_dim);
    BH_HIP_TRY(launch_attn_prefill(qkv, ld, nullptr, (int)seq_len, qkv + n_heads * head_dim, ld, (int)seq_len, rope_sin, rope_cos, static_cast<float *>(kcache),
                                   static_cast<float *>(vcache), (int)n_heads, (int)n_kv_heads, (int)head_dim, (int)max_pos, workspace, workspace_bytes,
                                   static_cast<float *>(out), (hipStream_t)stream, 0, 0, flags));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_attention_prefill_gathered_dev(const float *q, size_t ld_q, const int32_t *q_block_pos, size_t n_q, const void *kv_gathered,
                                              size_t n_ctx, size_t world, int kv_is_f16, const float *rope_sin, const float *rope_cos,
                                              void *kcache, void *vcache, int cache_f16, size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_pos,
                                              void *workspace, size_t workspace_bytes, float *out, void *stream) {
    // cache_f16 is a BOOLEAN on this entry (any non-zero value = f16 caches, as before the phase entry gave the word its flag bits)
    return bitnet_hip_attention_prefill_gathered_phase_dev(q, ld_q, q_block_pos, n_q, kv_gathered, n_ctx, world, kv_is_f16, rope_sin, rope_cos, kcache, vcache,
                                                           cache_f16 ? BITNET_HIP_ATTN_CACHE_F16 : 0, n_heads, n_kv_heads, head_dim, max_pos, workspace, workspace_bytes, out, 0, stream);
}

int bitnet_hip_attention_prefill_gathered_phase_dev(const float *q, size_t ld_q, const int32_t *q_block_pos, size_t n_q, const void *kv_gathered,
                                                    size_t n_ctx, size_t world, int kv_is_f16, const float *rope_sin, const float *rope_cos,
                                                    void *kcache, void *vcache, int cache_f16, size_t n_heads, size_t n_kv_heads, size_t head_dim,
                                                    size_t max_pos, void *workspace, size_t workspace_bytes, float *out, int phase, void *stream) {
    BH_GUARD_BEGIN
    if (phase < 0 || phase > 2) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "attention_prefill_gathered: phase must be 0, 1 or 2, got %d", phase);
    if (cache_f16 & ~(BITNET_HIP_ATTN_CACHE_F16 | BITNET_HIP_ATTN_OUT_F16))  // a flag word here: unknown bits are refused, as attention_prefill_flags_dev does
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "attention_prefill_gathered_phase: unknown flag bits 0x%x (BITNET_HIP_ATTN_CACHE_F16 | BITNET_HIP_ATTN_OUT_F16)", cache_f16);
    int rc = check_prefill_args(q, kv_gathered, rope_sin, rope_cos, kcache, vcache, workspace, out, n_heads, n_kv_heads, head_dim, max_pos, n_ctx);
    if (rc) return rc;
    if (n_q == 0 || !q_block_pos) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "attention_prefill_gathered: n_q and q_block_pos must be given");
    if (world == 0 || n_ctx % (2 * world * 64) != 0)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "attention_prefill_gathered: context %zu must be a multiple of 2 * world * 64 (world %zu)", n_ctx, world);
    if (ld_q < n_heads * head_dim) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "attention_prefill_gathered: row stride too small (ld_q %zu)", ld_q);
    const size_t need = attn_prefill_workspace_bytes((int)n_heads, (int)n_kv_heads, (int)n_q, (int)n_ctx);
    if (workspace_bytes < need)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "workspace too small: expected %zu, got %zu", need, workspace_bytes);
    BH_HIP_TRY(launch_attn_prefill(q, (int)ld_q, q_block_pos, (int)n_q, static_cast<const float *>(kv_gathered), (int)(2 * n_kv_heads * head_dim), (int)n_ctx,
                                   rope_sin, rope_cos, static_cast<float *>(kcache), static_cast<float *>(vcache), (int)n_heads, (int)n_kv_heads, (int)head_dim,
                                   (int)max_pos, workspace, workspace_bytes, out, (hipStream_t)stream, (int)world, kv_is_f16 ? 1 : 0, cache_f16 & 3, phase));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_pack_cols_dev(const float *src_dev, size_t ld, size_t col0, size_t ncols, size_t rows, void *dst_dev, int as_f16, void *stream) {
    BH_GUARD_BEGIN
    if (!src_dev || !dst_dev) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to pack_cols_dev");
    if (col0 + ncols > ld) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "pack_cols: columns [%zu, %zu) outside the row stride %zu", col0, col0 + ncols, ld);
    BH_HIP_TRY(launch_pack_cols(src_dev, ld, col0, ncols, rows, dst_dev, as_f16 ? 1 : 0, (hipStream_t)stream));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

/* K/rocm/attention.rs:54-65: q, k, v, output [batch, num_heads, seq_len, head_dim] row-major f32 (host). */
int bitnet_hip_attention(const float *q, size_t q_len, const float *k, size_t k_len, const float *v, size_t v_len, float *output,
                         size_t out_len, size_t seq_len, size_t num_heads, size_t head_dim, int causal, float scale) {
    BH_GUARD_BEGIN
    if (!q || !k || !v || !output) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to attention");
    BH_CHECK_DIMS(num_heads, head_dim, seq_len);
    if (num_heads == 0 || head_dim == 0 || seq_len == 0)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "dimensions must be > 0: num_heads=%zu, head_dim=%zu, seq_len=%zu", num_heads, head_dim, seq_len);
    if (head_dim != 128) return set_error(BITNET_HIP_ERR_UNSUPPORTED, "attention: head_dim %zu unsupported (128)", head_dim);
    const size_t per_batch = num_heads * seq_len * head_dim;
    if (q_len % per_batch != 0 || q_len == 0 || k_len != q_len || v_len != q_len || out_len != q_len)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "attention: q/k/v/output must all hold batch * %zu elements (got %zu, %zu, %zu, %zu)",
                         per_batch, q_len, k_len, v_len, out_len);
    int rc = ensure_init();
    if (rc) return rc;
    const size_t batch = q_len / per_batch, wsb = attn_prefill_workspace_bytes((int)num_heads, (int)num_heads, (int)seq_len, (int)seq_len);
    DevBuf qd, kd, vd, od, ws;
    if (qd.alloc(per_batch * 4) != hipSuccess || kd.alloc(per_batch * 4) != hipSuccess || vd.alloc(per_batch * 4) != hipSuccess ||
        od.alloc(per_batch * 4) != hipSuccess || ws.alloc(wsb) != hipSuccess)
        return set_error(BITNET_HIP_ERR_GPU, "hipMalloc failed in attention");
    for (size_t b = 0; b < batch; ++b) {
        if (hipMemcpy(qd.p, q + b * per_batch, per_batch * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(kd.p, k + b * per_batch, per_batch * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(vd.p, v + b * per_batch, per_batch * 4, hipMemcpyHostToDevice) != hipSuccess)
            return set_error(BITNET_HIP_ERR_GPU, "hipMemcpy H2D failed in attention");
        hipError_t e = launch_attn_generic(qd.as<float>(), kd.as<float>(), vd.as<float>(), od.as<float>(), (int)num_heads, (int)seq_len,
                                           causal != 0, scale, ws.p, wsb, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "kernel launch failed: %s", hipGetErrorString(e));
        if (hipMemcpy(output + b * per_batch, od.p, per_batch * 4, hipMemcpyDeviceToHost) != hipSuccess)
            return set_error(BITNET_HIP_ERR_GPU, "hipMemcpy D2H failed in attention");
    }
    return BITNET_HIP_OK;
    BH_GUARD_END
}

/* qk256_gemv_hip_batch (K/rocm/qk256_gemv.rs:73-82): the items one after the other. */
int bitnet_hip_qk256_gemv_batch(const bitnet_hip_gemv_item *items, size_t n_items) {
    if (!items && n_items) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to qk256_gemv_batch");
    for (size_t i = 0; i < n_items; ++i) {
        const bitnet_hip_gemv_item &it = items[i];
        const int rc = bitnet_hip_qk256_gemv(it.weights, it.weights_len, it.scales, it.scales_len, it.input, it.input_len, it.output,
                                             it.output_len, it.m, it.n, it.k);
        if (rc) return rc;
    }
    return BITNET_HIP_OK;
}

size_t bitnet_hip_attention_scratch_bytes(size_t n_kv_heads, size_t max_pos) {
    return attn_scratch_floats((int)n_kv_heads, (int)max_pos) * sizeof(float);
}

int bitnet_hip_logits_f16_dev(const void *table, const float *x, const float *gamma, float eps, size_t hidden, size_t vocab,
                              float *logits, void *scratch, size_t n_wg, int32_t *token_dev, int32_t *pos_dev,
                              int32_t *history_dev, const int32_t *n_forced_dev, void *stream) {
    BH_GUARD_BEGIN
    if (!table || !x || !logits || !scratch) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to logits_f16_dev");
    if (hidden % 512 != 0 || hidden > 8192) return set_error(BITNET_HIP_ERR_UNSUPPORTED, "logits: hidden %zu must be a multiple of 512, <= 8192", hidden);
    if (n_wg == 0 || n_wg > 65535) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "logits: bad workgroup count %zu", n_wg);
    float *bv = static_cast<float *>(scratch);
    int *bi = reinterpret_cast<int *>(bv + n_wg);
    BH_HIP_TRY(launch_logits_f16(table, x, gamma, eps, (int)hidden, (int)vocab, logits, bv, bi, (int)n_wg, (hipStream_t)stream));
    if (token_dev || pos_dev)
        BH_HIP_TRY(launch_argmax_final(bv, bi, (int)n_wg, token_dev, pos_dev, history_dev, n_forced_dev, (hipStream_t)stream));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_argmax_dev(const float *v, size_t n, void *scratch, size_t n_wg, int32_t *token_dev, void *stream) {
    BH_GUARD_BEGIN
    if (!v || !scratch || !token_dev) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to argmax_dev");
    if (n == 0 || n_wg == 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "argmax: empty input");
    float *bv = static_cast<float *>(scratch);
    int *bi = reinterpret_cast<int *>(bv + n_wg);
    BH_HIP_TRY(launch_argmax(v, (int)n, bv, bi, (int)n_wg, token_dev, (hipStream_t)stream));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

/* ------------------------------------------------- host-pointer drop-ins */

int bitnet_hip_gemv_qk256(const uint8_t *qs_data, size_t qs_len, const float *x, size_t x_len, float *y_out,
                          size_t y_len, size_t rows, size_t cols, size_t row_stride_bytes) {
    BH_GUARD_BEGIN
    if (!qs_data || !x || !y_out)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to gemv_qk256");
    BH_CHECK_DIMS(rows, cols);
    BH_CHECK_DIMS(rows, row_stride_bytes);
    // Q/i2s_qk256.rs:301-311, same order, same wording
    if (y_len != rows)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "I2S_QK256: y_out length %zu != rows %zu", y_len, rows);
    if (x_len < cols)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "I2S_QK256: x length %zu < cols %zu", x_len, cols);
    const size_t expected_total = rows * row_stride_bytes;
    if (qs_len < expected_total)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "I2S_QK256: data too short: %zu < %zu", qs_len, expected_total);
    if (rows == 0) return BITNET_HIP_OK;  // nothing to write; the reference's loop is empty too
    if (cols == 0) {
        memset(y_out, 0, rows * sizeof(float));  // empty dot product
        return BITNET_HIP_OK;
    }
    if (row_stride_bytes != div_ceil(cols, 256) * 64)  // debug_assert :200-207
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT,
                         "I2S_QK256: row bytes mismatch: got %zu, expected %zu for %zu cols", row_stride_bytes,
                         div_ceil(cols, 256) * 64, cols);
    bitnet_hip_weights_t h = 0;
    int rc = bitnet_hip_weights_upload_qk256(qs_data, expected_total, rows, cols, row_stride_bytes, &h);
    if (rc) return rc;
    DevBuf xd, yd;
    rc = BITNET_HIP_OK;
    if (xd.alloc(cols * 4) != hipSuccess || yd.alloc(rows * 4) != hipSuccess ||
        hipMemcpy(xd.p, x, cols * 4, hipMemcpyHostToDevice) != hipSuccess)
        rc = set_error(BITNET_HIP_ERR_GPU, "device staging failed in gemv_qk256");
    if (!rc) rc = bitnet_hip_gemv_dev(h, xd.as<float>(), yd.as<float>(), nullptr);
    if (!rc && hipMemcpy(y_out, yd.p, rows * 4, hipMemcpyDeviceToHost) != hipSuccess)
        rc = set_error(BITNET_HIP_ERR_GPU, "hipMemcpy D2H failed in gemv_qk256");
    std::string keep = g_last_error;
    bitnet_hip_weights_free(h);
    if (rc) g_last_error = keep;
    return rc;
    BH_GUARD_END
}

int bitnet_hip_i2s_matmul_f32(const float *act, size_t act_len, const uint8_t *wp, size_t w_len,
                              const float *scales, size_t scales_len, float *out, size_t out_len, size_t m,
                              size_t n, size_t k, size_t block_size) {
    BH_GUARD_BEGIN
    if (!act || !wp || !scales || !out)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to i2s_matmul_f32");
    // K/cpu/quantized_matmul.rs:204-256
    if (block_size == 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "block_size must be > 0");
    BH_CHECK_DIMS(m, n, k);
    if (m == 0 || n == 0 || k == 0)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "dimensions must be > 0: m=%zu, n=%zu, k=%zu", m, n, k);
    const size_t packed_k = div_ceil(k, 4), nblk = div_ceil(k, block_size);
    if (act_len < m * k)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "activations too small: expected %zu, got %zu", m * k, act_len);
    if (w_len < packed_k * n)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "weights_packed too small: expected %zu, got %zu", packed_k * n, w_len);
    if (scales_len < n * nblk)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "scales too small: expected %zu, got %zu", n * nblk, scales_len);
    if (out_len < m * n)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "output too small: expected %zu, got %zu", m * n, out_len);
    bitnet_hip_weights_t h = 0;
    int rc = bitnet_hip_weights_upload_i2s(wp, w_len, scales, scales_len, n, k, block_size, &h);
    if (rc) return rc;
    DevBuf xd, yd;
    if (xd.alloc(m * k * 4) != hipSuccess || yd.alloc(m * n * 4) != hipSuccess ||
        hipMemcpy(xd.p, act, m * k * 4, hipMemcpyHostToDevice) != hipSuccess)
        rc = set_error(BITNET_HIP_ERR_GPU, "device staging failed in i2s_matmul_f32");
    if (!rc) rc = bitnet_hip_matmul_dev(h, xd.as<float>(), yd.as<float>(), m, nullptr);
    if (!rc) {
        memset(out, 0, out_len * sizeof(float));  // out.fill(0.0) :72
        if (hipMemcpy(out, yd.p, m * n * 4, hipMemcpyDeviceToHost) != hipSuccess)
            rc = set_error(BITNET_HIP_ERR_GPU, "hipMemcpy D2H failed in i2s_matmul_f32");
    }
    std::string keep = g_last_error;
    bitnet_hip_weights_free(h);
    if (rc) g_last_error = keep;
    return rc;
    BH_GUARD_END
}

int bitnet_hip_qk256_gemv(const uint8_t *weights, size_t w_len, const float *scales, size_t scales_len,
                          const float *input, size_t in_len, float *output, size_t out_len, size_t m, size_t n,
                          size_t k) {
    if (k % 256 != 0 || k == 0) {  // K/cuda/qk256_gemv.rs:60-68
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT,
                         "QK256 GEMV inner dimension k=%zu must be a positive multiple of 256", k);
    }
    return bitnet_hip_i2s_matmul_f32(input, in_len, weights, w_len, scales, scales_len, output, out_len, m, n, k, 256);
}

int bitnet_hip_matmul_i2s(const int8_t *a, size_t a_len, const uint8_t *b, size_t b_len, float *c, size_t c_len,
                          size_t m, size_t n, size_t k) {
    BH_GUARD_BEGIN
    if (!a || !b || !c) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to matmul_i2s");
    BH_CHECK_DIMS(m, n, k);
    // K/cpu/fallback.rs:49-63
    if (a_len != m * k)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Matrix A dimension mismatch: expected %zu, got %zu", m * k, a_len);
    if (b_len != k * n)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Matrix B dimension mismatch: expected %zu, got %zu", k * n, b_len);
    if (c_len != m * n)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Matrix C dimension mismatch: expected %zu, got %zu", m * n, c_len);
    if (m * n == 0) return BITNET_HIP_OK;
    int rc = ensure_init();
    if (rc) return rc;
    DevBuf ad, bd, cd;
    if (ad.alloc(a_len) != hipSuccess || bd.alloc(b_len) != hipSuccess || cd.alloc(c_len * 4) != hipSuccess)
        return set_error(BITNET_HIP_ERR_GPU, "hipMalloc failed in matmul_i2s");
    BH_HIP_TRY(hipMemcpy(ad.p, a, a_len, hipMemcpyHostToDevice));
    BH_HIP_TRY(hipMemcpy(bd.p, b, b_len, hipMemcpyHostToDevice));
    // integer tiles (v_dot4_u32_u8) unless the caller pinned the reference-order kernel or the shape is odd
    if (g_kernel.load(std::memory_order_relaxed) != BITNET_HIP_KERNEL_EXACT && matmul_i2s_tiled_ok(m, n, k))
        BH_HIP_TRY(launch_matmul_i2s_tiled(ad.as<int8_t>(), bd.as<uint8_t>(), cd.as<float>(), m, n, k, nullptr));
    else
        BH_HIP_TRY(launch_matmul_i2s_u8(ad.as<int8_t>(), bd.as<uint8_t>(), cd.as<float>(), m, n, k, nullptr));
    BH_HIP_TRY(hipMemcpy(c, cd.p, c_len * 4, hipMemcpyDeviceToHost));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

/* QuantizedLinear::quantized_matmul_i2s (crates/bitnet-inference/src/layers/quantized_linear.rs:704-802), all on the device:
 * input -> i8 by clamp(x, -2, 1).round() (:1762-1773); packed 2-bit weights -> raw codes 0..3 as the [k, n] u8 operand
 * (:769-776); matmul_i2s; per-output scale (:779-802). */
int bitnet_hip_quantized_matmul_i2s(const float *input, size_t in_len, const uint8_t *weights_packed, size_t w_len, const float *scales,
                                    size_t scales_len, size_t block_size, float *output, size_t out_len, size_t m, size_t n, size_t k) {
    BH_GUARD_BEGIN
    if (!input || !weights_packed || !output || (!scales && scales_len))
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to quantized_matmul_i2s");
    BH_CHECK_DIMS(m, n, k);
    if (block_size == 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "block_size must be > 0");
    if (in_len != m * k) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Matrix A dimension mismatch: expected %zu, got %zu", m * k, in_len);
    if (w_len < div_ceil(k * n, 4))  // unpack_2bit_values would come up short and matmul_i2s refuse b (K/cpu/fallback.rs:54-58)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Matrix B dimension mismatch: expected %zu, got %zu", k * n, w_len * 4);
    if (out_len != m * n) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Matrix C dimension mismatch: expected %zu, got %zu", m * n, out_len);
    if (m * n == 0) return BITNET_HIP_OK;
    int rc = ensure_init();
    if (rc) return rc;
    DevBuf xd, qd, wd, bd, cd, sd;
    if (xd.alloc(in_len * 4) != hipSuccess || qd.alloc(in_len + 16) != hipSuccess || wd.alloc(w_len) != hipSuccess || bd.alloc(k * n + 16) != hipSuccess ||
        cd.alloc(out_len * 4) != hipSuccess || sd.alloc(scales_len * 4) != hipSuccess)
        return set_error(BITNET_HIP_ERR_GPU, "hipMalloc failed in quantized_matmul_i2s");
    BH_HIP_TRY(hipMemcpy(xd.p, input, in_len * 4, hipMemcpyHostToDevice));
    BH_HIP_TRY(hipMemcpy(wd.p, weights_packed, w_len, hipMemcpyHostToDevice));
    if (scales_len) BH_HIP_TRY(hipMemcpy(sd.p, scales, scales_len * 4, hipMemcpyHostToDevice));
    BH_HIP_TRY(launch_quant_input_i2s(xd.as<float>(), qd.as<int8_t>(), in_len, nullptr));
    BH_HIP_TRY(launch_unpack_codes_u8(wd.as<uint8_t>(), bd.as<uint8_t>(), k * n, nullptr));
    if (g_kernel.load(std::memory_order_relaxed) != BITNET_HIP_KERNEL_EXACT && matmul_i2s_tiled_ok(m, n, k))
        BH_HIP_TRY(launch_matmul_i2s_tiled(qd.as<int8_t>(), bd.as<uint8_t>(), cd.as<float>(), m, n, k, nullptr));
    else
        BH_HIP_TRY(launch_matmul_i2s_u8(qd.as<int8_t>(), bd.as<uint8_t>(), cd.as<float>(), m, n, k, nullptr));
    BH_HIP_TRY(launch_apply_scales(cd.as<float>(), m, n, sd.as<float>(), scales_len, k, block_size, nullptr));
    BH_HIP_TRY(hipMemcpy(output, cd.p, out_len * 4, hipMemcpyDeviceToHost));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_quantize(const float *input, size_t input_len, uint8_t *output, size_t output_len, float *scales,
                        size_t scales_len, int qtype) {
    BH_GUARD_BEGIN
    if (!input || !output || !scales) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to quantize");
    if (qtype < 0 || qtype > 2) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Invalid quantization type");
    if (qtype != BITNET_HIP_QTYPE_I2S)
        return set_error(BITNET_HIP_ERR_UNSUPPORTED, "quantize: only I2_S is implemented on the ROCm hot path (qtype=%d)", qtype);
    // K/cpu/fallback.rs:104-124
    const size_t num_blocks = div_ceil(input_len, 32);
    if (output_len < input_len / 4)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Output buffer too small for I2_S: expected %zu, got %zu", input_len / 4, output_len);
    if (scales_len < num_blocks)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Scales buffer too small: expected %zu, got %zu", num_blocks, scales_len);
    if (input_len == 0) return BITNET_HIP_OK;
    int rc = ensure_init();
    if (rc) return rc;
    DevBuf id, od, sd;
    if (id.alloc(input_len * 4) != hipSuccess || od.alloc(output_len) != hipSuccess || sd.alloc(num_blocks * 4) != hipSuccess)
        return set_error(BITNET_HIP_ERR_GPU, "hipMalloc failed in quantize");
    BH_HIP_TRY(hipMemcpy(id.p, input, input_len * 4, hipMemcpyHostToDevice));
    BH_HIP_TRY(hipMemcpy(od.p, output, output_len, hipMemcpyHostToDevice));  // OR-pack semantics
    if (input_len % 32 == 0 && g_kernel.load(std::memory_order_relaxed) != BITNET_HIP_KERNEL_EXACT)
        BH_HIP_TRY(launch_quantize_i2s_fast(id.as<float>(), input_len, od.as<uint8_t>(), output_len, sd.as<float>(), nullptr));  // same values, coalesced
    else
        BH_HIP_TRY(launch_quantize_i2s(id.as<float>(), input_len, od.as<uint8_t>(), output_len, sd.as<float>(), nullptr));
    BH_HIP_TRY(hipMemcpy(output, od.p, output_len, hipMemcpyDeviceToHost));
    BH_HIP_TRY(hipMemcpy(scales, sd.p, num_blocks * 4, hipMemcpyDeviceToHost));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_dequant_i2s(const uint8_t *bytes, size_t bytes_len, size_t rows, size_t cols, int inv_scale, float k,
                           int transposed, float *out, size_t out_len) {
    BH_GUARD_BEGIN
    if (!bytes || !out) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to dequant_i2s");
    if (rows == 0 || cols == 0) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "I2_S: empty tensor dims");
    BH_CHECK_DIMS(rows, cols);
    if (out_len < rows * cols)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "I2_S: output bounds exceeded: %zu > %zu", rows * cols, out_len);
    // M/quant/i2s.rs:205-214 (non-transposed tries 256 first, which the candidate
    // order {256,128,64,32} already does)
    size_t block = 0;
    for (size_t b : {size_t(256), size_t(128), size_t(64), size_t(32)}) {
        if (rows * div_ceil(cols, b) * (b / 4 + 2) == bytes_len) {
            block = b;
            break;
        }
    }
    if (!block)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT,
                         "I2_S: byte length mismatch (got %zu for %zux%zu): no block size in {256,128,64,32} fits",
                         bytes_len, rows, cols);
    int rc = ensure_init();
    if (rc) return rc;
    DevBuf bd, od;
    if (bd.alloc(bytes_len) != hipSuccess || od.alloc(rows * cols * 4) != hipSuccess)
        return set_error(BITNET_HIP_ERR_GPU, "hipMalloc failed in dequant_i2s");
    BH_HIP_TRY(hipMemcpy(bd.p, bytes, bytes_len, hipMemcpyHostToDevice));
    BH_HIP_TRY(launch_dequant_i2s(bd.as<uint8_t>(), rows, cols, block, inv_scale, k, transposed, od.as<float>(), nullptr));
    BH_HIP_TRY(hipMemcpy(out, od.p, rows * cols * 4, hipMemcpyDeviceToHost));
    return BITNET_HIP_OK;
    BH_GUARD_END
}

int bitnet_hip_hbm_read_ceiling(size_t bytes, int iters, double *best_gbs, double *mean_gbs, void *stream) {
    BH_GUARD_BEGIN
    if (!best_gbs || !mean_gbs) return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "Null pointer passed to hbm_read_ceiling");
    if (bytes < (size_t(1) << 20) || bytes > (size_t(64) << 30) || iters < 1 || iters > 1000)
        return set_error(BITNET_HIP_ERR_INVALID_ARGUMENT, "hbm_read_ceiling: bytes in [1 MiB, 64 GiB], iters in [1, 1000]");
    int rc = ensure_init();
    if (rc) return rc;
    bytes &= ~size_t(15);
    DevBuf buf, sink;
    if (buf.alloc(bytes) != hipSuccess || sink.alloc(64) != hipSuccess)
        return set_error(BITNET_HIP_ERR_GPU, "hipMalloc(%zu) failed in hbm_read_ceiling", bytes);
    hipStream_t s = (hipStream_t)stream;
    BH_HIP_TRY(hipMemsetAsync(buf.p, 0x5a, bytes, s));
    hipEvent_t e0, e1;
    BH_HIP_TRY(hipEventCreate(&e0));
    BH_HIP_TRY(hipEventCreate(&e1));
    double best = 0.0, sum = 0.0;
    hipError_t e = launch_stream_read(buf.p, bytes, sink.as<unsigned>(), s);  // warm-up (code load)
    for (int i = 0; i < iters && e == hipSuccess; ++i) {
        e = hipEventRecord(e0, s);
        if (e == hipSuccess) e = launch_stream_read(buf.p, bytes, sink.as<unsigned>(), s);
        if (e == hipSuccess) e = hipEventRecord(e1, s);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && ms > 0.f) {
            const double g = (double)bytes / (ms * 1e-3) / 1e9;
            best = g > best ? g : best;
            sum += g;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess) return set_error(BITNET_HIP_ERR_GPU, "hbm_read_ceiling failed: %s", hipGetErrorString(e));
    *best_gbs = best;
    *mean_gbs = sum / iters;
    return BITNET_HIP_OK;
    BH_GUARD_END
}

#ifdef BH_STAMPS
/* Diagnostic build only (not declared in include/bitnet_hip.h, absent from the
 * production library): device buffer of 8 x u64 per workgroup for in-kernel stamps. */
int bitnet_hip_debug_set_stamps(void *dev_buffer) {
    bitnet_hip::g_mfma_stamps = static_cast<unsigned long long *>(dev_buffer);
    return BITNET_HIP_OK;
}
#endif

}  // extern "C"
