// ccsp_host_util.h -- small host helpers: debug scatter kernels, nblk, dispatch_h (hidden_dim -> compile-time width), StreamBuf.
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.

// NaN rows for the edge-output debug API, then scatter sorted -> original order
__global__ void k_fill(float* p, long n, float v) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_unsort_edges(int E_act, int P, const int* __restrict__ e_orig, const int* __restrict__ ent_pos,
                               const float* __restrict__ O, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)E_act * 2 * P) return;
    const int k = (int)(idx / (2 * P)), j = (int)(idx % (2 * P));
    const int sl = j / P, p = j % P;
    out[(size_t)e_orig[k] * 2 * P + j] = O[(size_t)ent_pos[2 * k + sl] * P + p];
}

inline int nblk(long n, int b) { return (int)((n + b - 1) / b); }

// stream-ordered scratch of the operator entry points, released on every path out of the scope
// hidden_dim -> the kernels' compile-time H: the widths built are 64, 128 and 256 (ccsp_model_create rejects the rest)
template <typename F>
auto dispatch_h(int H, F&& f) {
    if (H == 256) return f(std::integral_constant<int, 256>{});
    if (H == 128) return f(std::integral_constant<int, 128>{});
    if (H == 64) return f(std::integral_constant<int, 64>{});
    // every other multiple of 64 up to 512 (train_utils.py:107 takes any -hidden_dim): the same templates through their generic tile
    // configurations (EdgeCfg / EdgeBfCfg / BwdCfg primaries); the f16x2 kernels and their residency tuning are hidden_dim 256's
    if (H == 192) return f(std::integral_constant<int, 192>{});
    if (H == 320) return f(std::integral_constant<int, 320>{});
    if (H == 384) return f(std::integral_constant<int, 384>{});
    if (H == 448) return f(std::integral_constant<int, 448>{});
    return f(std::integral_constant<int, 512>{});
}

struct StreamBuf {
    void* p = nullptr;
    hipStream_t s;
    explicit StreamBuf(hipStream_t st) : s(st) {}
    ~StreamBuf() { if (p) (void)hipFreeAsync(p, s); }
    StreamBuf(const StreamBuf&) = delete;
    StreamBuf& operator=(const StreamBuf&) = delete;
    int alloc(size_t bytes) { HIP_TRY(hipMallocAsync(&p, bytes, s)); return 0; }
    float* f() const { return (float*)p; }
};

