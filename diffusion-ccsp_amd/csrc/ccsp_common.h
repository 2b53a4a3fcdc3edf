// ccsp_common.h -- error reporting (fail / HIP_TRY), tile constants, the activations.
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.
thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

constexpr int TILE_M = 64;    // U-row tile of k_ugemm (rows never straddle a (type,slot) group)
constexpr int TILE_N = 128;   // U-column tile
constexpr int BK = 32;        // K chunk staged through LDS
constexpr int LDS_LD = BK + 1;  // padded row stride: fragment reads and staging writes are conflict free
constexpr int NODE_TILE = 16; // nodes per workgroup in the node kernels

typedef float floatx16 __attribute__((ext_vector_type(16)));

// SiLU.  silu_f: IEEE division + libm-grade expf (set-up kernels).  silu_fast: v_exp_f32 + v_rcp_f32
// (~1 ulp each, relative error of the result ~3e-7), 6 VALU instructions instead of ~35 -- the
// activation sits on the operand path of the MFMA kernels, where VALU issue competes with the
// matrix pipe.  Limits: v -> -inf gives -0 (exp2 -> inf, rcp -> 0), v -> +inf gives v, NaN stays NaN.
__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float silu_fast(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}
__device__ __forceinline__ float mish_f(float v) {
    const float sp = v > 20.0f ? v : log1pf(expf(v));
    return v * tanhf(sp);
}

