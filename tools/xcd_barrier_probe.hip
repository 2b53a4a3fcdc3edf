// what does a barrier cost when every workgroup of a persistent launch sits behind ONE XCD's L2?
// tools/barrier_probe.hip (round 2) paid for agent-scope fences across eight non-coherent L2s: 4.7 us at 48 workgroups, 25 us at 256.
// Here: a stream created with a CU mask that keeps the launch on one XCD (the mask -> XCD mapping is found by a census of
// HW_REG_XCC_ID, not assumed), 32 workgroups (one per CU), the same three-phase producer / consumer "evaluation" with every word
// checked, and a barrier made of what one coherent L2 needs:
//     producer: plain stores -> s_waitcnt vmcnt(0) (the L1 is write-through: the data is in the XCD's L2) -> __syncthreads ->
//               one relaxed WORKGROUP-scope fetch_add (executes in the L2) -> poll the generation word with sc1 loads (served by the L2)
//     consumer: reads the partner's slot with sc1 loads (bypass the CU's L1; no buffer_inv, no buffer_wbl2)
// next to (a) three dependent launches on the same masked stream and (b) the same persistent kernel with agent-scope fences.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/xcd_barrier_probe tools/xcd_barrier_probe.hip      run: tools/xcd_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int THREADS = 256;

__global__ void k_census(unsigned* xcc) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) xcc[blockIdx.x] = id & 0xf;
    // keep the workgroup alive long enough for the whole grid to be resident (one per CU)
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) __builtin_amdgcn_s_sleep(8);
}

struct Bar { unsigned count, gen, error, pad[29]; };

__device__ __forceinline__ float4 ld_sc1(const float4* p) {
    float4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned ld_u32_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// MODE 0: XCD-local barrier (no fences); MODE 1: agent-scope fences (the form of tools/barrier_probe.hip)
template <int MODE>
__device__ __forceinline__ bool barrier(Bar* b, unsigned G, unsigned& my_gen) {
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have reached the L2
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (MODE == 1) __threadfence();
        const unsigned target = my_gen + 1;
        unsigned old;
        if (MODE == 0) old = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else old = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == G - 1) {
            if (MODE == 0) {
                __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&b->gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            unsigned spins = 0;
            for (;;) {
                const unsigned g = MODE == 0 ? ld_u32_sc1(&b->gen) : __hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (g == target) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { ok = false; atomicAdd(&b->error, 1u); break; }     // bounded: never hang the GPU
            }
        }
        if (MODE == 1) __threadfence();
    }
    my_gen++;
    __syncthreads();
    return ok;
}

// filter > 1: the grid has filter x G workgroups and only those with blockIdx % filter == 0 take part (block b runs on XCD b % 8 with
// the dispatcher's round-robin placement -- observed, not guaranteed: every participant checks its XCC id against the first one's)
template <int MODE>
__global__ void k_persistent(int G, int iters, float* buf0, float* buf1, int payload_f4, unsigned* bad, Bar* bar, int filter, unsigned* xcc_seen) {
    if (blockIdx.x % filter != 0) return;
    const int w = blockIdx.x / filter;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        atomicOr(xcc_seen, 1u << (id & 0xf));
    }
    unsigned my_gen = 0;
    const int partner = (w + G / 2 + 1) % G;
    for (int iter = 0; iter < iters; ++iter)
        for (int phase = 0; phase < 3; ++phase) {
            const int step = iter * 3 + phase;
            float* rd = (step & 1) ? buf0 : buf1;
            float* wr = (step & 1) ? buf1 : buf0;
            if (step > 0) {
                const float4* theirs = reinterpret_cast<const float4*>(rd) + (size_t)partner * payload_f4;
                for (int i = threadIdx.x; i < payload_f4; i += THREADS) {
                    const float4 v = MODE == 0 ? ld_sc1(theirs + i) : theirs[i];
                    if (v.x != (float)step || v.w != (float)step) atomicAdd(bad, 1u);
                }
            }
            float4* mine = reinterpret_cast<float4*>(wr) + (size_t)w * payload_f4;
            const float t = (float)(step + 1);
            for (int i = threadIdx.x; i < payload_f4; i += THREADS) mine[i] = make_float4(t, t, t, t);
            if (!barrier<MODE>(bar, (unsigned)G, my_gen)) return;
        }
}

__global__ void k_phase(int G, int phase, int iter, float* buf0, float* buf1, int payload_f4, unsigned* bad) {
    const int w = blockIdx.x;
    const int step = iter * 3 + phase;
    float* rd = (step & 1) ? buf0 : buf1;
    float* wr = (step & 1) ? buf1 : buf0;
    const int partner = (w + G / 2 + 1) % G;
    if (step > 0) {
        const float4* theirs = reinterpret_cast<const float4*>(rd) + (size_t)partner * payload_f4;
        for (int i = threadIdx.x; i < payload_f4; i += THREADS)
            if (theirs[i].x != (float)step) atomicAdd(bad, 1u);
    }
    float4* mine = reinterpret_cast<float4*>(wr) + (size_t)w * payload_f4;
    const float t = (float)(step + 1);
    for (int i = threadIdx.x; i < payload_f4; i += THREADS) mine[i] = make_float4(t, t, t, t);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device: %s, %d CUs\n", prop.name, ncu);
    unsigned* d_xcc;
    CHECK(hipMalloc(&d_xcc, 64 * sizeof(unsigned)));
    // which CU mask keeps a launch on one XCD?  candidates: every 8th bit; one contiguous run of ncu / 8 bits
    hipStream_t best = nullptr;
    // candidates: bits i with (i / s) % 8 == 0 for s = 1 .. 32 (s = 32: one contiguous run of ncu / 8 bits)
    for (int sft = 0; sft <= 5 && !best; ++sft) {
        const int st = 1 << sft;
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        int nbits = 0;
        for (int i = 0; i < ncu; ++i)
            if ((i / st) % 8 == 0) { mask[i >> 5] |= 1u << (i & 31); ++nbits; }
        hipStream_t s;
        CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
        CHECK(hipMemsetAsync(d_xcc, 0xff, 64 * sizeof(unsigned), s));
        hipLaunchKernelGGL(k_census, dim3(32), dim3(THREADS), 0, s, d_xcc);
        CHECK(hipStreamSynchronize(s));
        unsigned h[32];
        CHECK(hipMemcpy(h, d_xcc, sizeof(h), hipMemcpyDeviceToHost));
        unsigned seen = 0;
        for (int i = 0; i < 32; ++i) seen |= 1u << h[i];
        printf("CU mask {i : (i / %2d) %% 8 == 0} (%d bits): 32 workgroups ran on XCC ids {", st, nbits);
        for (int x = 0; x < 16; ++x) if (seen & (1u << x)) printf(" %d", x);
        printf(" }\n");
        if (__builtin_popcount(seen) == 1) best = s; else CHECK(hipStreamDestroy(s));
    }
    int filter = 1;
    if (!best) {
        printf("no CU mask keeps a launch on one XCD (the mask is applied inside every XCD); falling back to the dispatcher's round-robin placement:\n"
               "a grid of 8 x 32 workgroups on an ordinary stream in which only blocks with blockIdx %% 8 == 0 take part\n");
        CHECK(hipStreamCreateWithFlags(&best, hipStreamNonBlocking));
        filter = 8;
    }
    const int G = 32, iters = 2000;
    unsigned* d_seen;
    CHECK(hipMalloc(&d_seen, 4));
    for (int payload_f4 : {16, 256}) {                             // 256 B and 4 KB per workgroup and phase
        float *b0, *b1;
        unsigned* bad;
        Bar* bar;
        CHECK(hipMalloc(&b0, (size_t)G * payload_f4 * 16)); CHECK(hipMalloc(&b1, (size_t)G * payload_f4 * 16));
        CHECK(hipMalloc(&bad, 4)); CHECK(hipMalloc(&bar, sizeof(Bar)));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        auto reset = [&]() { CHECK(hipMemset(bad, 0, 4)); CHECK(hipMemset(d_seen, 0, 4)); CHECK(hipMemset(bar, 0, sizeof(Bar))); CHECK(hipMemset(b0, 0, (size_t)G * payload_f4 * 16)); CHECK(hipMemset(b1, 0, (size_t)G * payload_f4 * 16)); };
        float ms;
        unsigned h_bad, h_seen;
        Bar h_bar;
        // (a) three dependent launches per evaluation
        reset();
        CHECK(hipEventRecord(e0, best));
        for (int it = 0; it < iters; ++it)
            for (int ph = 0; ph < 3; ++ph) hipLaunchKernelGGL(k_phase, dim3(G), dim3(THREADS), 0, best, G, ph, it, b0, b1, payload_f4, bad);
        CHECK(hipEventRecord(e1, best)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(&h_bad, bad, 4, hipMemcpyDeviceToHost));
        printf("payload %5d B: dependent launches      %7.2f us per phase   (stale words: %u)\n", payload_f4 * 16, 1e3 * ms / (3.0 * iters), h_bad);
        for (int mode = 0; mode < 2; ++mode) {
            reset();
            CHECK(hipEventRecord(e0, best));
            if (mode == 0) hipLaunchKernelGGL(k_persistent<0>, dim3(G * filter), dim3(THREADS), 0, best, G, iters, b0, b1, payload_f4, bad, bar, filter, d_seen);
            else hipLaunchKernelGGL(k_persistent<1>, dim3(G * filter), dim3(THREADS), 0, best, G, iters, b0, b1, payload_f4, bad, bar, filter, d_seen);
            CHECK(hipEventRecord(e1, best)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(&h_bad, bad, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&h_bar, bar, sizeof(Bar), hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&h_seen, d_seen, 4, hipMemcpyDeviceToHost));
            printf("payload %5d B: %-22s %7.2f us per phase   (stale words: %u, barrier time-outs: %u, XCC ids of the participants: 0x%x%s)\n", payload_f4 * 16,
                   mode == 0 ? "XCD-local barrier" : "agent-scope barrier", 1e3 * ms / (3.0 * iters), h_bad, h_bar.error, h_seen,
                   __builtin_popcount(h_seen) == 1 ? " = one XCD" : " = SEVERAL XCDs: the XCD-local form is not valid here");
        }
        CHECK(hipFree(b0)); CHECK(hipFree(b1)); CHECK(hipFree(bad)); CHECK(hipFree(bar));
    }
    return 0;
}
