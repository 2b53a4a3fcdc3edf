// What does a grid-wide barrier cost on this chip next to a kernel boundary?  ("one persistent launch per
// timestep, three grid barriers per evaluation".)  Both forms run the same three-phase "evaluation" N times:
//   phase p of workgroup w stores PAYLOAD bytes to its slot, the next phase loads the slot of workgroup (w + G/2 + 1) % G --
//   another XCD with the default round-robin placement -- and checks it, so the synchronisation has to make data visible.
//   (a) three dependent kernel launches per evaluation on one stream;
//   (b) ONE persistent kernel, G co-resident workgroups, a sense-reversing counter barrier in device memory (agent-scope
//       atomics, s_sleep back-off, bounded spin: a lost wake-up ends the kernel with an error flag instead of hanging the GPU).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/barrier_probe tools/barrier_probe.hip      run: tools/barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int THREADS = 256;

struct Bar {
    unsigned count;
    unsigned gen;
    unsigned error;
    unsigned pad[29];
    unsigned xcd_count[8][32];                                    // one counter per XCD, 128 bytes apart
};

// (a phase rewrites its own slot while the partner of the previous phase may still be reading it, so the slots are double-buffered
// by phase parity in both forms)
__global__ void k_phase(int G, int phase, int iter, float* buf0, float* buf1, int payload_f4, unsigned* bad) {
    const int w = blockIdx.x;
    const int step = iter * 3 + phase;
    float* rd = (step & 1) ? buf0 : buf1;                         // written by step - 1
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

// TREE = false: every workgroup increments ONE counter.  TREE = true: workgroups of an XCD (blockIdx % 8 with the default
// round-robin placement) meet on their own counter first, the last one of each XCD goes on to the global one.
template <bool TREE>
__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned G, unsigned& my_gen) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();                                          // release this workgroup's stores at agent scope
        bool last;
        if (TREE) {
            const unsigned x = blockIdx.x & 7u, nx = (G + 7u - x) / 8u;                // workgroups on this XCD
            last = false;
            if (__hip_atomic_fetch_add(&b->xcd_count[x][0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nx - 1) {
                __hip_atomic_store(&b->xcd_count[x][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned nxcd = G < 8u ? G : 8u;
                last = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nxcd - 1;
            }
        } else {
            last = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == G - 1;
        }
        if (last) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == my_gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000u) { __hip_atomic_store(&b->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
            }
        }
        __threadfence();                                          // acquire
    }
    my_gen++;
    __syncthreads();
    return ok;
}

template <bool TREE>
__global__ void k_persistent(int G, int iters, float* buf0, float* buf1, int payload_f4, unsigned* bad, Bar* bar) {
    const int w = blockIdx.x;
    unsigned my_gen = 0;
    const int partner = (w + G / 2 + 1) % G;
    for (int step = 0; step < iters * 3; ++step) {
        float* rd = (step & 1) ? buf0 : buf1;
        float* wr = (step & 1) ? buf1 : buf0;
        if (step > 0) {
            const float4* theirs = reinterpret_cast<const float4*>(rd) + (size_t)partner * payload_f4;
            for (int i = threadIdx.x; i < payload_f4; i += THREADS) {
                // (plain loads: the barrier's acquire fence is what has to make the partner's stores visible)
                if (theirs[i].x != (float)step) atomicAdd(bad, 1u);
            }
        }
        float4* mine = reinterpret_cast<float4*>(wr) + (size_t)w * payload_f4;
        const float t = (float)(step + 1);
        for (int i = threadIdx.x; i < payload_f4; i += THREADS) mine[i] = make_float4(t, t, t, t);
        if (!grid_barrier<TREE>(bar, (unsigned)G, my_gen)) return;
        if (__hip_atomic_load(&bar->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; %d evaluations of three phases each\n", prop.gcnArchName, prop.multiProcessorCount, iters);
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int Gs[] = {48, 128, 256, 512, 640};
    const int payloads[] = {0, 1024, 65536};                      // bytes stored (and then loaded by the partner) per workgroup and phase
    printf("%6s %9s | %22s | %s\n", "WGs", "payload B", "3 launches: us / phase", "persistent, us / phase: one counter | per-XCD counters");
    for (int payload : payloads)
        for (int G : Gs) {
            const int payload_f4 = payload / 16;
            float *buf0, *buf1;
            unsigned* bad;
            Bar* bar;
            const size_t bytes = (size_t)G * (payload > 0 ? payload : 16);
            CHECK(hipMalloc(&buf0, bytes)); CHECK(hipMalloc(&buf1, bytes));
            CHECK(hipMalloc(&bad, sizeof(unsigned))); CHECK(hipMalloc(&bar, sizeof(Bar)));
            CHECK(hipMemset(buf0, 0, bytes)); CHECK(hipMemset(buf1, 0, bytes));
            CHECK(hipMemset(bad, 0, sizeof(unsigned))); CHECK(hipMemset(bar, 0, sizeof(Bar)));
            // (a) launches
            for (int warm = 0; warm < 2; ++warm) {
                CHECK(hipMemsetAsync(bad, 0, sizeof(unsigned), s));
                CHECK(hipEventRecord(e0, s));
                for (int it = 0; it < iters; ++it)
                    for (int p = 0; p < 3; ++p) hipLaunchKernelGGL(k_phase, dim3(G), dim3(THREADS), 0, s, G, p, it, buf0, buf1, payload_f4, bad);
                CHECK(hipEventRecord(e1, s));
                CHECK(hipStreamSynchronize(s));
            }
            float ms_a = 0;
            CHECK(hipEventElapsedTime(&ms_a, e0, e1));
            unsigned h_bad_a = 0;
            CHECK(hipMemcpy(&h_bad_a, bad, sizeof(unsigned), hipMemcpyDeviceToHost));
            // (b) persistent (co-residency: G <= CUs x resident workgroups; 256 threads, no LDS -> 8 per CU)
            float ms_b[2] = {0, 0};
            unsigned h_bad_b[2] = {0, 0};
            bool spin[2] = {false, false};
            for (int tree = 0; tree < 2; ++tree) {
                Bar h_bar{};
                for (int warm = 0; warm < 2; ++warm) {
                    CHECK(hipMemsetAsync(bad, 0, sizeof(unsigned), s));
                    CHECK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
                    CHECK(hipEventRecord(e0, s));
                    if (tree) hipLaunchKernelGGL(k_persistent<true>, dim3(G), dim3(THREADS), 0, s, G, iters, buf0, buf1, payload_f4, bad, bar);
                    else hipLaunchKernelGGL(k_persistent<false>, dim3(G), dim3(THREADS), 0, s, G, iters, buf0, buf1, payload_f4, bad, bar);
                    CHECK(hipEventRecord(e1, s));
                    CHECK(hipStreamSynchronize(s));
                }
                CHECK(hipEventElapsedTime(&ms_b[tree], e0, e1));
                CHECK(hipMemcpy(&h_bad_b[tree], bad, sizeof(unsigned), hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(&h_bar, bar, sizeof(Bar), hipMemcpyDeviceToHost));
                spin[tree] = h_bar.error != 0;
            }
            printf("%6d %9d | %14.2f (bad %u) | %10.2f (bad %u%s) | %10.2f (bad %u%s)\n", G, payload, 1e3 * ms_a / (3.0 * iters), h_bad_a,
                   1e3 * ms_b[0] / (3.0 * iters), h_bad_b[0], spin[0] ? ", SPIN LIMIT HIT" : "",
                   1e3 * ms_b[1] / (3.0 * iters), h_bad_b[1], spin[1] ? ", SPIN LIMIT HIT" : "");
            fflush(stdout);
            CHECK(hipFree(buf0)); CHECK(hipFree(buf1)); CHECK(hipFree(bad)); CHECK(hipFree(bar));
        }
    return 0;
}
