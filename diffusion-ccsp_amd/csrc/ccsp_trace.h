// ccsp_trace.h -- phase-trace macros of the profiling builds (-DCCSP_TRACE / -DCCSP_TRACE2); empty in the product build.
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.

// Profiling builds only (tools/trace_build.py compiles with -DCCSP_TRACE): s_memtime stamps at the phase boundaries of the
// evaluation kernels, one record per sampled workgroup, read back through ccsp_debug_trace.  The product build has none.
#ifdef CCSP_TRACE
__device__ unsigned long long g_trace[3 * 256 * 32];
#define CCSP_TRK(kern, k)                                                                             \
    do {                                                                                              \
        if (threadIdx.x == 0 && (blockIdx.x & 7) == 0 && blockIdx.x < 2048)                           \
            g_trace[((kern) * 256 + (blockIdx.x >> 3)) * 32 + (k)] = __builtin_amdgcn_s_memtime();    \
    } while (0)
// the same on the chip-wide 100 MHz clock (s_memtime counters are per shader engine: not comparable across workgroups)
#define CCSP_TRK_RT(kern, k)                                                                          \
    do {                                                                                              \
        if (threadIdx.x == 0 && (blockIdx.x & 7) == 0 && blockIdx.x < 2048)                           \
            g_trace[((kern) * 256 + (blockIdx.x >> 3)) * 32 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define CCSP_TRK(kern, k) do { } while (0)
#define CCSP_TRK_RT(kern, k) do { } while (0)
#endif

// Second profiling build (tools/trace2_build.py, -DCCSP_TRACE2; round 5): the phase boundaries of EVERY workgroup of k_rowgemm_h2 at the product
// kernel's own residency (three workgroups per CU in MODE 0), with the hardware slot the workgroup ran on (HW_ID: shader engine, CU, SIMD of
// wave 0; XCC_ID), so that the phases of the workgroups that SHARE a compute unit can be laid next to each other on that CU's own clock.  Stamps
// go to LDS (one ds_write_b32 of lane 0, a dword each: the low half of s_memtime) and leave for global memory once, at the kernel's end.
#ifdef CCSP_TRACE2
__device__ unsigned int g_trace2[4096 * 40];
// (scalar stores: no vector register, no exec-mask change, nothing added to the kernel's 168-VGPR budget; s_dcache_wb at the end)
#define CCSP_TRK2_DECL unsigned int* const trk2_ptr = g_trace2 + (size_t)(blockIdx.x < 4096 ? blockIdx.x : 4095) * 40;
#define CCSP_TRK2(k)                                                                                                              \
    do {                                                                                                                          \
        const unsigned int lo_ = (unsigned int)__builtin_amdgcn_s_memtime();                                                      \
        const unsigned int off_ = 4u * (unsigned int)(k);                                                                         \
        asm volatile("s_store_dword %0, %1, %2 glc" :: "s"(lo_), "s"(trk2_ptr), "s"(off_) : "memory");                            \
    } while (0)
#define CCSP_TRK2_FLUSH()                                                                                                         \
    do {                                                                                                                          \
        unsigned int h0_, h1_;                                                                                                    \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h0_));                                                         \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(h1_));                                                        \
        const unsigned int rt_ = (unsigned int)__builtin_amdgcn_s_memrealtime();                                                  \
        const unsigned int o0_ = 4u * 36u, o1_ = 4u * 37u, o2_ = 4u * 38u;                                                        \
        asm volatile("s_store_dword %0, %1, %2 glc" :: "s"(h0_), "s"(trk2_ptr), "s"(o0_) : "memory");                             \
        asm volatile("s_store_dword %0, %1, %2 glc" :: "s"(h1_), "s"(trk2_ptr), "s"(o1_) : "memory");                             \
        asm volatile("s_store_dword %0, %1, %2 glc" :: "s"(rt_), "s"(trk2_ptr), "s"(o2_) : "memory");                             \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");                                                        \
    } while (0)
#else
#define CCSP_TRK2_DECL
#define CCSP_TRK2(k) do { } while (0)
#define CCSP_TRK2_FLUSH() do { } while (0)
#endif
