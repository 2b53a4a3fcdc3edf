// ccsp_host_rccl.h -- host objects, part 1: exp_env (experiment switches) and RCCL bound at run time by dlopen.
// A fragment of the ONE translation unit csrc/ccsp_hip.hip (included there, at this position, inside its namespaces): not a standalone header.
// ==========================================================================================
// host objects
// ==========================================================================================

// RCCL, bound at run time (dlopen): the library has no link-time dependency on it, and a process that already carries an RCCL
// (PyTorch-ROCm ships one) gets that same instance.  Only what the MALA global-batch reduction needs.
namespace {
// Variants that lost their same-call A/Bs (profiles/r0*_findings.md) are compiled only with -DCCSP_EXPERIMENTS
// (diffusion-ccsp_amd/_lib.py build(experiments=True) -> libccsp_hip_exp.so; tests marked gpu_experiments); their switches are read through
// exp_env, which is nullptr in the product build.
#ifdef CCSP_EXPERIMENTS
inline const char* exp_env(const char* name) { return getenv(name); }
#else
inline const char* exp_env(const char*) { return nullptr; }
#endif
}  // namespace

namespace {
struct RcclId { char internal[128]; };        // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
struct RcclApi {
    void* lib = nullptr;
    int version = 0;
    int (*get_unique_id)(RcclId*) = nullptr;
    int (*comm_init_rank)(void**, int, RcclId, int) = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    int (*comm_count)(void*, int*) = nullptr;
    int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*error_string)(int) = nullptr;
};
// The instance the process already carries is found by its soname (PyTorch-ROCm loads librccl.so.1): RTLD_NOLOAD first, so that a second
// RCCL from /opt/rocm is never mapped next to torch's; then the versioned name, then the unversioned one.  ncclFloat32 = 7 and ncclSum = 0
// and the by-value 128-byte id are the NCCL 2.x ABI: ncclGetVersion must report major version 2.
RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("CCSP_RCCL_LIB");
        if (env && *env) api.lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
        for (const char* n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!api.lib) return;
        api.get_unique_id = (int (*)(RcclId*))dlsym(api.lib, "ncclGetUniqueId");
        api.comm_init_rank = (int (*)(void**, int, RcclId, int))dlsym(api.lib, "ncclCommInitRank");
        api.comm_destroy = (int (*)(void*))dlsym(api.lib, "ncclCommDestroy");
        api.comm_count = (int (*)(void*, int*))dlsym(api.lib, "ncclCommCount");
        api.all_reduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(api.lib, "ncclAllReduce");
        api.error_string = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
        int (*get_version)(int*) = (int (*)(int*))dlsym(api.lib, "ncclGetVersion");
        if (get_version) get_version(&api.version);
        const int major = api.version >= 10000 ? api.version / 10000 : api.version / 1000;     // NCCL_VERSION_CODE: X*10000 + Y*100 + Z since 2.9
        if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_reduce || !api.comm_count || major != 2) {
            dlclose(api.lib);
            api.lib = nullptr;
        }
    });
    return api.lib ? &api : nullptr;
}
const char* rccl_err(RcclApi* a, int rc) { return a && a->error_string ? a->error_string(rc) : "?"; }
}  // namespace

