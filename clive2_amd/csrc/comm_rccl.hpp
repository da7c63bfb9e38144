// comm_rccl.hpp -- the one collective of the multi-GPU sample split: an in-place RCCL
// all-reduce (sum) of the packed accumulators acc[8][B] on the renderer's own stream, plus a tiny
// all-reduce of host doubles (barrier, max-over-ranks timing, ray tallies).
//
// The reference has no multi-device code at all; samples are i.i.d. and the accumulators
// are pure sums (src/renderer.py:269-273), so each rank renders its own samples of the replicated
// scene and the sums are combined once.
//
// librccl is loaded with dlopen on the first communicator call: a one-GPU process never maps the
// 570 MB library, and the library itself has no link-time dependency on it.  RCCL and this library
// then share the ROCm runtime of /opt/rocm (no second HIP runtime in the process).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <mutex>
#include <string>

namespace cl2 {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    std::string error;
};

// Returns nullptr (and fills `why`) when librccl cannot be loaded or lacks a symbol.
inline RcclApi* rccl_api(std::string& why) {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) { api.error = std::string("dlopen(librccl.so.1): ") + dlerror(); return; }
#define CL2_SYM(field, name)                                                                \
        api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name));         \
        if (!api.field && api.error.empty()) api.error = std::string("librccl lacks ") + name
        CL2_SYM(GetUniqueId, "ncclGetUniqueId");
        CL2_SYM(CommInitRank, "ncclCommInitRank");
        CL2_SYM(CommDestroy, "ncclCommDestroy");
        CL2_SYM(CommAbort, "ncclCommAbort");
        CL2_SYM(AllReduce, "ncclAllReduce");
        CL2_SYM(CommGetAsyncError, "ncclCommGetAsyncError");
        CL2_SYM(GetErrorString, "ncclGetErrorString");
        CL2_SYM(CommCount, "ncclCommCount");
        CL2_SYM(CommUserRank, "ncclCommUserRank");
        CL2_SYM(CommCuDevice, "ncclCommCuDevice");
#undef CL2_SYM
    });
    if (!api.error.empty()) { why = api.error; return nullptr; }
    return &api;
}

}  // namespace cl2
