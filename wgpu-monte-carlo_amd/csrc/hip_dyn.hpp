// hip_dyn.hpp -- run-time binding to the HIP runtime.
//
// PyTorch-ROCm wheels bundle their own libamdhip64.so (+ libhsa-runtime64.so); a library linked against
// /opt/rocm/lib/libamdhip64.so.7 would bring a SECOND HIP runtime into the process: torch's stream handles
// (and its null stream) would not be the streams our launches are ordered on, and whichever runtime
// initialises second may find no GPU. libmcx therefore links no HIP runtime: it binds to the instance that
// is already mapped in the process (torch's, when torch is imported or was preloaded by
// wgpu_montecarlo/runtime.py), else loads MCX_HIP_RUNTIME, else the system runtime.
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hiprtc.h>
#include <rccl/rccl.h>

namespace mcx {

struct HipApi {
    const char* (*GetErrorString)(hipError_t);
    hipError_t (*GetDeviceCount)(int*);
    hipError_t (*GetDeviceProperties)(hipDeviceProp_t*, int);          // symbol hipGetDevicePropertiesR0600
    hipError_t (*SetDevice)(int);
    hipError_t (*StreamCreateWithFlags)(hipStream_t*, unsigned int);
    hipError_t (*StreamDestroy)(hipStream_t);
    hipError_t (*StreamSynchronize)(hipStream_t);
    hipError_t (*EventCreate)(hipEvent_t*);
    hipError_t (*EventDestroy)(hipEvent_t);
    hipError_t (*EventRecord)(hipEvent_t, hipStream_t);
    hipError_t (*EventSynchronize)(hipEvent_t);
    hipError_t (*EventElapsedTime)(float*, hipEvent_t, hipEvent_t);
    hipError_t (*Malloc)(void**, size_t);
    hipError_t (*Free)(void*);
    hipError_t (*HostMalloc)(void**, size_t, unsigned int);
    hipError_t (*HostFree)(void*);
    hipError_t (*Memcpy)(void*, const void*, size_t, hipMemcpyKind);
    hipError_t (*MemcpyAsync)(void*, const void*, size_t, hipMemcpyKind, hipStream_t);
    hipError_t (*MemsetAsync)(void*, int, size_t, hipStream_t);
    hipError_t (*ModuleLoadData)(hipModule_t*, const void*);
    hipError_t (*ModuleUnload)(hipModule_t);
    hipError_t (*ModuleGetFunction)(hipFunction_t*, hipModule_t, const char*);
    hipError_t (*ModuleLaunchKernel)(hipFunction_t, unsigned int, unsigned int, unsigned int, unsigned int,
                                     unsigned int, unsigned int, unsigned int, hipStream_t, void**, void**);
    hipError_t (*FuncGetAttribute)(int*, hipFunction_attribute, hipFunction_t);
    hipError_t (*StreamWaitEvent)(hipStream_t, hipEvent_t, unsigned int);
    hipError_t (*EventCreateWithFlags)(hipEvent_t*, unsigned int);
    hipError_t (*HostGetDevicePointer)(void**, void*, unsigned int);  // optional (null if the runtime lacks it)
    const char* library;      // path or soname of the runtime that was bound
};

// Bound on first use; nullptr (and an error text in `why`) if no HIP runtime can be loaded.
const HipApi* hip_api(const char** why = nullptr);

// hiprtc is bound the same way, but to ONE fixed compiler: the system ROCm's (MCX_HIPRTC overrides). torch
// ships a libhiprtc.so.7 of an older ROCm under the same soname; resolving by soname would make the compiler
// (and the code-object cache key) depend on whether torch was imported first.
struct HiprtcApi {
    hiprtcResult (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char**, const char**);
    hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char**);
    hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t*);
    hiprtcResult (*GetProgramLog)(hiprtcProgram, char*);
    hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t*);
    hiprtcResult (*GetCode)(hiprtcProgram, char*);
    hiprtcResult (*DestroyProgram)(hiprtcProgram*);
    hiprtcResult (*Version)(int*, int*);
    const char* (*GetErrorString)(hiprtcResult);
    const char* library;
};
const HiprtcApi* hiprtc_api(const char** why = nullptr);

// RCCL, bound the same way as the HIP runtime and for the same reason: PyTorch-ROCm bundles its own librccl.so
// linked against its own libamdhip64.so. Preference: an instance already mapped in the process, MCX_RCCL, the
// system library. Only the single-process entry points are used (one host thread drives every GPU of the node).
struct RcclApi {
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*GetVersion)(int*);
    const char* (*GetErrorString)(ncclResult_t);
    const char* library;
};
const RcclApi* rccl_api(const char** why = nullptr);

}  // namespace mcx
