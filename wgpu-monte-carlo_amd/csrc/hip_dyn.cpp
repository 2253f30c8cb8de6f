// hip_dyn.cpp -- see hip_dyn.hpp
#include "hip_dyn.hpp"

#include <dlfcn.h>

#include <cstdlib>
#include <mutex>
#include <string>

namespace mcx {

static HipApi g_api;
static bool g_ok = false;
static std::string g_why;
static std::once_flag g_once;

static void* open_runtime(std::string* name) {
    // 1. a runtime that is already mapped: torch's bundled one has no soname (file name libamdhip64.so)
    for (const char* cand : {"libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"}) {
        if (void* h = dlopen(cand, RTLD_NOW | RTLD_NOLOAD)) { *name = std::string(cand) + " (already loaded)"; return h; }
    }
    // 2. an explicit choice
    if (const char* env = getenv("MCX_HIP_RUNTIME")) {
        if (void* h = dlopen(env, RTLD_NOW | RTLD_GLOBAL)) { *name = env; return h; }
    }
    // 3. the system runtime
    for (const char* cand : {"libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so.7", "libamdhip64.so"}) {
        if (void* h = dlopen(cand, RTLD_NOW | RTLD_GLOBAL)) { *name = cand; return h; }
    }
    return nullptr;
}

template <class F>
static bool bind(void* lib, const char* symbol, F* slot, std::string* why) {
    void* p = dlsym(lib, symbol);
    if (!p) { *why = std::string("the library lacks the symbol ") + symbol; return false; }
    *slot = reinterpret_cast<F>(p);
    return true;
}

const HipApi* hip_api(const char** why) {
    std::call_once(g_once, [] {
      try {
        static std::string name;
        void* lib = open_runtime(&name);
        if (!lib) {                       // dlerror() clears the error it returns: read it exactly once
            const char* err = dlerror();
            g_why = std::string("no HIP runtime (libamdhip64) could be loaded: ") + (err ? err : "");
            return;
        }
        HipApi& a = g_api;
        std::string& w = g_why;
        g_ok = bind(lib, "hipGetErrorString", &a.GetErrorString, &w) && bind(lib, "hipGetDeviceCount", &a.GetDeviceCount, &w) &&
               bind(lib, "hipGetDevicePropertiesR0600", &a.GetDeviceProperties, &w) && bind(lib, "hipSetDevice", &a.SetDevice, &w) &&
               bind(lib, "hipStreamCreateWithFlags", &a.StreamCreateWithFlags, &w) && bind(lib, "hipStreamDestroy", &a.StreamDestroy, &w) &&
               bind(lib, "hipStreamSynchronize", &a.StreamSynchronize, &w) && bind(lib, "hipEventCreate", &a.EventCreate, &w) &&
               bind(lib, "hipEventDestroy", &a.EventDestroy, &w) && bind(lib, "hipEventRecord", &a.EventRecord, &w) &&
               bind(lib, "hipEventSynchronize", &a.EventSynchronize, &w) && bind(lib, "hipEventElapsedTime", &a.EventElapsedTime, &w) &&
               bind(lib, "hipMalloc", &a.Malloc, &w) && bind(lib, "hipFree", &a.Free, &w) &&
               bind(lib, "hipHostMalloc", &a.HostMalloc, &w) && bind(lib, "hipHostFree", &a.HostFree, &w) &&
               bind(lib, "hipMemcpy", &a.Memcpy, &w) && bind(lib, "hipMemcpyAsync", &a.MemcpyAsync, &w) &&
               bind(lib, "hipMemsetAsync", &a.MemsetAsync, &w) && bind(lib, "hipModuleLoadData", &a.ModuleLoadData, &w) &&
               bind(lib, "hipModuleUnload", &a.ModuleUnload, &w) && bind(lib, "hipModuleGetFunction", &a.ModuleGetFunction, &w) &&
               bind(lib, "hipModuleLaunchKernel", &a.ModuleLaunchKernel, &w) &&
               bind(lib, "hipFuncGetAttribute", &a.FuncGetAttribute, &w) && bind(lib, "hipStreamWaitEvent", &a.StreamWaitEvent, &w) &&
               bind(lib, "hipEventCreateWithFlags", &a.EventCreateWithFlags, &w);
        a.HostGetDevicePointer = reinterpret_cast<decltype(a.HostGetDevicePointer)>(dlsym(lib, "hipHostGetDevicePointer"));
        a.library = name.c_str();
      } catch (...) { g_ok = false; g_why = "exception while binding the HIP runtime"; }
    });
    if (!g_ok) {
        if (why) *why = g_why.c_str();
        return nullptr;
    }
    return &g_api;
}

static HiprtcApi g_rtc;
static bool g_rtc_ok = false;
static std::string g_rtc_why;
static std::once_flag g_rtc_once;

const HiprtcApi* hiprtc_api(const char** why) {
    std::call_once(g_rtc_once, [] {
      try {
        static std::string name;
        void* lib = nullptr;
        const char* env = getenv("MCX_HIPRTC");
        for (const char* cand : {env ? env : "", "/opt/rocm/lib/libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so", "libhiprtc.so.7",
                                 "libhiprtc.so"}) {
            if (!*cand) continue;
            if ((lib = dlopen(cand, RTLD_NOW | RTLD_LOCAL))) { name = cand; break; }
        }
        if (!lib) {
            const char* err = dlerror();
            g_rtc_why = std::string("hiprtc could not be loaded: ") + (err ? err : "");
            return;
        }
        HiprtcApi& a = g_rtc;
        std::string& w = g_rtc_why;
        g_rtc_ok = bind(lib, "hiprtcCreateProgram", &a.CreateProgram, &w) && bind(lib, "hiprtcCompileProgram", &a.CompileProgram, &w) &&
                   bind(lib, "hiprtcGetProgramLogSize", &a.GetProgramLogSize, &w) && bind(lib, "hiprtcGetProgramLog", &a.GetProgramLog, &w) &&
                   bind(lib, "hiprtcGetCodeSize", &a.GetCodeSize, &w) && bind(lib, "hiprtcGetCode", &a.GetCode, &w) &&
                   bind(lib, "hiprtcDestroyProgram", &a.DestroyProgram, &w) && bind(lib, "hiprtcVersion", &a.Version, &w) &&
                   bind(lib, "hiprtcGetErrorString", &a.GetErrorString, &w);
        a.library = name.c_str();
      } catch (...) { g_rtc_ok = false; g_rtc_why = "exception while binding hiprtc"; }
    });
    if (!g_rtc_ok) {
        if (why) *why = g_rtc_why.c_str();
        return nullptr;
    }
    return &g_rtc;
}

static RcclApi g_rccl;
static bool g_rccl_ok = false;
static std::string g_rccl_why;
static std::once_flag g_rccl_once;

const RcclApi* rccl_api(const char** why) {
    std::call_once(g_rccl_once, [] {
      try {
        static std::string name;
        void* lib = nullptr;
        for (const char* cand : {"librccl.so", "librccl.so.1"})               // torch's bundled copy has no soname
            if ((lib = dlopen(cand, RTLD_NOW | RTLD_NOLOAD))) { name = std::string(cand) + " (already loaded)"; break; }
        if (!lib) {
            const char* env = getenv("MCX_RCCL");
            for (const char* cand : {env ? env : "", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
                if (!*cand) continue;
                if ((lib = dlopen(cand, RTLD_NOW | RTLD_GLOBAL))) { name = cand; break; }
            }
        }
        if (!lib) {
            const char* err = dlerror();
            g_rccl_why = std::string("RCCL (librccl) could not be loaded: ") + (err ? err : "");
            return;
        }
        RcclApi& a = g_rccl;
        std::string& w = g_rccl_why;
        g_rccl_ok = bind(lib, "ncclCommInitAll", &a.CommInitAll, &w) && bind(lib, "ncclCommDestroy", &a.CommDestroy, &w) &&
                    bind(lib, "ncclAllReduce", &a.AllReduce, &w) && bind(lib, "ncclGroupStart", &a.GroupStart, &w) &&
                    bind(lib, "ncclGroupEnd", &a.GroupEnd, &w) && bind(lib, "ncclGetVersion", &a.GetVersion, &w) &&
                    bind(lib, "ncclGetErrorString", &a.GetErrorString, &w);
        a.library = name.c_str();
      } catch (...) { g_rccl_ok = false; g_rccl_why = "exception while binding RCCL"; }
    });
    if (!g_rccl_ok) {
        if (why) *why = g_rccl_why.c_str();
        return nullptr;
    }
    return &g_rccl;
}

}  // namespace mcx
