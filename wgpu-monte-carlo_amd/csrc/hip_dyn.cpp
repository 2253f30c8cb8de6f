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
    if (!p) { *why = std::string("HIP runtime lacks ") + symbol; return false; }
    *slot = reinterpret_cast<F>(p);
    return true;
}

const HipApi* hip_api(const char** why) {
    std::call_once(g_once, [] {
        static std::string name;
        void* lib = open_runtime(&name);
        if (!lib) { g_why = std::string("no HIP runtime (libamdhip64) could be loaded: ") + (dlerror() ? dlerror() : ""); return; }
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
               bind(lib, "hipModuleLaunchKernel", &a.ModuleLaunchKernel, &w);
        a.library = name.c_str();
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
        static std::string name;
        void* lib = nullptr;
        const char* env = getenv("MCX_HIPRTC");
        for (const char* cand : {env ? env : "", "/opt/rocm/lib/libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so", "libhiprtc.so.7",
                                 "libhiprtc.so"}) {
            if (!*cand) continue;
            if ((lib = dlopen(cand, RTLD_NOW | RTLD_LOCAL))) { name = cand; break; }
        }
        if (!lib) { g_rtc_why = std::string("hiprtc could not be loaded: ") + (dlerror() ? dlerror() : ""); return; }
        HiprtcApi& a = g_rtc;
        std::string& w = g_rtc_why;
        g_rtc_ok = bind(lib, "hiprtcCreateProgram", &a.CreateProgram, &w) && bind(lib, "hiprtcCompileProgram", &a.CompileProgram, &w) &&
                   bind(lib, "hiprtcGetProgramLogSize", &a.GetProgramLogSize, &w) && bind(lib, "hiprtcGetProgramLog", &a.GetProgramLog, &w) &&
                   bind(lib, "hiprtcGetCodeSize", &a.GetCodeSize, &w) && bind(lib, "hiprtcGetCode", &a.GetCode, &w) &&
                   bind(lib, "hiprtcDestroyProgram", &a.DestroyProgram, &w) && bind(lib, "hiprtcVersion", &a.Version, &w) &&
                   bind(lib, "hiprtcGetErrorString", &a.GetErrorString, &w);
        a.library = name.c_str();
    });
    if (!g_rtc_ok) {
        if (why) *why = g_rtc_why.c_str();
        return nullptr;
    }
    return &g_rtc;
}

}  // namespace mcx
