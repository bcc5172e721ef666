// Per-kernel HIP-event timing of everything libdycon_hip.so launches (bench.py's roofline object; diagnostics).
//
// hipcc lowers every `kernel<<<grid, block, lds, stream>>>(...)` of this library to a call of hipLaunchKernel.  This file DEFINES
// that symbol and the link makes it LOCAL to the library (exports.map: only dycon_* is exported), so the library's own launch sites
// bind to it at link time, no other library is affected, and it forwards to the runtime's hipLaunchKernel, looked up once with dlsym.  While timing is switched on
// (dycon_kernel_timing(1)), each launch is bracketed by two timing events recorded on the stream the kernel is launched on, created
// with hipEventDisableSystemFence (no cache write-back / invalidate at the record); otherwise the forwarder adds one branch.
// The durations are therefore KERNEL durations -- a finalize / reduce launch that an entry point enqueues after its main kernel is
// a record of its own -- which is what `rocprofv3 --kernel-trace --stats` reports as well.
#include <cxxabi.h>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/dycon_hip.h"

void dycon_set_error(const char* fmt, ...);

namespace {
typedef hipError_t (*launch_fn)(const void*, dim3, dim3, void**, size_t, hipStream_t);

launch_fn real_launch() {
    static launch_fn fn = [] {
        void* h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_GLOBAL);
        return h ? (launch_fn)dlsym(h, "hipLaunchKernel") : (launch_fn) nullptr;
    }();
    return fn;
}

struct Rec {
    hipEvent_t e0, e1;
    hipStream_t stream;
    int name;
};
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<std::string> g_names;
std::map<const void*, int> g_name_of;

int name_id(const void* fn, hipStream_t stream) {
    auto it = g_name_of.find(fn);
    if (it != g_name_of.end()) return it->second;
    const char* raw = hipKernelNameRefByPtr(fn, stream);
    std::string s = raw ? raw : "?";
    int st = 0;
    char* dm = abi::__cxa_demangle(s.c_str(), nullptr, nullptr, &st);
    if (st == 0 && dm) s = dm;
    free(dm);
    size_t par = s.rfind('(');                       // drop the argument list, keep "name<template args>"
    if (par != std::string::npos && s.find('>') != std::string::npos && par > s.find('<')) s = s.substr(0, par);
    else if (par != std::string::npos && s.find('<') == std::string::npos) s = s.substr(0, par);
    if (s.compare(0, 5, "void ") == 0) s = s.substr(5);
    g_names.push_back(s);
    return g_name_of[fn] = (int)g_names.size() - 1;
}
}  // namespace

extern "C" hipError_t hipLaunchKernel(const void* fn, dim3 grid, dim3 block, void** args, size_t lds, hipStream_t stream) {
    launch_fn real = real_launch();
    if (!real) return hipErrorNotInitialized;
    if (!g_on) return real(fn, grid, block, args, lds, stream);
    Rec r;
    r.stream = stream;
    r.name = name_id(fn, stream);
    if (hipEventCreateWithFlags(&r.e0, hipEventDisableSystemFence) != hipSuccess && hipEventCreate(&r.e0) != hipSuccess)
        return hipErrorOutOfMemory;
    if (hipEventCreateWithFlags(&r.e1, hipEventDisableSystemFence) != hipSuccess && hipEventCreate(&r.e1) != hipSuccess)
        return hipErrorOutOfMemory;
    (void)hipEventRecord(r.e0, stream);
    hipError_t e = real(fn, grid, block, args, lds, stream);
    (void)hipEventRecord(r.e1, stream);
    g_recs.push_back(r);
    return e;
}

extern "C" int dycon_kernel_timing(int on) {
    if (on) {
        for (Rec& r : g_recs) {
            (void)hipEventDestroy(r.e0);
            (void)hipEventDestroy(r.e1);
        }
        g_recs.clear();
    }
    g_on = on != 0;
    return DYCON_OK;
}

extern "C" long long dycon_kernel_timing_count(void) { return (long long)g_recs.size(); }

extern "C" int dycon_kernel_timing_fetch(long long first, long long n, float* ms, unsigned long long* stream, int* name) {
    if (first < 0 || n < 0 || first + n > (long long)g_recs.size()) {
        dycon_set_error("dycon_kernel_timing_fetch: range [%lld, %lld) outside the %zu recorded launches", first, first + n, g_recs.size());
        return DYCON_ERR_INVALID;
    }
    for (long long i = 0; i < n; ++i) {
        Rec& r = g_recs[first + i];
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms[i], r.e0, r.e1) != hipSuccess) {
            dycon_set_error("dycon_kernel_timing_fetch: launch %lld has not completed cleanly", first + i);
            return DYCON_ERR_LAUNCH;
        }
        stream[i] = (unsigned long long)(uintptr_t)r.stream;
        name[i] = r.name;
    }
    return DYCON_OK;
}

extern "C" const char* dycon_kernel_timing_name(int id) {
    return (id >= 0 && id < (int)g_names.size()) ? g_names[id].c_str() : "";
}
