// cvgs_ranges.cpp -- profiling ranges (reference tests/nvtx.h PUSH_RANGE/POP_RANGE) on roctx.
// libroctx64 is looked up lazily so the library has no hard dependency on the profiler SDK.
#include <dlfcn.h>

#include "../../include/cvgs_hip.h"

namespace {
using push_fn = int (*)(const char*);
using pop_fn = int (*)();
push_fn g_push = nullptr;
pop_fn g_pop = nullptr;
bool g_tried = false;

void resolve() {
    if (g_tried) return;
    g_tried = true;
    void* h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so.4", RTLD_LAZY | RTLD_GLOBAL);
    if (!h) return;
    g_push = (push_fn)dlsym(h, "roctxRangePushA");
    g_pop = (pop_fn)dlsym(h, "roctxRangePop");
}
} // namespace

extern "C" void cvgs_range_push(const char* name) {
    resolve();
    if (g_push) g_push(name ? name : "cvgs");
}

extern "C" void cvgs_range_pop(void) {
    resolve();
    if (g_pop) g_pop();
}
