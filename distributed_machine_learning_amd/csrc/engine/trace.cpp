// Run-time binding to the roctx marker API (see dml_trace.h).
#include "dml_trace.h"

#include <dlfcn.h>

#include <atomic>

namespace dml {

namespace {
typedef int (*push_fn)(const char*);
typedef int (*pop_fn)();
std::atomic<bool> g_on{false};
push_fn g_push = nullptr;
pop_fn g_pop = nullptr;
bool g_tried = false;

void load() {
    if (g_tried) return;
    g_tried = true;
    // rocprofv3 listens to the SDK's marker library; the roctracer one is the older name
    const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so",
                           "libroctx64.so.4", "libroctx64.so"};
    for (const char* n : names) {
        void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        g_push = (push_fn)dlsym(h, "roctxRangePushA");
        g_pop = (pop_fn)dlsym(h, "roctxRangePop");
        if (g_push && g_pop) return;
        g_push = nullptr;
        g_pop = nullptr;
    }
}
}  // namespace

void set_trace_ranges(bool on) {
    if (on) load();
    g_on.store(on && g_push && g_pop);
}

bool trace_ranges_enabled() { return g_on.load(std::memory_order_relaxed); }

void trace_push(const char* name) {
    if (g_push) g_push(name);
}

void trace_pop() {
    if (g_pop) g_pop();
}

}  // namespace dml
