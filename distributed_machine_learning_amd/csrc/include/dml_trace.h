// roctx ranges around the phases of a training step (forward / backward main chain / weight-
// gradient side stream / gradient buckets / manual sync / optimizer), visible in a
// `rocprofv3 --marker-trace --kernel-trace` run.  The reference's only instrumentation is a
// wall-clock bracket around the iteration (/root/reference/part1/main.py:36,53); ranges let a
// profile attribute host enqueue time to phases.
//
// The marker library is looked up at run time (dlopen) the first time ranges are switched on,
// so the extension has no link-time dependency on a profiler; when off, a range costs one load.
#pragma once

namespace dml {

void set_trace_ranges(bool on);     // returns silently if no roctx library can be loaded
bool trace_ranges_enabled();
void trace_push(const char* name);
void trace_pop();

struct TraceRange {
    explicit TraceRange(const char* name) : on_(trace_ranges_enabled()) {
        if (on_) trace_push(name);
    }
    ~TraceRange() {
        if (on_) trace_pop();
    }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;

  private:
    bool on_;
};

}  // namespace dml
