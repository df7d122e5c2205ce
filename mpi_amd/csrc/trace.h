// trace.h -- named ranges for the profilers (roctx: what rocprofv3 --marker-trace shows between the kernels).
// The marker library is looked up at run time (librocprofiler-sdk-roctx.so, the one rocprofv3 listens to; libroctx64.so
// otherwise) -- libxmpi.so does not link against a profiler.  XMPI_ROCTX=1 / 0 forces ranges on / off; unset: on when a
// rocprofiler tool is attached to the process (ROCP_TOOL_LIBRARIES is what rocprofv3 exports to its child).
// The reference has no tracing of its own; an 8-rank trace of anonymous kernels is unreadable without.
#pragma once
#include <cstdarg>
#include <cstdio>

namespace xmpi {

bool roctx_enabled();
void roctx_push(const char* msg);
void roctx_pop();

// "xmpi:<what> k=v ..." from construction to the end of the scope
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char* fmt, ...) __attribute__((format(printf, 2, 3))) : on(roctx_enabled()) {
    if (!on) return;
    char buf[192];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    roctx_push(buf);
  }
  ~RoctxRange() {
    if (on) roctx_pop();
  }
  RoctxRange(const RoctxRange&) = delete;
  RoctxRange& operator=(const RoctxRange&) = delete;
};

const char* coll_name(int coll);
const char* algo_name(int algo);

}  // namespace xmpi
