// trace.cpp -- roctx ranges, the marker library resolved at run time (trace.h)
#include "trace.h"
#include "../../include/xmpi.h"

#include <dlfcn.h>

#include <cstdlib>
#include <mutex>

namespace xmpi {
namespace {

using push_fn = int (*)(const char*);
using pop_fn = int (*)();

struct Roctx {
  push_fn push = nullptr;
  pop_fn pop = nullptr;
  bool on = false;
};

const Roctx& roctx() {
  static Roctx r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* want = getenv("XMPI_ROCTX");
    const bool asked = want && atoi(want) != 0, refused = want && atoi(want) == 0;
    if (refused || (!asked && !getenv("ROCP_TOOL_LIBRARIES"))) return;
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void* h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
      if (!h) continue;
      r.push = (push_fn)dlsym(h, "roctxRangePushA");
      r.pop = (pop_fn)dlsym(h, "roctxRangePop");
      if (r.push && r.pop) {
        r.on = true;
        return;
      }
      r.push = nullptr;
      r.pop = nullptr;
    }
  });
  return r;
}

}  // namespace

bool roctx_enabled() { return roctx().on; }
void roctx_push(const char* msg) {
  if (roctx().on) (void)roctx().push(msg);
}
void roctx_pop() {
  if (roctx().on) (void)roctx().pop();
}

const char* coll_name(int coll) {
  static const char* const names[] = {"allreduce", "allgather", "bcast", "reduce"};  // plan.h CollKind
  return coll >= 0 && coll < 4 ? names[coll] : "?";
}
const char* algo_name(int algo) {
  static const char* const names[] = {"auto", "ring", "rhd", "direct", "tree", "zcopy", "zpush", "ll", "ring_push", "rhd_push", "tree_push"};  // xmpi.h xmpi_algo
  return algo >= 0 && algo < XMPI_ALGO_COUNT ? names[algo] : "?";
}

}  // namespace xmpi
