// The one piece of process-wide state of the library: the tuning / A-B struct (cd360_tuning.h).  Written only by cd360_set_tuning,
// read by the launch functions.  Defaults (-1 everywhere) select the measured-best kernel for every shape; nothing here changes
// results except `whatif`, which only a -DCD360_WHATIF probe build honours.
#include "cd360_common.h"
#include "cd360_tuning.h"
#include <string.h>

namespace {
cd360_tuning make_default() {
  cd360_tuning t;
  memset(&t, 0xff, sizeof(t));  // every int32 field = -1
  t.size = (int32_t)sizeof(cd360_tuning);
  return t;
}
cd360_tuning g_tuning = make_default();
}  // namespace

const cd360_tuning& cd360_tune() { return g_tuning; }

// t == NULL restores the defaults.  Not thread-safe against concurrent launches (set it between launches, as an A/B harness does).
extern "C" int cd360_set_tuning(const cd360_tuning* t) {
  if (!t) {
    g_tuning = make_default();
    return CD360_OK;
  }
  if (t->size != (int32_t)sizeof(cd360_tuning)) return CD360_ERR_ARG;
  g_tuning = *t;
#ifndef CD360_WHATIF
  g_tuning.whatif = -1;
#endif
  return CD360_OK;
}

extern "C" int cd360_get_tuning(cd360_tuning* t) {
  if (!t) return CD360_ERR_ARG;
  *t = g_tuning;
  return CD360_OK;
}

// 1 when the library was built with -DCD360_WHATIF (what-if timing bits and probe kernels compiled in), else 0
extern "C" int cd360_whatif_build(void) {
#ifdef CD360_WHATIF
  return 1;
#else
  return 0;
#endif
}
