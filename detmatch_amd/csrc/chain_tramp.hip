// Call trampolines of the chain interpreter (chain.hip): host code only, no kernels.
//
// This translation unit deliberately does NOT include include/detmatch_hip.h: chain_tramp.inc re-declares every
// launchable entry point with its ABI-level argument classes (all pointers `void *`) so that one generated
// trampoline per entry can unpack the interpreter's uniform 64-bit argument array.  extern "C" symbols carry no
// types, so the calls bind to the real definitions; the letters of each signature come from the same table the
// ctypes binding uses (tools/gen_chain_tramp.py, detmatch_amd/_lib.py:SIGNATURES) and tests/test_chain.py checks
// the committed .inc against it.
#include <stddef.h>
#include <string.h>

struct dm_chain_entry {
  const char *name;
  const char *sig;   // one letter per argument: p pointer, i int, l long long, z size_t, f float, d double
  int (*call)(const long long *args);
};

static inline float dm_chain_f32(long long bits) {
  unsigned int u = (unsigned int)(unsigned long long)bits;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static inline double dm_chain_f64(long long bits) {
  double d;
  memcpy(&d, &bits, 8);
  return d;
}

#include "chain_tramp.inc"

extern "C" const dm_chain_entry *dm_chain_table_(int *n) {
  *n = DM_CHAIN_TABLE_N;
  return DM_CHAIN_TABLE;
}
