// Chain-level issue: one C-ABI call launches a whole static sub-graph of the training step.
//
// A chain is a host-side table of ops; each op names one launchable entry point of this library and gives its
// arguments as (slot, immediate) pairs: value = slots[slot] + immediate (slot < 0: the immediate alone).  The
// slot table is what changes from call to call — the base addresses of the activation arena / the inputs / the
// workspace, the stream, data-dependent row counts —, the op table is built once per shape signature by the host
// layer (detmatch_amd/chain.py).  dm_chain_run walks the table and calls the entries in order on the calling
// thread: the kernels, their order and their arguments are exactly those of the op-by-op path, only the ~15 us of
// interpreter / autograd / marshalling work per launch are gone (profiles/r04_launch_cost.txt: a launch from C is
// 2.7 us).  No hipGraph is involved (replay through the runtime's helper thread measured slower on this ROCm).
#include <string.h>

#include "dm_common.h"

struct dm_chain_entry {
  const char *name;
  const char *sig;
  int (*call)(const long long *args);
};
extern "C" const dm_chain_entry *dm_chain_table_(int *n);

extern "C" int dm_chain_fn_count(void) {
  int n;
  dm_chain_table_(&n);
  return n;
}

extern "C" int dm_chain_fn_index(const char *name) {
  int n;
  const dm_chain_entry *t = dm_chain_table_(&n);
  if (!name) return -1;
  for (int i = 0; i < n; ++i)
    if (strcmp(t[i].name, name) == 0) return i;
  return -1;
}

extern "C" const char *dm_chain_fn_name(int fn) {
  int n;
  const dm_chain_entry *t = dm_chain_table_(&n);
  return (fn >= 0 && fn < n) ? t[fn].name : NULL;
}

extern "C" const char *dm_chain_fn_signature(int fn) {
  int n;
  const dm_chain_entry *t = dm_chain_table_(&n);
  return (fn >= 0 && fn < n) ? t[fn].sig : NULL;
}

extern "C" int dm_chain_run(const dm_chain_op *ops_host, int n_ops, const long long *slots_host, int n_slots,
                            int *failed_op_host) {
  int n;
  const dm_chain_entry *t = dm_chain_table_(&n);
  if (failed_op_host) *failed_op_host = -1;
  if (n_ops < 0 || (n_ops > 0 && !ops_host) || (n_slots > 0 && !slots_host)) return DM_ERR_INVALID_ARG;
  long long a[DM_CHAIN_MAX_ARGS] = {0};
  for (int i = 0; i < n_ops; ++i) {
    const dm_chain_op &op = ops_host[i];
    // the trampoline of an entry reads exactly strlen(sig) arguments: an op with another count would hand it stale
    // values of the previous op as pointers and sizes
    if (op.fn < 0 || op.fn >= n || op.nargs < 0 || op.nargs > DM_CHAIN_MAX_ARGS ||
        op.nargs != (int)strlen(t[op.fn].sig)) {
      if (failed_op_host) *failed_op_host = i;
      return DM_ERR_INVALID_ARG;
    }
    for (int k = 0; k < op.nargs; ++k) {
      int s = op.slot[k];
      if (s >= n_slots) {
        if (failed_op_host) *failed_op_host = i;
        return DM_ERR_INVALID_ARG;
      }
      a[k] = (s >= 0 ? slots_host[s] : 0) + op.imm[k];
    }
    int rc = t[op.fn].call(a);
    if (rc != DM_OK) {
      if (failed_op_host) *failed_op_host = i;
      return rc;
    }
  }
  return DM_OK;
}
