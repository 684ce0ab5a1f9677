// plans.hip -- process-wide switches of the gas-optics kernels and the extension entry points that set them or read the
// per-context state (gas_optics_common.h); the plans themselves are built where they are used (tau_absorption.hip).
#include "gas_optics_common.h"

// process-wide tuning switches (set from any thread: relaxed atomics)
std::atomic<int> g_tau_force_direct{0};
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
std::atomic<int> g_tau_no_zero_check{env_int("RTE_HIP_NO_ZERO_CHECK", 0)};  // A/B: accumulate without looking whether tau is zero
// (RTE_HIP_SHARE_GEOMETRY=1: the opt-in of rte_hip_share_geometry for an unchanged binary)
std::atomic<int> g_share_geom_default{env_int("RTE_HIP_SHARE_GEOMETRY", 0)};  // what a context starts with (the last rte_hip_share_geometry of any context)

extern "C" {

int rte_hip_share_geometry(int on) {
  rte::CtxLock l;
  g_share_geom_default = on; gs().share_geom = on; gs().shared.seq = -1; gs().imask.seq = -1;
  return 0;
}
int rte_hip_force_direct_gather(int on) { g_tau_force_direct = on; return 0; }
int rte_hip_tau_zero_check(int on) { g_tau_no_zero_check = on ? 0 : 1; return 0; }
int rte_hip_invalidate_plans(void) {
  RTE_TRY
  rte::CtxLock l;
  ++gs().plan_epoch;
  rte::drop_table_copies();  // (host-mirror mode: cached device copies of host tables)
  RTE_CATCH("rte_hip_invalidate_plans")
  return 0;
}
// diagnostics (synchronises): 0 = (column tile, layer, band) triples the last compute_tau_absorption call handed to the
// direct-gather worklist, 1 = (column tile, band) pairs of the last compute_Planck_source call
int rte_hip_stat(int which) {
  if (which < 0 || which > 3) return -1;
  RTE_TRY
  rte::CtxLock l;
  int v = 0;
  HIP_CHECK(hipStreamSynchronize(rte::stream()));
  HIP_CHECK(hipMemcpy(&v, stats_dev() + which, sizeof(int), hipMemcpyDeviceToHost));
  return v;
  RTE_CATCH("rte_hip_stat")
  return -1;
}
}  // extern "C"
