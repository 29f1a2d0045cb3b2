#include "host_util.hpp"

#include "idocp_hip.h"

namespace idocp_host {

static thread_local std::string g_last_error;

void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* last_error() { return g_last_error.c_str(); }

}  // namespace idocp_host

extern "C" const char* idocp_last_error(void) { return idocp_host::last_error(); }
extern "C" const char* idocp_version(void) { return "idocp-hip 0.1 (gfx950)"; }

extern "C" int idocp_abi_check(unsigned long model_size, unsigned long cost_size, unsigned long constraints_size) {
  if (model_size != sizeof(idocp_model_t) || cost_size != sizeof(idocp_cost_t) || constraints_size != sizeof(idocp_constraints_t)) {
    idocp_host::set_last_error("ABI mismatch: the caller was compiled against a different idocp_hip.h than libidocp_hip.so");
    return IDOCP_E_ARG;
  }
  return IDOCP_OK;
}
