#include "host_util.hpp"

#include "idocp_hip.h"

namespace idocp_host {

static thread_local std::string g_last_error;

void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* last_error() { return g_last_error.c_str(); }

}  // namespace idocp_host

extern "C" const char* idocp_last_error(void) { return idocp_host::last_error(); }
extern "C" const char* idocp_version(void) { return "idocp-hip 0.1 (gfx950)"; }
