// Small host-side helpers shared by the C-ABI translation units.
#ifndef IDOCP_HOST_UTIL_HPP_
#define IDOCP_HOST_UTIL_HPP_

#include <string>

namespace idocp_host {

// Thread-local message returned by idocp_last_error().
void set_last_error(const std::string& msg);
const char* last_error();

}  // namespace idocp_host

#endif  // IDOCP_HOST_UTIL_HPP_
