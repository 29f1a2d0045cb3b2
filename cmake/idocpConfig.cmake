# find_package(idocp) for this library: what the reference's package gives a user project (README "Usage": find_package(idocp REQUIRED),
# target_link_libraries(... idocp::idocp), target_include_directories(... ${IDOCP_INCLUDE_DIR})) -- the imported target idocp::idocp and
# IDOCP_INCLUDE_DIR, pointing at this tree's facade headers and libidocp_hip.so.  Use it in place:
#     cmake -S <your project> -B build -Didocp_DIR=<repo>/cmake
# The reference's own examples/iiwa14 and examples/anymal projects configure and build with it unchanged (tests/test_reference_drivers.py).
get_filename_component(_idocp_root "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
set(IDOCP_INCLUDE_DIR "${_idocp_root}/include")
set(IDOCP_LIBRARY "${_idocp_root}/idocp_amd/lib/libidocp_hip.so")
if(NOT EXISTS "${IDOCP_LIBRARY}")
  message(FATAL_ERROR "idocp: ${IDOCP_LIBRARY} is not built (python -c \"import __graft_entry__ as g; g.build()\" in ${_idocp_root})")
endif()
if(NOT TARGET idocp::idocp)
  add_library(idocp::idocp SHARED IMPORTED)
  set_target_properties(idocp::idocp PROPERTIES
    IMPORTED_LOCATION "${IDOCP_LIBRARY}"
    IMPORTED_NO_SONAME TRUE
    INTERFACE_INCLUDE_DIRECTORIES "${IDOCP_INCLUDE_DIR}"
    INTERFACE_COMPILE_FEATURES cxx_std_17)      # the facade headers are C++17 (the reference's projects ask for 11: CMake takes the higher one)
  # Eigen: the real library where it is installed; otherwise "Eigen/Core" (which drivers include directly) forwards to the facade's stand-in types
  find_package(Eigen3 QUIET NO_MODULE)
  if(TARGET Eigen3::Eigen)
    set_property(TARGET idocp::idocp APPEND PROPERTY INTERFACE_LINK_LIBRARIES Eigen3::Eigen)
  else()
    set_property(TARGET idocp::idocp APPEND PROPERTY INTERFACE_INCLUDE_DIRECTORIES "${IDOCP_INCLUDE_DIR}/idocp/compat")
  endif()
endif()
set(idocp_FOUND TRUE)
