// Shim for envpool/core/xla.h (which needs the un-vendored xla/ffi headers and a
// CUDA driver): py_envpool.h only needs the names below for PyEnvPool::Xla().
// Found first on the include path (-Iintegration/refbind/shims before
// -I/root/reference), so the reference's py_envpool.h itself compiles unchanged.
// The device-resident analogue of the XLA custom call is epa_send_device /
// epa_recv_device (include/envpool_amd.h).
#ifndef INTEGRATION_REFBIND_SHIMS_XLA_H_
#define INTEGRATION_REFBIND_SHIMS_XLA_H_

#include <pybind11/pybind11.h>

#include <stdexcept>
#include <tuple>
#include <type_traits>

#include "envpool/core/array.h"
#include "envpool/core/dict.h"

template <typename EnvPool>
struct XlaSend {};
template <typename EnvPool>
struct XlaRecv {};
template <typename EnvPool, typename Call>
struct CustomCall {
  static pybind11::object Xla(EnvPool* /*pool*/) {
    throw std::runtime_error("XLA is not available in this build");
  }
};

// py_envpool.h:222-229 guards (the originals live in xla_template.h / xla.h)
template <typename Dict>
bool HasContainerType(const Dict& /*specs*/) {
  return false;
}
template <typename Dict>
bool HasDynamicDim(const Dict& /*specs*/) {
  return false;
}

#endif  // INTEGRATION_REFBIND_SHIMS_XLA_H_
