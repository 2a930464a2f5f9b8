// Test-infrastructure shim (NOT product code): minimal stand-in for abseil's
// LOG()/DLOG() macros so that the reference's header-only runtime
// (/root/reference/envpool/core/*.h) compiles without abseil.
// Written from the macro *usage* in the reference (LOG(INFO|ERROR|FATAL) << ...).
#ifndef ORACLE_SHIM_ABSL_LOG_LOG_H_
#define ORACLE_SHIM_ABSL_LOG_LOG_H_
#include <cstdlib>
#include <iostream>
#include <sstream>

namespace oracle_shim {
class LogLine {
 public:
  LogLine(const char* sev, bool fatal, bool enabled)
      : fatal_(fatal), enabled_(enabled) {
    if (enabled_) os_ << "[" << sev << "] ";
  }
  ~LogLine() {
    if (enabled_) std::cerr << os_.str() << std::endl;
    if (fatal_) std::abort();
  }
  template <typename T>
  LogLine& operator<<(const T& v) {
    if (enabled_) os_ << v;
    return *this;
  }

 private:
  std::ostringstream os_;
  bool fatal_, enabled_;
};
struct Voidify {
  void operator&(const LogLine&) {}
};
}  // namespace oracle_shim

#define ORACLE_SHIM_SEV_INFO "I", false, false
#define ORACLE_SHIM_SEV_WARNING "W", false, true
#define ORACLE_SHIM_SEV_ERROR "E", false, true
#define ORACLE_SHIM_SEV_FATAL "F", true, true
#define LOG(sev) ::oracle_shim::LogLine(ORACLE_SHIM_SEV_##sev)
#define DLOG(sev) \
  true ? (void)0 : ::oracle_shim::Voidify() & ::oracle_shim::LogLine("D", false, false)
#endif  // ORACLE_SHIM_ABSL_LOG_LOG_H_
