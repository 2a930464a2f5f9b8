// Test-infrastructure shim (NOT product code): CHECK*/DCHECK* macros used by
// the reference's core headers, without abseil.
#ifndef ORACLE_SHIM_ABSL_LOG_CHECK_H_
#define ORACLE_SHIM_ABSL_LOG_CHECK_H_
#include "absl/log/log.h"

#define ORACLE_SHIM_CHECK_IMPL(cond, text)                              \
  (cond) ? (void)0                                                      \
         : ::oracle_shim::Voidify() &                                   \
               ::oracle_shim::LogLine("F", true, true)                  \
                   << __FILE__ << ":" << __LINE__ << " Check failed: " text " "
#define CHECK(c) ORACLE_SHIM_CHECK_IMPL((c), #c)
#define CHECK_EQ(a, b) ORACLE_SHIM_CHECK_IMPL((a) == (b), #a " == " #b)
#define CHECK_NE(a, b) ORACLE_SHIM_CHECK_IMPL((a) != (b), #a " != " #b)
#define CHECK_LE(a, b) ORACLE_SHIM_CHECK_IMPL((a) <= (b), #a " <= " #b)
#define CHECK_LT(a, b) ORACLE_SHIM_CHECK_IMPL((a) < (b), #a " < " #b)
#define CHECK_GE(a, b) ORACLE_SHIM_CHECK_IMPL((a) >= (b), #a " >= " #b)
#define CHECK_GT(a, b) ORACLE_SHIM_CHECK_IMPL((a) > (b), #a " > " #b)
// NDEBUG-style: debug checks compile to nothing but keep the stream syntax.
#define ORACLE_SHIM_DCHECK_IMPL(expr) \
  true ? (void)0 : ::oracle_shim::Voidify() & ::oracle_shim::LogLine("D", false, false)
#define DCHECK(c) ORACLE_SHIM_DCHECK_IMPL(c)
#define DCHECK_EQ(a, b) ORACLE_SHIM_DCHECK_IMPL((a) == (b))
#define DCHECK_NE(a, b) ORACLE_SHIM_DCHECK_IMPL((a) != (b))
#define DCHECK_LE(a, b) ORACLE_SHIM_DCHECK_IMPL((a) <= (b))
#define DCHECK_LT(a, b) ORACLE_SHIM_DCHECK_IMPL((a) < (b))
#define DCHECK_GE(a, b) ORACLE_SHIM_DCHECK_IMPL((a) >= (b))
#define DCHECK_GT(a, b) ORACLE_SHIM_DCHECK_IMPL((a) > (b))
#endif  // ORACLE_SHIM_ABSL_LOG_CHECK_H_
