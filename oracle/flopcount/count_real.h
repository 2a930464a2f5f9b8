// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// Operation-counting scalar for the instrumented build of oracle/mjcpu (-DMJC_COUNT_FLOPS).
// oracle/mjcpu/{engine,model,models}.c are compiled UNCHANGED as C++ with `double` renamed to
// this class (see oracle/Makefile, target _build/libmjc_count.so): every +, -, *, / and sqrt
// the restatement executes on an mjtNum is then counted, per pipeline stage (MJC_STAGE(k)
// markers in engine.c = rows M1-M9 of SURVEY.md section 8a).  The arithmetic itself is the
// same IEEE double arithmetic in the same order (no contraction: -ffp-contract=off), which
// tools/count_flops.py checks by comparing the counted rollout with the plain port bit for bit.
//
// Counted kinds (per stage):  add (+ and -), mul, div, sqrt, trans (sin cos tan atan2 acos asin
// exp log pow), and -- not flops, reported separately -- cmp (comparisons, fmin/fmax/fabs).
// "flops" = add + mul + div + sqrt + trans; a multiply-add pair counts 2, as in the vendor's
// peak figures.  Unary minus and copies count nothing.
#ifndef ORACLE_FLOPCOUNT_COUNT_REAL_H_
#define ORACLE_FLOPCOUNT_COUNT_REAL_H_

#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>

typedef double mjc_real_t;

enum { MJC_K_ADD = 0, MJC_K_MUL, MJC_K_DIV, MJC_K_SQRT, MJC_K_TRANS, MJC_K_CMP, MJC_NKIND };
enum { MJC_NSTAGE = 12 };  // 0 = outside any stage, 1..9 = M1..M9, 10 = mj_rnePostConstraint

extern thread_local unsigned long long mjc_count[MJC_NSTAGE][MJC_NKIND];
extern thread_local int mjc_stage;
#define MJC_TICK(kind) (++mjc_count[mjc_stage][kind])

struct CountD {
  mjc_real_t v;
  CountD() = default;
  constexpr CountD(mjc_real_t x) : v(x) {}  // NOLINT: implicit on purpose
  constexpr CountD(int x) : v(x) {}         // NOLINT
  explicit operator mjc_real_t() const { return v; }
  explicit operator int() const { return static_cast<int>(v); }
  explicit operator bool() const { return v != 0; }
  CountD operator-() const { return CountD(-v); }
  CountD operator+() const { return *this; }
  CountD& operator+=(CountD o) { MJC_TICK(MJC_K_ADD); v += o.v; return *this; }
  CountD& operator-=(CountD o) { MJC_TICK(MJC_K_ADD); v -= o.v; return *this; }
  CountD& operator*=(CountD o) { MJC_TICK(MJC_K_MUL); v *= o.v; return *this; }
  CountD& operator/=(CountD o) { MJC_TICK(MJC_K_DIV); v /= o.v; return *this; }
};

#define MJC_BINOP(op, kind)                                                                    \
  inline CountD operator op(CountD a, CountD b) { MJC_TICK(kind); return CountD(a.v op b.v); } \
  inline CountD operator op(CountD a, mjc_real_t b) { MJC_TICK(kind); return CountD(a.v op b); } \
  inline CountD operator op(mjc_real_t a, CountD b) { MJC_TICK(kind); return CountD(a op b.v); } \
  inline CountD operator op(CountD a, int b) { MJC_TICK(kind); return CountD(a.v op b); }      \
  inline CountD operator op(int a, CountD b) { MJC_TICK(kind); return CountD(a op b.v); }
MJC_BINOP(+, MJC_K_ADD)
MJC_BINOP(-, MJC_K_ADD)
MJC_BINOP(*, MJC_K_MUL)
MJC_BINOP(/, MJC_K_DIV)
#undef MJC_BINOP

#define MJC_CMPOP(op)                                                                      \
  inline bool operator op(CountD a, CountD b) { MJC_TICK(MJC_K_CMP); return a.v op b.v; }  \
  inline bool operator op(CountD a, mjc_real_t b) { MJC_TICK(MJC_K_CMP); return a.v op b; } \
  inline bool operator op(mjc_real_t a, CountD b) { MJC_TICK(MJC_K_CMP); return a op b.v; } \
  inline bool operator op(CountD a, int b) { MJC_TICK(MJC_K_CMP); return a.v op b; }       \
  inline bool operator op(int a, CountD b) { MJC_TICK(MJC_K_CMP); return a op b.v; }
MJC_CMPOP(<)
MJC_CMPOP(>)
MJC_CMPOP(<=)
MJC_CMPOP(>=)
MJC_CMPOP(==)
MJC_CMPOP(!=)
#undef MJC_CMPOP

#define MJC_FN1(name, kind) \
  inline CountD name(CountD a) { MJC_TICK(kind); return CountD(std::name(a.v)); }
MJC_FN1(sqrt, MJC_K_SQRT)
MJC_FN1(sin, MJC_K_TRANS)
MJC_FN1(cos, MJC_K_TRANS)
MJC_FN1(tan, MJC_K_TRANS)
MJC_FN1(acos, MJC_K_TRANS)
MJC_FN1(asin, MJC_K_TRANS)
MJC_FN1(atan, MJC_K_TRANS)
MJC_FN1(exp, MJC_K_TRANS)
MJC_FN1(log, MJC_K_TRANS)
MJC_FN1(fabs, MJC_K_CMP)
MJC_FN1(floor, MJC_K_CMP)
#undef MJC_FN1
#define MJC_FN2(name, kind)                                                                          \
  inline CountD name(CountD a, CountD b) { MJC_TICK(kind); return CountD(std::name(a.v, b.v)); }     \
  inline CountD name(CountD a, mjc_real_t b) { MJC_TICK(kind); return CountD(std::name(a.v, b)); }   \
  inline CountD name(mjc_real_t a, CountD b) { MJC_TICK(kind); return CountD(std::name(a, b.v)); }
MJC_FN2(atan2, MJC_K_TRANS)
MJC_FN2(pow, MJC_K_TRANS)
MJC_FN2(fmin, MJC_K_CMP)
MJC_FN2(fmax, MJC_K_CMP)
MJC_FN2(copysign, MJC_K_CMP)
#undef MJC_FN2
inline bool mjc_isfinite(CountD a) { return std::isfinite(a.v); }
inline bool mjc_isnan(CountD a) { return std::isnan(a.v); }
#undef isfinite
#undef isnan
#define isfinite(x) mjc_isfinite(x)
#define isnan(x) mjc_isnan(x)

#define MJC_STAGE(k) (mjc_stage = (k))

// from here on the C sources see CountD wherever they wrote `double`
#define double CountD
#define _Thread_local thread_local

#endif  // ORACLE_FLOPCOUNT_COUNT_REAL_H_
