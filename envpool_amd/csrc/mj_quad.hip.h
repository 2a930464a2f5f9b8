// Lane-group ("quad") execution layer for the MuJoCo kernels that split ONE env
// over FOUR adjacent lanes of a wavefront (mj_ant4.hip.h: one lane per leg).
//
// The solver code is written once against a tiny vocabulary
//   Sel(c, a, b)   per-lane select            Sum4(x) / Max4(x)  all-reduce over the quad
//   All4(c)        and-reduce over the quad   AnyWave(c)         wave-wide "any" (scalar branch)
//   MaskSet / MaskSame                        per-lane active-set bit masks
// with two realisations:
//   * device: a value IS a lane's scalar (double / float), conditions are `bool`,
//     and the quad reductions are two DPP quad_perm butterflies (v_mov_dpp + add,
//     full rate, no LDS traffic).  Both butterfly steps add the same two numbers
//     in every lane (fp addition is commutative), so the four lanes of a quad hold
//     BIT-IDENTICAL results and every per-env decision derived from them is
//     automatically uniform over the quad;
//   * host (tests/cpu_harness): a value is Q4<T>, four lanes in a struct, with the
//     same reduction order -- so the product source runs on a CPU box against
//     oracle/mjcpu.
#ifndef ENVPOOL_AMD_CSRC_MJ_QUAD_HIP_H_
#define ENVPOOL_AMD_CSRC_MJ_QUAD_HIP_H_

#include "mj_cheetah.hip.h"  // EPA_HD, static_for, WaveAny, SinCos, Rsqrt

namespace epa {
namespace mj {

// ---- host emulation types -----------------------------------------------------
struct B4 {
  bool v[4];
  friend inline B4 operator&(B4 a, B4 b) { return {{a.v[0] && b.v[0], a.v[1] && b.v[1], a.v[2] && b.v[2], a.v[3] && b.v[3]}}; }
  friend inline B4 operator|(B4 a, B4 b) { return {{a.v[0] || b.v[0], a.v[1] || b.v[1], a.v[2] || b.v[2], a.v[3] || b.v[3]}}; }
  friend inline B4 operator!(B4 a) { return {{!a.v[0], !a.v[1], !a.v[2], !a.v[3]}}; }
};
struct U4 {
  unsigned v[4];
};

template <typename T>
struct Q4 {
  T v[4];
  Q4() = default;
  template <typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type>
  Q4(U x) : v{(T)x, (T)x, (T)x, (T)x} {}  // NOLINT: broadcast
#define EPA_Q4_BIN(op)                                                         \
  friend inline Q4 operator op(const Q4& a, const Q4& b) {                     \
    Q4 r;                                                                      \
    for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] op b.v[i];                     \
    return r;                                                                  \
  }
  EPA_Q4_BIN(+)
  EPA_Q4_BIN(-)
  EPA_Q4_BIN(*)
  EPA_Q4_BIN(/)
#undef EPA_Q4_BIN
  friend inline Q4 operator-(const Q4& a) {
    Q4 r;
    for (int i = 0; i < 4; ++i) r.v[i] = -a.v[i];
    return r;
  }
  Q4& operator+=(const Q4& b) { return *this = *this + b; }
  Q4& operator-=(const Q4& b) { return *this = *this - b; }
  Q4& operator*=(const Q4& b) { return *this = *this * b; }
#define EPA_Q4_CMP(op)                                                         \
  friend inline B4 operator op(const Q4& a, const Q4& b) {                     \
    B4 r;                                                                      \
    for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] op b.v[i];                     \
    return r;                                                                  \
  }
  EPA_Q4_CMP(<)
  EPA_Q4_CMP(<=)
  EPA_Q4_CMP(>)
  EPA_Q4_CMP(>=)
  EPA_Q4_CMP(==)
  EPA_Q4_CMP(!=)
#undef EPA_Q4_CMP
};

template <typename T>
inline Q4<T> sqrt(const Q4<T>& x) {  // found by ADL from mj::Sqrt
  Q4<T> r;
  for (int i = 0; i < 4; ++i) r.v[i] = std::sqrt(x.v[i]);
  return r;
}
template <typename T>
inline void SinCos(Q4<T> x, Q4<T>* s, Q4<T>* c) {  // more specialised than mj::SinCos<T>
  for (int i = 0; i < 4; ++i) {
    s->v[i] = std::sin(x.v[i]);
    c->v[i] = std::cos(x.v[i]);
  }
}

// ---- the vocabulary: host realisation -------------------------------------------
template <typename T>
inline Q4<T> Sel(B4 c, const Q4<T>& a, const Q4<T>& b) {
  Q4<T> r;
  for (int i = 0; i < 4; ++i) r.v[i] = c.v[i] ? a.v[i] : b.v[i];
  return r;
}
template <typename T>
inline Q4<T> Sum4(const Q4<T>& x) {  // same association as the DPP butterfly
  return Q4<T>((x.v[0] + x.v[1]) + (x.v[2] + x.v[3]));
}
template <typename T>
inline Q4<T> Max4(const Q4<T>& x) {
  T a = x.v[0] > x.v[1] ? x.v[0] : x.v[1], b = x.v[2] > x.v[3] ? x.v[2] : x.v[3];
  return Q4<T>(a > b ? a : b);
}
inline B4 All4(B4 c) {
  bool a = c.v[0] && c.v[1] && c.v[2] && c.v[3];
  return {{a, a, a, a}};
}
inline bool AnyWave(B4 c) { return c.v[0] || c.v[1] || c.v[2] || c.v[3]; }
inline void MaskSet(U4& m, B4 on, int bit) {
  for (int i = 0; i < 4; ++i) m.v[i] |= (on.v[i] ? 1u : 0u) << bit;
}
inline B4 MaskSame(const U4& a, const U4& b) {
  return {{a.v[0] == b.v[0], a.v[1] == b.v[1], a.v[2] == b.v[2], a.v[3] == b.v[3]}};
}
inline U4 MaskFill(U4, unsigned x) { return {{x, x, x, x}}; }
inline int MaskCount4(const U4& m) { return __builtin_popcount(m.v[0] | m.v[1] | m.v[2] | m.v[3]); }
// per-lane sets of small integers (sphere classes): a lane pops ITS lowest member per trip of a wave-level loop
inline B4 AnySlot(const U4& rem) { return {{rem.v[0] != 0u, rem.v[1] != 0u, rem.v[2] != 0u, rem.v[3] != 0u}}; }
inline U4 PopSlot(U4& rem) {  // (0 where the set is empty: the caller masks with AnySlot)
  U4 r;
  for (int i = 0; i < 4; ++i) {
    r.v[i] = rem.v[i] ? (unsigned)__builtin_ctz(rem.v[i]) : 0u;
    rem.v[i] &= rem.v[i] - 1u;
  }
  return r;
}
inline U4 SlotsWhere(const U4& own, B4 on) {
  return {{on.v[0] ? own.v[0] : 0u, on.v[1] ? own.v[1] : 0u, on.v[2] ? own.v[2] : 0u, on.v[3] ? own.v[3] : 0u}};
}
inline U4 MaskClear(const U4& m, B4 where, unsigned bits) {
  U4 r;
  for (int i = 0; i < 4; ++i) r.v[i] = where.v[i] ? (m.v[i] & ~bits) : m.v[i];
  return r;
}
inline B4 UGe(const U4& a, unsigned b) { return {{a.v[0] >= b, a.v[1] >= b, a.v[2] >= b, a.v[3] >= b}}; }
inline B4 UEq(const U4& a, unsigned b) { return {{a.v[0] == b, a.v[1] == b, a.v[2] == b, a.v[3] == b}}; }
// 6-bit entry number idx of a packed table
inline U4 Tab6(unsigned long long tab, const U4& idx) {
  U4 r;
  for (int i = 0; i < 4; ++i) r.v[i] = (unsigned)(tab >> (6u * idx.v[i])) & 63u;
  return r;
}
inline U4 UMad(const U4& a, unsigned mul, unsigned add) {
  return {{a.v[0] * mul + add, a.v[1] * mul + add, a.v[2] * mul + add, a.v[3] * mul + add}};
}
inline void MaskSetAt(U4& m, B4 on, const U4& bit) {
  for (int i = 0; i < 4; ++i) m.v[i] |= (on.v[i] ? 1u : 0u) << bit.v[i];
}
// lane i writes ITS slot slot.v[i] of a per-lane block, where `on`
template <typename Lds, typename T>
inline void ScatterSlot(Lds&& lds, const U4& slot, const Q4<T>& x, B4 on) {
  for (int i = 0; i < 4; ++i) {
    if (on.v[i]) lds((int)slot.v[i]).v[i] = x.v[i];
  }
}
// lane i reads ITS slot slot.v[i] of a per-lane block
template <typename Lds>
inline auto GatherSlot(Lds&& lds, const U4& slot) -> typename std::decay<decltype(lds(0))>::type {
  typename std::decay<decltype(lds(0))>::type r;
  for (int i = 0; i < 4; ++i) r.v[i] = lds((int)slot.v[i]).v[i];
  return r;
}
template <typename T>
inline Q4<T> Rsq(const Q4<T>& x) {
  Q4<T> r;
  for (int i = 0; i < 4; ++i) r.v[i] = T(1) / std::sqrt(x.v[i]);
  return r;
}
template <typename T>
inline B4 IsFinite(const Q4<T>& x) {
  return {{std::isfinite(x.v[0]), std::isfinite(x.v[1]), std::isfinite(x.v[2]), std::isfinite(x.v[3])}};
}

// ---- the vocabulary: device realisation (and the scalar host one, for 1-lane tests)
template <typename T>
EPA_HD T Sel(bool c, T a, T b) {
  return c ? a : b;
}
EPA_HD float Rsq(float x) { return Rsqrt(x); }
EPA_HD double Rsq(double x) { return Rsqrt(x); }
EPA_HD bool AnyWave(bool c) { return WaveAny(c); }
EPA_HD void MaskSet(unsigned& m, bool on, int bit) { m |= (on ? 1u : 0u) << bit; }
EPA_HD bool MaskSame(unsigned a, unsigned b) { return a == b; }
EPA_HD unsigned MaskFill(unsigned, unsigned x) { return x; }
EPA_HD bool AnySlot(unsigned rem) { return rem != 0u; }
EPA_HD unsigned PopSlot(unsigned& rem) {
  const unsigned s = rem ? (unsigned)__builtin_ctz(rem) : 0u;
  rem &= rem - 1u;
  return s;
}
EPA_HD unsigned SlotsWhere(unsigned own, bool on) { return on ? own : 0u; }
EPA_HD unsigned MaskClear(unsigned m, bool where, unsigned bits) { return where ? (m & ~bits) : m; }
EPA_HD bool UGe(unsigned a, unsigned b) { return a >= b; }
EPA_HD bool UEq(unsigned a, unsigned b) { return a == b; }
EPA_HD unsigned Tab6(unsigned long long tab, unsigned idx) { return (unsigned)(tab >> (6u * idx)) & 63u; }
EPA_HD unsigned UMad(unsigned a, unsigned mul, unsigned add) { return a * mul + add; }
EPA_HD void MaskSetAt(unsigned& m, bool on, unsigned bit) { m |= (on ? 1u : 0u) << bit; }
template <typename Lds, typename T>
EPA_HD void ScatterSlot(Lds&& lds, unsigned slot, T x, bool on) {
  if (on) lds((int)slot) = x;
}
template <typename Lds>
EPA_HD auto GatherSlot(Lds&& lds, unsigned slot) -> typename std::decay<decltype(lds(0))>::type {
  return lds((int)slot);  // the device accessor takes a per-lane slot number as it takes a uniform one
}
#if defined(__HIP_DEVICE_COMPILE__)
// quad_perm selectors: [1,0,3,2] swaps neighbours, [2,3,0,1] swaps pairs
constexpr int kDppSwap1 = 0xB1, kDppSwap2 = 0x4E;
template <int CTRL>
__device__ __forceinline__ int DppMov(int x) {
  // bound_ctrl = true: a quad_perm never reads out of bounds, and with full row/bank masks the compiler then
  // knows the `old` operand is dead -- with bound_ctrl = false every DPP move carries a `v_mov_b32 dst, 0` in
  // front of it (864 of the 1116 DPP moves of the round-2 Humanoid kernel).  Measured: Ant +3 % without those
  // moves.  The round-2 Humanoid quad kernel was 1.2 % slower without them and kept them (-DEPA_DPP_OLD_ZERO,
  // profiles/archive/r2u_bench.jsonl); since the hybrid-PGS rewrite of round 3 it is faster WITHOUT them
  // (profiles/archive/r3s_standup_hybrid_pgs.md: Humanoid 10.85 -> 9.97 ms with the leaner visits + bound_ctrl), so no
  // TU of the Makefile defines EPA_DPP_OLD_ZERO any more; the switch stays for A/B builds.
#ifdef EPA_DPP_OLD_ZERO
  return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, false);
#else
  return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
#endif
}
template <int CTRL>
__device__ __forceinline__ float DppMov(float x) {
  return __builtin_bit_cast(float, DppMov<CTRL>(__builtin_bit_cast(int, x)));
}
template <int CTRL>
__device__ __forceinline__ double DppMov(double x) {
  const int lo = DppMov<CTRL>(__double2loint(x));
  const int hi = DppMov<CTRL>(__double2hiint(x));
  return __hiloint2double(hi, lo);
}
#define EPA_QUAD_REDUCE(x, OP)            \
  do {                                    \
    auto y_ = DppMov<kDppSwap1>(x);       \
    x = OP(x, y_);                        \
    y_ = DppMov<kDppSwap2>(x);            \
    x = OP(x, y_);                        \
  } while (0)
#else
// single-lane host instantiation of the scalar vocabulary (hipcc's host pass, 1-lane tests)
#define EPA_QUAD_REDUCE(x, OP) ((void)0)
#endif
#define EPA_QUAD_ADD(a, b) ((a) + (b))
#define EPA_QUAD_MAX(a, b) ((a) > (b) ? (a) : (b))
#define EPA_QUAD_AND(a, b) ((a) & (b))
#define EPA_QUAD_OR(a, b) ((a) | (b))
EPA_HD float Sum4(float x) {
  EPA_QUAD_REDUCE(x, EPA_QUAD_ADD);
  return x;
}
EPA_HD double Sum4(double x) {
  EPA_QUAD_REDUCE(x, EPA_QUAD_ADD);
  return x;
}
EPA_HD float Max4(float x) {
  EPA_QUAD_REDUCE(x, EPA_QUAD_MAX);
  return x;
}
EPA_HD double Max4(double x) {
  EPA_QUAD_REDUCE(x, EPA_QUAD_MAX);
  return x;
}
EPA_HD bool All4(bool c) {
  int x = c ? 1 : 0;
  EPA_QUAD_REDUCE(x, EPA_QUAD_AND);
  return x != 0;
}
// number of bits set in the OR of the quad's masks
EPA_HD int MaskCount4(unsigned m) {
  int x = (int)m;
  EPA_QUAD_REDUCE(x, EPA_QUAD_OR);
  return __builtin_popcount((unsigned)x);
}
EPA_HD bool IsFinite(float x) { return x - x == 0.0f; }  // false for NaN and +-inf
EPA_HD bool IsFinite(double x) { return x - x == 0.0; }

}  // namespace mj
}  // namespace epa

#endif  // ENVPOOL_AMD_CSRC_MJ_QUAD_HIP_H_
