// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Exact FLOP counter of the CPU restatement ("the build's oracle must carry an exact FLOP counter").
// `make liboracle_flops.so` builds the SAME sources with the scalar of mat.hpp replaced by oracle::Counted: a double whose
// every +, -, *, /, sqrt and transcendental call adds one to a counter, attributed to the REGION of the hot path that is
// executing (scope guards FLOP_REGION(...) at the entry of the functions that restate SURVEY 8(a)'s rows).  What is counted is what
// the restatement -- hence, by construction, the reference's formulation on Eigen dense blocks -- executes: a dense product
// counts its structural zeros, like Eigen's GEMM does.  A fused multiply-add counts as one mul and one add.
// In every other build FLOP_REGION(...) is empty and nothing of this header exists.
#ifndef ORACLE_FLOPS_HPP_
#define ORACLE_FLOPS_HPP_

#ifdef ORACLE_COUNT_FLOPS
#include <cmath>
#include <type_traits>

namespace oracle {

enum FlopKind { F_ADD = 0, F_MUL, F_DIV, F_SQRT, F_TRANS, F_NKIND };
// regions = rows of SURVEY.md 8(a) as the kernels group them
enum FlopRegionId {
  R_OTHER = 0,        // anything outside the hot path (set-up, discretiser, KKT error)
  R_KINEMATICS,       // a4  Robot::updateKinematics + frame terms
  R_RNEA,             // a1/a3/a7  Robot::RNEA(+Impulse), setContactForces
  R_RNEA_DERIV,       // a2/a7  RNEADerivatives (+ impulse twin)
  R_BAUMGARTE,        // a5/a6  Baumgarte / impulse velocity / contact position residuals and derivatives
  R_MJTJINV,          // a8  computeMJtJinv / computeMinv
  R_LIE,              // a9  Lie-group operations
  R_COST_CONSTRAINTS, // a10/a16 + the multiplier terms of a12's linearize half (state equation, cost, IPM, l += dt [dID; dC]^T [beta; mu])
  R_CONDENSE,         // a12/a13  condenseContactDynamics / condenseImpulseDynamics (the products behind computeMJtJinv)
  R_SWITCH,           // a14  switching constraint (linearize + condense)
  R_UNCONDENSE,       // a11  UnconstrainedDynamics (fixed base)
  R_RICCATI_BWD,      // a17/a18  backward Riccati recursion (incl. the constrained step)
  R_RICCATI_FWD,      // a17/a18  forward recursion
  R_EXPAND,           // direction expansion: primal / dual of the condensed variables, slack / dual directions, step sizes
  R_INTEGRATE,        // updatePrimal / updateDual
  R_KKT_INVERSE,      // a21  SplitKKTMatrixInverter + coarse update
  R_CORRECTION,       // a21  serial / parallel backward and forward corrections
  R_NREGION
};
extern thread_local int flop_region;
extern unsigned long long flop_count[R_NREGION][F_NKIND];      // (the counting build runs single-threaded)
inline void flopAdd(int kind) { ++flop_count[flop_region][kind]; }

struct Counted {
  double v;
  Counted() : v(0.0) {}
  template <typename T, typename = typename std::enable_if<std::is_arithmetic<T>::value>::type>
  Counted(T x) : v((double)x) {}
  operator double() const { return v; }
  Counted operator-() const { return Counted(-v); }
  Counted& operator+=(Counted o) { flopAdd(F_ADD); v += o.v; return *this; }
  Counted& operator-=(Counted o) { flopAdd(F_ADD); v -= o.v; return *this; }
  Counted& operator*=(Counted o) { flopAdd(F_MUL); v *= o.v; return *this; }
  Counted& operator/=(Counted o) { flopAdd(F_DIV); v /= o.v; return *this; }
};
#define ORACLE_COUNTED_BINOP(op, kind)                                                                                                  \
  inline Counted operator op(Counted a, Counted b) { flopAdd(kind); return Counted(a.v op b.v); }                                       \
  template <typename T, typename = typename std::enable_if<std::is_arithmetic<T>::value>::type>                                         \
  inline Counted operator op(Counted a, T b) { flopAdd(kind); return Counted(a.v op (double)b); }                                       \
  template <typename T, typename = typename std::enable_if<std::is_arithmetic<T>::value>::type>                                         \
  inline Counted operator op(T a, Counted b) { flopAdd(kind); return Counted((double)a op b.v); }
ORACLE_COUNTED_BINOP(+, F_ADD)
ORACLE_COUNTED_BINOP(-, F_ADD)
ORACLE_COUNTED_BINOP(*, F_MUL)
ORACLE_COUNTED_BINOP(/, F_DIV)
#undef ORACLE_COUNTED_BINOP
#define ORACLE_COUNTED_CMP(op)                                                                                                          \
  inline bool operator op(Counted a, Counted b) { return a.v op b.v; }                                                                  \
  template <typename T, typename = typename std::enable_if<std::is_arithmetic<T>::value>::type>                                         \
  inline bool operator op(Counted a, T b) { return a.v op (double)b; }                                                                  \
  template <typename T, typename = typename std::enable_if<std::is_arithmetic<T>::value>::type>                                         \
  inline bool operator op(T a, Counted b) { return (double)a op b.v; }
ORACLE_COUNTED_CMP(<)
ORACLE_COUNTED_CMP(>)
ORACLE_COUNTED_CMP(<=)
ORACLE_COUNTED_CMP(>=)
ORACLE_COUNTED_CMP(==)
ORACLE_COUNTED_CMP(!=)
#undef ORACLE_COUNTED_CMP

struct FlopScope {
  int prev;
  explicit FlopScope(int r) : prev(flop_region) { flop_region = r; }
  ~FlopScope() { flop_region = prev; }
};

}  // namespace oracle

// the <cmath> calls of the restatement are written std::sqrt(x) ...: overloads for the counting scalar
namespace std {
inline oracle::Counted sqrt(oracle::Counted x) { oracle::flopAdd(oracle::F_SQRT); return oracle::Counted(std::sqrt(x.v)); }
inline oracle::Counted fabs(oracle::Counted x) { return oracle::Counted(std::fabs(x.v)); }
inline oracle::Counted abs(oracle::Counted x) { return oracle::Counted(std::fabs(x.v)); }
inline oracle::Counted sin(oracle::Counted x) { oracle::flopAdd(oracle::F_TRANS); return oracle::Counted(std::sin(x.v)); }
inline oracle::Counted cos(oracle::Counted x) { oracle::flopAdd(oracle::F_TRANS); return oracle::Counted(std::cos(x.v)); }
inline oracle::Counted tan(oracle::Counted x) { oracle::flopAdd(oracle::F_TRANS); return oracle::Counted(std::tan(x.v)); }
inline oracle::Counted acos(oracle::Counted x) { oracle::flopAdd(oracle::F_TRANS); return oracle::Counted(std::acos(x.v)); }
inline oracle::Counted asin(oracle::Counted x) { oracle::flopAdd(oracle::F_TRANS); return oracle::Counted(std::asin(x.v)); }
inline oracle::Counted atan2(oracle::Counted y, oracle::Counted x) { oracle::flopAdd(oracle::F_TRANS); return oracle::Counted(std::atan2(y.v, x.v)); }
inline oracle::Counted exp(oracle::Counted x) { oracle::flopAdd(oracle::F_TRANS); return oracle::Counted(std::exp(x.v)); }
inline oracle::Counted log(oracle::Counted x) { oracle::flopAdd(oracle::F_TRANS); return oracle::Counted(std::log(x.v)); }
inline oracle::Counted pow(oracle::Counted x, oracle::Counted y) { oracle::flopAdd(oracle::F_TRANS); return oracle::Counted(std::pow(x.v, y.v)); }
inline bool isnan(oracle::Counted x) { return std::isnan(x.v); }
inline bool isfinite(oracle::Counted x) { return std::isfinite(x.v); }
}  // namespace std

#define ORACLE_FLOP_CAT2(a, b) a##b
#define ORACLE_FLOP_CAT(a, b) ORACLE_FLOP_CAT2(a, b)
#define FLOP_REGION(r) oracle::FlopScope ORACLE_FLOP_CAT(flop_scope_, __LINE__)(oracle::r)
// inside a function that holds a FLOP_REGION guard: switch the region for the statements that follow (the guard restores the caller's)
#define FLOP_REGION_SET(r) do { oracle::flop_region = oracle::r; } while (0)
#else
#define FLOP_REGION(r) do { } while (0)
#define FLOP_REGION_SET(r) do { } while (0)
#endif

#endif  // ORACLE_FLOPS_HPP_
