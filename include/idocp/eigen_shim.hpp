// Minimal Eigen stand-in for the facade when Eigen3 is not installed.
//
// The reference's public API passes Eigen::VectorXd / Eigen::MatrixXd
// (e.g. UnOCPSolver::updateSolution, include/idocp/unocp/unocp_solver.hpp:84).
// If <Eigen/Core> is available it is used unchanged; otherwise this header
// provides the tiny subset of the two types that the drivers touch: size(),
// data(), operator[] / operator(), Zero(), Constant(), setZero(), resize().
#ifndef IDOCP_EIGEN_SHIM_HPP_
#define IDOCP_EIGEN_SHIM_HPP_

#if defined(__has_include) && !defined(IDOCP_EIGEN_IS_THE_SHIM)      // (idocp/compat/Eigen/Core: a forwarder to this file, for drivers that include "Eigen/Core")
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define IDOCP_HAVE_EIGEN 1
#endif
#endif

#ifndef IDOCP_HAVE_EIGEN
#include <cassert>
#include <ostream>
#include <vector>

namespace Eigen {

// `v << a, b, c;` of Eigen's CommaInitializer, for the examples' reference vectors
template <typename V>
class CommaInit {
 public:
  CommaInit(V& v, double first) : v_(v), i_(0) { put(first); }
  CommaInit& operator,(double x) { put(x); return *this; }
 private:
  V& v_;
  int i_;
  void put(double x) { assert(i_ < v_.size()); v_[i_++] = x; }
};

class VectorXd {
 public:
  VectorXd() {}
  explicit VectorXd(int n) : d_(n, 0.0) {}
  static VectorXd Zero(int n) { return VectorXd(n); }
  static VectorXd Constant(int n, double v) { VectorXd x(n); for (auto& e : x.d_) e = v; return x; }
  int size() const { return (int)d_.size(); }
  void resize(int n) { d_.assign(n, 0.0); }
  void setZero() { for (auto& e : d_) e = 0.0; }
  double* data() { return d_.data(); }
  const double* data() const { return d_.data(); }
  double& operator[](int i) { assert(i >= 0 && i < size()); return d_[i]; }
  double operator[](int i) const { assert(i >= 0 && i < size()); return d_[i]; }
  double& operator()(int i) { return (*this)[i]; }
  double operator()(int i) const { return (*this)[i]; }
  double& coeffRef(int i) { return (*this)[i]; }
  double coeff(int i) const { return (*this)[i]; }
  CommaInit<VectorXd> operator<<(double first) { return CommaInit<VectorXd>(*this, first); }
  VectorXd operator-(const VectorXd& o) const { assert(size() == o.size()); VectorXd r(size()); for (int i = 0; i < size(); ++i) r[i] = d_[i] - o.d_[i]; return r; }
  VectorXd operator+(const VectorXd& o) const { assert(size() == o.size()); VectorXd r(size()); for (int i = 0; i < size(); ++i) r[i] = d_[i] + o.d_[i]; return r; }
 private:
  std::vector<double> d_;
};
inline std::ostream& operator<<(std::ostream& os, const VectorXd& v) { for (int i = 0; i < v.size(); ++i) os << (i ? " " : "") << v[i]; return os; }

class Vector3d {
 public:
  Vector3d() : d_{0.0, 0.0, 0.0} {}
  Vector3d(double x, double y, double z) : d_{x, y, z} {}
  static Vector3d Zero() { return Vector3d(); }
  static Vector3d Constant(double v) { return Vector3d(v, v, v); }
  int size() const { return 3; }
  double* data() { return d_; }
  const double* data() const { return d_; }
  double& operator[](int i) { assert(i >= 0 && i < 3); return d_[i]; }
  double operator[](int i) const { assert(i >= 0 && i < 3); return d_[i]; }
  double& operator()(int i) { return (*this)[i]; }
  double operator()(int i) const { return (*this)[i]; }
  double& coeffRef(int i) { return (*this)[i]; }
  double coeff(int i) const { return (*this)[i]; }
  CommaInit<Vector3d> operator<<(double first) { return CommaInit<Vector3d>(*this, first); }
  double squaredNorm() const { return d_[0] * d_[0] + d_[1] * d_[1] + d_[2] * d_[2]; }
  // DenseBase::isApprox with the default precision of double: |a - b|^2 <= 1e-24 min(|a|^2, |b|^2)
  bool isApprox(const Vector3d& o, const double prec = 1e-12) const {
    const Vector3d e(d_[0] - o.d_[0], d_[1] - o.d_[1], d_[2] - o.d_[2]);
    const double na = squaredNorm(), nb = o.squaredNorm();
    return e.squaredNorm() <= prec * prec * (na < nb ? na : nb);
  }
 private:
  double d_[3];
};

class Matrix3d {   // column-major storage; `m << ...` fills row by row like Eigen's comma initializer
 public:
  Matrix3d() : d_{0, 0, 0, 0, 0, 0, 0, 0, 0} {}
  static Matrix3d Zero() { return Matrix3d(); }
  static Matrix3d Identity() { Matrix3d m; m(0, 0) = m(1, 1) = m(2, 2) = 1.0; return m; }
  int rows() const { return 3; }
  int cols() const { return 3; }
  double* data() { return d_; }
  const double* data() const { return d_; }
  double& operator()(int i, int j) { assert(i >= 0 && i < 3 && j >= 0 && j < 3); return d_[3 * j + i]; }
  double operator()(int i, int j) const { assert(i >= 0 && i < 3 && j >= 0 && j < 3); return d_[3 * j + i]; }
  class RowFill {
   public:
    RowFill(Matrix3d& m, double first) : m_(m), i_(0) { put(first); }
    RowFill& operator,(double x) { put(x); return *this; }
   private:
    Matrix3d& m_;
    int i_;
    void put(double x) { assert(i_ < 9); m_(i_ / 3, i_ % 3) = x; ++i_; }
  };
  RowFill operator<<(double first) { return RowFill(*this, first); }
 private:
  double d_[9];
};

class MatrixXd {   // column-major
 public:
  MatrixXd() : r_(0), c_(0) {}
  MatrixXd(int r, int c) : r_(r), c_(c), d_((size_t)r * c, 0.0) {}
  static MatrixXd Zero(int r, int c) { return MatrixXd(r, c); }
  int rows() const { return r_; }
  int cols() const { return c_; }
  void resize(int r, int c) { r_ = r; c_ = c; d_.assign((size_t)r * c, 0.0); }
  void setZero() { for (auto& e : d_) e = 0.0; }
  double* data() { return d_.data(); }
  const double* data() const { return d_.data(); }
  double& operator()(int i, int j) { assert(i >= 0 && i < r_ && j >= 0 && j < c_); return d_[(size_t)j * r_ + i]; }
  double operator()(int i, int j) const { assert(i >= 0 && i < r_ && j >= 0 && j < c_); return d_[(size_t)j * r_ + i]; }
 private:
  int r_, c_;
  std::vector<double> d_;
};

}  // namespace Eigen
#endif  // !IDOCP_HAVE_EIGEN
#endif  // IDOCP_EIGEN_SHIM_HPP_
