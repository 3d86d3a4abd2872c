// Forward-mode automatic differentiation for user-supplied device models (user_model.h).
//
// The reference differentiates whatever Python callables it is given by finite differences (quattro_ilqr_tf.py:149-275:
// 2(n+m) dynamics evaluations and ~(n+m)^2 cost evaluations per step, eps = 1e-5 in fp64).  fp32 cannot take second
// differences at that step, so a user model written once as templates over its scalar type is differentiated exactly
// instead: Dual<float> carries one directional derivative (a column of [A | B]), Dual<Dual<float>> one entry of a Hessian
// together with both of its gradient entries.
#pragma once
#include <type_traits>

#include "models_device.h"

namespace qtad {

template <class T>
struct Dual {
  T v, d;   // value, derivative along the seeded direction
  __device__ __forceinline__ Dual() {}
  __device__ __forceinline__ Dual(const T& v_, const T& d_) : v(v_), d(d_) {}
  template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type>
  __device__ __forceinline__ Dual(S s) : v(T(s)), d(T(0.0f)) {}
};

template <class T>
struct is_dual : std::false_type {};
template <class T>
struct is_dual<Dual<T>> : std::true_type {};

// the plain number underneath any nesting (for comparisons and branches)
__device__ __forceinline__ float primal(float x) { return x; }
template <class T>
__device__ __forceinline__ float primal(const Dual<T>& x) { return primal(x.v); }

#define QT_AR template <class T, class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type>
template <class T> __device__ __forceinline__ Dual<T> operator+(const Dual<T>& a, const Dual<T>& b) { return {a.v + b.v, a.d + b.d}; }
template <class T> __device__ __forceinline__ Dual<T> operator-(const Dual<T>& a, const Dual<T>& b) { return {a.v - b.v, a.d - b.d}; }
template <class T> __device__ __forceinline__ Dual<T> operator*(const Dual<T>& a, const Dual<T>& b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
template <class T> __device__ __forceinline__ Dual<T> operator/(const Dual<T>& a, const Dual<T>& b) {
  const T q = a.v / b.v;
  return {q, (a.d - q * b.d) / b.v};
}
template <class T> __device__ __forceinline__ Dual<T> operator-(const Dual<T>& a) { return {-a.v, -a.d}; }
template <class T> __device__ __forceinline__ Dual<T> operator+(const Dual<T>& a) { return a; }
QT_AR __device__ __forceinline__ Dual<T> operator+(const Dual<T>& a, S s) { return {a.v + T(s), a.d}; }
QT_AR __device__ __forceinline__ Dual<T> operator+(S s, const Dual<T>& a) { return {T(s) + a.v, a.d}; }
QT_AR __device__ __forceinline__ Dual<T> operator-(const Dual<T>& a, S s) { return {a.v - T(s), a.d}; }
QT_AR __device__ __forceinline__ Dual<T> operator-(S s, const Dual<T>& a) { return {T(s) - a.v, -a.d}; }
QT_AR __device__ __forceinline__ Dual<T> operator*(const Dual<T>& a, S s) { return {a.v * T(s), a.d * T(s)}; }
QT_AR __device__ __forceinline__ Dual<T> operator*(S s, const Dual<T>& a) { return {T(s) * a.v, T(s) * a.d}; }
QT_AR __device__ __forceinline__ Dual<T> operator/(const Dual<T>& a, S s) { return {a.v / T(s), a.d / T(s)}; }
QT_AR __device__ __forceinline__ Dual<T> operator/(S s, const Dual<T>& a) { return Dual<T>(s) / a; }
template <class T> __device__ __forceinline__ Dual<T>& operator+=(Dual<T>& a, const Dual<T>& b) { a = a + b; return a; }
template <class T> __device__ __forceinline__ Dual<T>& operator-=(Dual<T>& a, const Dual<T>& b) { a = a - b; return a; }
template <class T> __device__ __forceinline__ Dual<T>& operator*=(Dual<T>& a, const Dual<T>& b) { a = a * b; return a; }
template <class T> __device__ __forceinline__ Dual<T>& operator/=(Dual<T>& a, const Dual<T>& b) { a = a / b; return a; }
QT_AR __device__ __forceinline__ Dual<T>& operator+=(Dual<T>& a, S s) { a = a + s; return a; }
QT_AR __device__ __forceinline__ Dual<T>& operator-=(Dual<T>& a, S s) { a = a - s; return a; }
QT_AR __device__ __forceinline__ Dual<T>& operator*=(Dual<T>& a, S s) { a = a * s; return a; }
QT_AR __device__ __forceinline__ Dual<T>& operator/=(Dual<T>& a, S s) { a = a / s; return a; }
// comparisons look at the value only (piecewise definitions differentiate the branch that is taken)
#define QT_CMP(op)                                                                                                       \
  template <class T> __device__ __forceinline__ bool operator op(const Dual<T>& a, const Dual<T>& b) { return primal(a) op primal(b); } \
  QT_AR __device__ __forceinline__ bool operator op(const Dual<T>& a, S s) { return primal(a) op (float)s; }             \
  QT_AR __device__ __forceinline__ bool operator op(S s, const Dual<T>& a) { return (float)s op primal(a); }
QT_CMP(<) QT_CMP(>) QT_CMP(<=) QT_CMP(>=)
#undef QT_CMP
#undef QT_AR

// elementary functions: float versions first (a model's rollout code and its differentiated code call the same names)
__device__ __forceinline__ float sin(float x) { float s, c; qt_sincos(x, &s, &c); return s; }
__device__ __forceinline__ float cos(float x) { float s, c; qt_sincos(x, &s, &c); return c; }
__device__ __forceinline__ void sincos(float x, float* s, float* c) { qt_sincos(x, s, c); }
__device__ __forceinline__ float tan(float x) { float s, c; qt_sincos(x, &s, &c); return s / c; }
__device__ __forceinline__ float exp(float x) { return ::expf(x); }
__device__ __forceinline__ float log(float x) { return ::logf(x); }
__device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
__device__ __forceinline__ float tanh(float x) { return ::tanhf(x); }
__device__ __forceinline__ float atan(float x) { return ::atanf(x); }
__device__ __forceinline__ float fabs(float x) { return ::fabsf(x); }
__device__ __forceinline__ float abs(float x) { return ::fabsf(x); }
__device__ __forceinline__ float pow(float x, float e) { return ::powf(x, e); }
__device__ __forceinline__ float fmax(float a, float b) { return ::fmaxf(a, b); }
__device__ __forceinline__ float fmin(float a, float b) { return ::fminf(a, b); }
__device__ __forceinline__ float square(float x) { return x * x; }
// log(1 + exp(beta z)) / beta, overflow-safe (the reference's barrier: quadrotor_mpc.py:74-100)
__device__ __forceinline__ float softplus(float z, float beta) { return qt_softplus(z, beta); }

template <class T> __device__ __forceinline__ void sincos(const Dual<T>& a, Dual<T>* s, Dual<T>* c) {
  T sv, cv;
  sincos(a.v, &sv, &cv);
  *s = Dual<T>(sv, cv * a.d);
  *c = Dual<T>(cv, -(sv * a.d));
}
template <class T> __device__ __forceinline__ Dual<T> sin(const Dual<T>& a) { Dual<T> s, c; sincos(a, &s, &c); return s; }
template <class T> __device__ __forceinline__ Dual<T> cos(const Dual<T>& a) { Dual<T> s, c; sincos(a, &s, &c); return c; }
template <class T> __device__ __forceinline__ Dual<T> tan(const Dual<T>& a) {
  const T t = tan(a.v);
  return {t, (t * t + 1.0f) * a.d};
}
template <class T> __device__ __forceinline__ Dual<T> exp(const Dual<T>& a) { const T e = exp(a.v); return {e, e * a.d}; }
template <class T> __device__ __forceinline__ Dual<T> log(const Dual<T>& a) { return {log(a.v), a.d / a.v}; }
template <class T> __device__ __forceinline__ Dual<T> sqrt(const Dual<T>& a) { const T r = sqrt(a.v); return {r, a.d / (r * 2.0f)}; }
template <class T> __device__ __forceinline__ Dual<T> tanh(const Dual<T>& a) { const T t = tanh(a.v); return {t, (1.0f - t * t) * a.d}; }
template <class T> __device__ __forceinline__ Dual<T> atan(const Dual<T>& a) { return {atan(a.v), a.d / (a.v * a.v + 1.0f)}; }
template <class T> __device__ __forceinline__ Dual<T> fabs(const Dual<T>& a) { return primal(a) < 0.0f ? -a : a; }
template <class T> __device__ __forceinline__ Dual<T> abs(const Dual<T>& a) { return fabs(a); }
template <class T> __device__ __forceinline__ Dual<T> pow(const Dual<T>& a, float e) {   // constant exponent
  return {pow(a.v, e), pow(a.v, e - 1.0f) * e * a.d};
}
template <class T> __device__ __forceinline__ Dual<T> fmax(const Dual<T>& a, const Dual<T>& b) { return primal(a) >= primal(b) ? a : b; }
template <class T> __device__ __forceinline__ Dual<T> fmin(const Dual<T>& a, const Dual<T>& b) { return primal(a) <= primal(b) ? a : b; }
template <class T> __device__ __forceinline__ Dual<T> square(const Dual<T>& a) { return a * a; }
template <class T> __device__ __forceinline__ Dual<T> softplus(const Dual<T>& z, float beta) {
  // d/dz softplus_beta(z) = sigmoid(beta z); written through exp / log so that every level of nesting differentiates it
  const Dual<T> bz = z * beta;
  const Dual<T> e = exp(-fabs(bz));
  return (fmax(bz, Dual<T>(0.0f)) + log(e + 1.0f)) / beta;
}

}  // namespace qtad
