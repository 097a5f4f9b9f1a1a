// TEST INFRASTRUCTURE: a declaration-only mock of the part of the Rcpp API that shim/phylomap_shim.cpp uses, so that the
// shim can be syntax- and type-checked (g++ -fsyntax-only) in an image without R.  Nothing here is linked or run, and it is
// not a stand-in for building the reference (the reference is not built here at all).
#pragma once
#include <cstddef>
#include <string>

typedef struct SEXPREC* SEXP;
#define RcppExport extern "C"
#define BEGIN_RCPP try {
#define END_RCPP } catch (...) { } return nullptr;

double unif_rand();

namespace Rcpp {
struct RNGScope { RNGScope(); ~RNGScope(); };
template <typename... A> [[noreturn]] void stop(const char* fmt, A...);

struct Proxy;
template <typename T> struct VecBase {
  VecBase(); VecBase(SEXP); VecBase(int); VecBase(const Proxy&);
  T* begin(); T* end(); const T* begin() const; const T* end() const;
  long size() const; T& operator[](long);
  operator SEXP() const;
};
struct NumericVector : VecBase<double> { using VecBase<double>::VecBase; };
struct IntegerVector : VecBase<int> { using VecBase<int>::VecBase; };
template <typename T> struct MatBase : VecBase<T> {
  MatBase(); MatBase(SEXP); MatBase(int, int); MatBase(const Proxy&);
  int nrow() const; int ncol() const;
};
struct NumericMatrix : MatBase<double> { using MatBase<double>::MatBase; };
struct IntegerMatrix : MatBase<int> { using MatBase<int>::MatBase; };

struct List;
struct Proxy { operator SEXP() const; template <typename T> operator T() const; template <typename T> Proxy& operator=(const T&); };
struct NamedArg { template <typename T> NamedArg operator=(const T&) const; };
NamedArg Named(const char*);
struct List {
  List(); List(SEXP); List(const Proxy&);
  Proxy operator[](const char*) const; Proxy operator[](int) const;
  List(int);
  long size() const; bool containsElementNamed(const char*) const;
  template <typename... A> static List create(A...);
  operator SEXP() const;
};
template <typename T> T as(SEXP);
template <typename T> T as(const Proxy&);
struct Function { template <typename... A> SEXP operator()(A...) const; };
struct Environment { Environment(const char*); Function operator[](const char*) const; };
}  // namespace Rcpp
