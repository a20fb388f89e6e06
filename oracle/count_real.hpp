// count_real.hpp — a counting scalar for the flop census of the CPU oracle (test infrastructure).
// oracle/Makefile builds amx_oracle.c once more as C++ with REAL = CountReal: every +, -, *, / (and sqrt, sin, cos, pow)
// the restated algorithm executes on a REAL adds one to a thread-local counter.  bench.py reads that count per
// env-step for its secondary FP32 figure (flops the algorithm needs / 157.3 TFLOP/s); it is never timed.
#pragma once
#include <math.h>
struct CountReal;
extern thread_local double amxo_flop_count;
struct CountReal {
  double v;
  CountReal() : v(0) {}
  CountReal(double x) : v(x) {}
  CountReal(float x) : v(x) {}
  CountReal(int x) : v(x) {}
  CountReal(long x) : v((double)x) {}
  CountReal(unsigned long x) : v((double)x) {}
  explicit operator double() const { return v; }
  explicit operator float() const { return (float)v; }
  explicit operator int() const { return (int)v; }
  explicit operator bool() const { return v != 0; }
  CountReal operator-() const { return CountReal(-v); }
  CountReal &operator+=(CountReal o) { amxo_flop_count += 1; v += o.v; return *this; }
  CountReal &operator-=(CountReal o) { amxo_flop_count += 1; v -= o.v; return *this; }
  CountReal &operator*=(CountReal o) { amxo_flop_count += 1; v *= o.v; return *this; }
  CountReal &operator/=(CountReal o) { amxo_flop_count += 1; v /= o.v; return *this; }
};
#define CR_BIN(op)                                                                                                  \
  inline CountReal operator op(CountReal a, CountReal b) { amxo_flop_count += 1; return CountReal(a.v op b.v); }    \
  inline CountReal operator op(CountReal a, double b) { amxo_flop_count += 1; return CountReal(a.v op b); }         \
  inline CountReal operator op(double a, CountReal b) { amxo_flop_count += 1; return CountReal(a op b.v); }         \
  inline CountReal operator op(CountReal a, int b) { amxo_flop_count += 1; return CountReal(a.v op b); }            \
  inline CountReal operator op(int a, CountReal b) { amxo_flop_count += 1; return CountReal(a op b.v); }
CR_BIN(+) CR_BIN(-) CR_BIN(*) CR_BIN(/)
#undef CR_BIN
#define CR_CMP(op)                                                           \
  inline bool operator op(CountReal a, CountReal b) { return a.v op b.v; }   \
  inline bool operator op(CountReal a, double b) { return a.v op b; }        \
  inline bool operator op(double a, CountReal b) { return a op b.v; }        \
  inline bool operator op(CountReal a, int b) { return a.v op b; }           \
  inline bool operator op(int a, CountReal b) { return a op b.v; }
CR_CMP(<) CR_CMP(>) CR_CMP(<=) CR_CMP(>=) CR_CMP(==) CR_CMP(!=)
#undef CR_CMP
