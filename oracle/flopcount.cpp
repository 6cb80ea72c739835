// flopcount.cpp -- TEST INFRASTRUCTURE: counts the floating-point operations of one fused env step by compiling the
// CPU oracle (shf_oracle.c, unchanged) with its real type replaced by a counting wrapper.  Used only by
// tools/count_flops.py to state F_alg (SURVEY.md 8d "restate with the real count") for bench.py's secondary roofline.
// Counting rule: add / sub / mul / div / sqrt = 1 flop, fma = 2; comparisons, negation, abs, floor/trunc/rint,
// conversions and min/max selections = 0.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

static thread_local unsigned long long g_flops = 0;

struct Counted {
  float v;
  Counted() : v(0.0f) {}
  template <class T> Counted(T x) : v((float)x) {}
  operator float() const { return v; }
  explicit operator double() const { return v; }
  explicit operator long() const { return (long)v; }
  explicit operator int() const { return (int)v; }
  explicit operator long long() const { return (long long)v; }
  explicit operator unsigned char() const { return (unsigned char)v; }
  Counted operator-() const { Counted r; r.v = -v; return r; }
  Counted& operator+=(Counted o) { v += o.v; g_flops++; return *this; }
  Counted& operator-=(Counted o) { v -= o.v; g_flops++; return *this; }
  Counted& operator*=(Counted o) { v *= o.v; g_flops++; return *this; }
  Counted& operator/=(Counted o) { v /= o.v; g_flops++; return *this; }
};
#define BIN(op)                                                                         \
  static inline Counted operator op(Counted a, Counted b) { Counted r; r.v = a.v op b.v; g_flops++; return r; }
BIN(+) BIN(-) BIN(*) BIN(/)
#define CMP(op) static inline bool operator op(Counted a, Counted b) { return a.v op b.v; }
CMP(<) CMP(>) CMP(<=) CMP(>=) CMP(==) CMP(!=)
static inline Counted c_fma(Counted a, Counted b, Counted c) { Counted r; r.v = fmaf(a.v, b.v, c.v); g_flops += 2; return r; }
static inline Counted c_sqrt(Counted a) { Counted r; r.v = sqrtf(a.v); g_flops++; return r; }
static inline Counted c_wrap(float x) { Counted r; r.v = x; return r; }

#define SHF_REAL_DOUBLE 0
#define SHF_FLOPCOUNT 1
typedef Counted R;
#define FMA(a, b, c) c_fma((a), (b), (c))
#define SQRT(x) c_sqrt(x)
#define RINT(x) c_wrap(rintf((float)(x)))
#define FABS(x) c_wrap(fabsf((float)(x)))
#define FLOOR(x) c_wrap(floorf((float)(x)))
#define TRUNC(x) c_wrap(truncf((float)(x)))
#define SUF(name) name##_cnt
#define SHF_ORACLE_CUSTOM_REAL 1
#define SHF_COUNT(n) (g_flops += (n))
extern "C" {
#include "shf_oracle.c"
unsigned long long shf_flopcount_read(int reset) { unsigned long long v = g_flops; if (reset) g_flops = 0; return v; }
}
