// TEST INFRASTRUCTURE (oracle/): a scalar that counts its floating-point operations -- bench.py's `roofline.useful` runs the oracle's assembly
// (set_entries!, residual evaluations, the data matrix of the IFT: the reference's formulas as restated in dojo_oracle.hpp) on it to say how many
// flops the reference's algorithm needs OUTSIDE its linear solves (those are counted as a block-sparse LU, oracle_math.hpp SparseLU).  Every + - * /
// counts one, so a multiply-add counts two like the device's FMA in the PMC figures; sqrt / sin / cos / atan / pow count one each.
#pragma once
#include <cmath>
#include <limits>
namespace orc {
struct Counted {
    double v;
    static long long& ops() { static thread_local long long n = 0; return n; }
    static int& paused() { static thread_local int p = 0; return p; }
    static void tick(int k = 1) { if (!paused()) ops() += k; }
    Counted() : v(0) {}
    Counted(double x) : v(x) {}
    Counted(float x) : v(x) {}
    Counted(int x) : v(x) {}
    Counted(long x) : v((double)x) {}
    Counted(long long x) : v((double)x) {}
    Counted(unsigned x) : v(x) {}
    Counted(unsigned long x) : v((double)x) {}
    Counted(long double x) : v((double)x) {}
    explicit operator double() const { return v; }
    explicit operator float() const { return (float)v; }
    explicit operator long double() const { return (long double)v; }
    explicit operator int() const { return (int)v; }
    explicit operator bool() const { return v != 0; }
    Counted operator-() const { return Counted(-v); }
    Counted& operator+=(Counted o) { tick(); v += o.v; return *this; }
    Counted& operator-=(Counted o) { tick(); v -= o.v; return *this; }
    Counted& operator*=(Counted o) { tick(); v *= o.v; return *this; }
    Counted& operator/=(Counted o) { tick(); v /= o.v; return *this; }
};
#define ORC_CNT_BIN(op) \
    inline Counted operator op(Counted a, Counted b) { Counted::tick(); return Counted(a.v op b.v); } \
    inline Counted operator op(Counted a, double b) { Counted::tick(); return Counted(a.v op b); } \
    inline Counted operator op(double a, Counted b) { Counted::tick(); return Counted(a op b.v); } \
    inline Counted operator op(Counted a, int b) { Counted::tick(); return Counted(a.v op b); } \
    inline Counted operator op(int a, Counted b) { Counted::tick(); return Counted(a op b.v); }
ORC_CNT_BIN(+) ORC_CNT_BIN(-) ORC_CNT_BIN(*) ORC_CNT_BIN(/)
#undef ORC_CNT_BIN
#define ORC_CNT_CMP(op) \
    inline bool operator op(Counted a, Counted b) { return a.v op b.v; } \
    inline bool operator op(Counted a, double b) { return a.v op b; } \
    inline bool operator op(double a, Counted b) { return a op b.v; } \
    inline bool operator op(Counted a, int b) { return a.v op b; } \
    inline bool operator op(int a, Counted b) { return a op b.v; }
ORC_CNT_CMP(<) ORC_CNT_CMP(>) ORC_CNT_CMP(<=) ORC_CNT_CMP(>=) ORC_CNT_CMP(==) ORC_CNT_CMP(!=)
#undef ORC_CNT_CMP
// counting stops inside the linear solves (RAII; a no-op for the other scalar types)
template <class T> struct OpPause { OpPause() {} };
template <> struct OpPause<Counted> { OpPause() { ++Counted::paused(); } ~OpPause() { --Counted::paused(); } };
}  // namespace orc
namespace std {
inline orc::Counted sqrt(orc::Counted a) { orc::Counted::tick(); return orc::Counted(std::sqrt(a.v)); }
inline orc::Counted fabs(orc::Counted a) { return orc::Counted(std::fabs(a.v)); }
inline orc::Counted abs(orc::Counted a) { return orc::Counted(std::fabs(a.v)); }
inline orc::Counted sin(orc::Counted a) { orc::Counted::tick(); return orc::Counted(std::sin(a.v)); }
inline orc::Counted cos(orc::Counted a) { orc::Counted::tick(); return orc::Counted(std::cos(a.v)); }
inline orc::Counted atan(orc::Counted a) { orc::Counted::tick(); return orc::Counted(std::atan(a.v)); }
inline orc::Counted pow(orc::Counted a, orc::Counted b) { orc::Counted::tick(); return orc::Counted(std::pow(a.v, b.v)); }
inline orc::Counted pow(orc::Counted a, double b) { orc::Counted::tick(); return orc::Counted(std::pow(a.v, b)); }
inline orc::Counted pow(orc::Counted a, int b) { orc::Counted::tick(); return orc::Counted(std::pow(a.v, b)); }
inline orc::Counted fmax(orc::Counted a, orc::Counted b) { return orc::Counted(std::fmax(a.v, b.v)); }
inline orc::Counted fmin(orc::Counted a, orc::Counted b) { return orc::Counted(std::fmin(a.v, b.v)); }
inline orc::Counted fmax(orc::Counted a, double b) { return orc::Counted(std::fmax(a.v, b)); }
inline orc::Counted fmin(orc::Counted a, double b) { return orc::Counted(std::fmin(a.v, b)); }
inline orc::Counted fmax(double a, orc::Counted b) { return orc::Counted(std::fmax(a, b.v)); }
inline orc::Counted fmin(double a, orc::Counted b) { return orc::Counted(std::fmin(a, b.v)); }
inline bool isfinite(orc::Counted a) { return std::isfinite(a.v); }
inline bool isnan(orc::Counted a) { return std::isnan(a.v); }
template <> struct numeric_limits<orc::Counted> : numeric_limits<double> {
    static orc::Counted infinity() { return orc::Counted(numeric_limits<double>::infinity()); }
    static orc::Counted epsilon() { return orc::Counted(numeric_limits<double>::epsilon()); }
    static orc::Counted quiet_NaN() { return orc::Counted(numeric_limits<double>::quiet_NaN()); }
    static orc::Counted max() { return orc::Counted(numeric_limits<double>::max()); }
    static orc::Counted min() { return orc::Counted(numeric_limits<double>::min()); }
    static orc::Counted lowest() { return orc::Counted(numeric_limits<double>::lowest()); }
};
}  // namespace std
