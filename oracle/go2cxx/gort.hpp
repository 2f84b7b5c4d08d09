// gort.hpp -- the run-time half of oracle/go2cxx: Go semantics as C++20 templates.
//
// TEST INFRASTRUCTURE (oracle/): the C++ that go2cxx.py emits from Go source is purely syntax-directed; everything Go's
// type system and run time decide lives here, so that the C++ compiler does the type checking the translator does not:
//   * Num<T>: sized integers / floats with Go's rules -- no implicit promotion or mixing (uint8 op uint8 is uint8,
//     wrap-around on overflow, float32 op float32 is ONE binary32 rounding; build with -ffp-contract=off), shifts by
//     >= width give 0 / sign fill, integer division by zero panics, conversions are explicit;
//   * UInt / UFloat: untyped constants (they take the other operand's type; `:=` gives them int / float64);
//   * Slice<T> (ptr, len, cap sharing a backing array; index and slice expressions bounds-checked, a failure aborts
//     like a Go panic), Array<T,N> (value semantics), String (immutable bytes; range decodes UTF-8), Map<K,V>
//     (reference semantics; iteration order is unspecified in Go -- here: key order), Chan<T> (unbuffered rendezvous
//     or buffered), goroutines (threads), sync.WaitGroup / Once / Mutex;
//   * the handful of standard-library functions Go programs of this kind call (math, strconv, fmt verbs, bytes,
//     strings, encoding/binary, log), restated from the Go documentation.
// Nothing in this file knows the program being translated.  Memory is never freed (Go is garbage collected; the
// processes that use this are short-lived checkers).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace go {

[[noreturn]] inline void panic_msg(const std::string& s) {
    std::fprintf(stderr, "panic: %s\n", s.c_str());
    std::fflush(stderr);
    std::abort();
}

// ---------------------------------------------------------------------------------------------------- numbers
struct UFloat;
struct UInt {  // untyped integer (and rune) constant
    __int128 v;
    constexpr UInt(long long x = 0) : v(x) {}
    constexpr explicit UInt(__int128 x, int) : v(x) {}
    static constexpr UInt big(__int128 x) { return UInt(x, 0); }
    friend constexpr UInt operator+(UInt a, UInt b) { return big(a.v + b.v); }
    friend constexpr UInt operator-(UInt a, UInt b) { return big(a.v - b.v); }
    friend constexpr UInt operator*(UInt a, UInt b) { return big(a.v * b.v); }
    friend constexpr UInt operator/(UInt a, UInt b) { return big(a.v / b.v); }
    friend constexpr UInt operator%(UInt a, UInt b) { return big(a.v % b.v); }
    friend constexpr UInt operator&(UInt a, UInt b) { return big(a.v & b.v); }
    friend constexpr UInt operator|(UInt a, UInt b) { return big(a.v | b.v); }
    friend constexpr UInt operator^(UInt a, UInt b) { return big(a.v ^ b.v); }
    friend constexpr UInt operator<<(UInt a, UInt b) { return big(a.v << (int)b.v); }
    friend constexpr UInt operator>>(UInt a, UInt b) { return big(a.v >> (int)b.v); }
    friend constexpr bool operator==(UInt a, UInt b) { return a.v == b.v; }
    friend constexpr bool operator!=(UInt a, UInt b) { return a.v != b.v; }
    friend constexpr bool operator<(UInt a, UInt b) { return a.v < b.v; }
    friend constexpr bool operator<=(UInt a, UInt b) { return a.v <= b.v; }
    friend constexpr bool operator>(UInt a, UInt b) { return a.v > b.v; }
    friend constexpr bool operator>=(UInt a, UInt b) { return a.v >= b.v; }
    constexpr UInt operator-() const { return big(-v); }
    constexpr UInt operator+() const { return *this; }
};
constexpr UInt bitnot(UInt a) { return UInt::big(~a.v); }
constexpr UInt andnot(UInt a, UInt b) { return UInt::big(a.v & ~b.v); }

struct UFloat {  // untyped floating-point constant (held in double: exact for every literal a double can hold)
    double v;
    constexpr UFloat(double x = 0) : v(x) {}
    constexpr UFloat(UInt x) : v((double)x.v) {}
    friend constexpr UFloat operator+(UFloat a, UFloat b) { return a.v + b.v; }
    friend constexpr UFloat operator-(UFloat a, UFloat b) { return a.v - b.v; }
    friend constexpr UFloat operator*(UFloat a, UFloat b) { return a.v * b.v; }
    friend constexpr UFloat operator/(UFloat a, UFloat b) { return a.v / b.v; }
    friend constexpr bool operator==(UFloat a, UFloat b) { return a.v == b.v; }
    friend constexpr bool operator!=(UFloat a, UFloat b) { return a.v != b.v; }
    friend constexpr bool operator<(UFloat a, UFloat b) { return a.v < b.v; }
    friend constexpr bool operator<=(UFloat a, UFloat b) { return a.v <= b.v; }
    friend constexpr bool operator>(UFloat a, UFloat b) { return a.v > b.v; }
    friend constexpr bool operator>=(UFloat a, UFloat b) { return a.v >= b.v; }
    constexpr UFloat operator-() const { return -v; }
    constexpr UFloat operator+() const { return *this; }
};

template <class T> struct Num;
template <class X> struct is_num : std::false_type {};
template <class T> struct is_num<Num<T>> : std::true_type {};
template <class X> constexpr bool is_int_like = false;
template <class T> constexpr bool is_int_like<Num<T>> = std::is_integral_v<T>;
template <> inline constexpr bool is_int_like<UInt> = true;

template <class X> constexpr long long to_i64(X x) {
    if constexpr (std::is_same_v<X, UInt>) return (long long)x.v;
    else if constexpr (is_num<X>::value) return (long long)x.v;
    else return (long long)x;
}
template <class X> constexpr unsigned long long shift_count(X x) {
    static_assert(is_int_like<X>, "shift count must be an integer");
    if constexpr (std::is_same_v<X, UInt>) {
        if (x.v < 0) panic_msg("negative shift amount");
        return (unsigned long long)x.v;
    } else {
        if constexpr (std::is_signed_v<decltype(x.v)>)
            if (x.v < 0) panic_msg("negative shift amount");
        return (unsigned long long)x.v;
    }
}

template <class T> struct Num {
    T v;
    using raw = T;
    static constexpr bool is_float = std::is_floating_point_v<T>;
    using U = std::conditional_t<is_float, T, std::make_unsigned_t<std::conditional_t<is_float, int, T>>>;
    constexpr Num() : v(0) {}
    constexpr Num(T x) : v(x) {}
    constexpr Num(UInt c) : v((T)c.v) {}  // an untyped integer constant takes the operand's type
    template <class F = T, class = std::enable_if_t<std::is_floating_point_v<F>>> constexpr Num(UFloat c) : v((T)c.v) {}
    // explicit numeric conversion T(x): integers wrap, float -> integer truncates toward zero
    template <class S, class = std::enable_if_t<!std::is_same_v<S, T>>> constexpr explicit Num(Num<S> o) : v((T)o.v) {}

#define GO_ARITH(op)                                                                                                 \
    friend constexpr Num operator op(Num a, Num b) {                                                                  \
        if constexpr (is_float) return Num((T)(a.v op b.v));                                                          \
        else return Num((T)((U)a.v op (U)b.v));                                                                       \
    }
    GO_ARITH(+) GO_ARITH(-) GO_ARITH(*)
#undef GO_ARITH
    friend constexpr Num operator/(Num a, Num b) {
        if constexpr (is_float) return Num((T)(a.v / b.v));
        else {
            if (b.v == 0) panic_msg("runtime error: integer divide by zero");
            if constexpr (std::is_signed_v<T>)
                if (b.v == (T)-1) return Num((T)((U)0 - (U)a.v));
            return Num((T)(a.v / b.v));
        }
    }
    friend constexpr Num operator%(Num a, Num b) {
        static_assert(!is_float, "operator % on a float");
        if (b.v == 0) panic_msg("runtime error: integer divide by zero");
        if constexpr (std::is_signed_v<T>)
            if (b.v == (T)-1) return Num((T)0);
        return Num((T)(a.v % b.v));
    }
#define GO_BIT(op)                                                                                                   \
    friend constexpr Num operator op(Num a, Num b) {                                                                  \
        static_assert(!is_float, "bit operator on a float");                                                         \
        return Num((T)(a.v op b.v));                                                                                  \
    }
    GO_BIT(&) GO_BIT(|) GO_BIT(^)
#undef GO_BIT
    template <class S, class = std::enable_if_t<is_int_like<S>>> friend constexpr Num operator<<(Num a, S s) {
        static_assert(!is_float, "shift of a float");
        unsigned long long n = shift_count(s);
        if (n >= sizeof(T) * 8) return Num((T)0);
        return Num((T)((U)a.v << n));
    }
    template <class S, class = std::enable_if_t<is_int_like<S>>> friend constexpr Num operator>>(Num a, S s) {
        static_assert(!is_float, "shift of a float");
        unsigned long long n = shift_count(s);
        if (n >= sizeof(T) * 8) return Num((T)(a.v < 0 ? -1 : 0));
        return Num((T)(a.v >> n));
    }
#define GO_CMP(op) friend constexpr bool operator op(Num a, Num b) { return a.v op b.v; }
    GO_CMP(==) GO_CMP(!=) GO_CMP(<) GO_CMP(<=) GO_CMP(>) GO_CMP(>=)
#undef GO_CMP
    constexpr Num operator-() const {
        if constexpr (is_float) return Num(-v);
        else return Num((T)((U)0 - (U)v));
    }
    constexpr Num operator+() const { return *this; }
#define GO_OPASSIGN(op)                                                                                              \
    template <class X> constexpr Num& operator op##=(X x) {                                                           \
        *this = *this op x;                                                                                           \
        return *this;                                                                                                 \
    }
    GO_OPASSIGN(+) GO_OPASSIGN(-) GO_OPASSIGN(*) GO_OPASSIGN(/) GO_OPASSIGN(%) GO_OPASSIGN(&) GO_OPASSIGN(|)
    GO_OPASSIGN(^) GO_OPASSIGN(<<) GO_OPASSIGN(>>)
#undef GO_OPASSIGN
};
// a non-constant shift of an untyped constant: the constant "takes the type it would have without the shift";
// every such site in ordinary code is an int context, and anything else fails to compile rather than misbehave
template <class T, class = std::enable_if_t<std::is_integral_v<T>>> constexpr Num<long long> operator<<(UInt a, Num<T> s) {
    return Num<long long>((long long)a.v) << s;
}
template <class T> constexpr Num<T> bitnot(Num<T> a) { return Num<T>((T)~a.v); }
template <class T> constexpr Num<T> andnot(Num<T> a, Num<T> b) { return Num<T>((T)(a.v & ~b.v)); }
template <class T> constexpr Num<T> andnot(Num<T> a, UInt b) { return andnot(a, Num<T>(b)); }
template <class T> constexpr Num<T> andnot(UInt a, Num<T> b) { return andnot(Num<T>(a), b); }
template <class A, class B> constexpr void andnot_assign(A& a, B b) { a = andnot(a, b); }

using int_ = Num<long long>;
using int8 = Num<int8_t>;
using int16 = Num<int16_t>;
using int32 = Num<int32_t>;
using int64 = Num<long long>;
using uint = Num<unsigned long long>;
using uint8 = Num<uint8_t>;
using uint16 = Num<uint16_t>;
using uint32 = Num<uint32_t>;
using uint64 = Num<unsigned long long>;
using uintptr = Num<unsigned long long>;
using byte = uint8;
using rune = int32;
using float32 = Num<float>;
using float64 = Num<double>;
using bool_ = bool;
static_assert(sizeof(float32) == 4 && sizeof(uint8) == 1, "Num<T> must be layout-compatible with T");

// `x := c`, `var x = c`: an untyped constant takes its default type; everything else keeps its own
constexpr int_ def(UInt c) { return int_((long long)c.v); }
constexpr float64 def(UFloat c) { return float64(c.v); }
template <class T> constexpr T def(T x) { return x; }
template <class T> constexpr void inc(T& x) { x = x + UInt(1); }
template <class T> constexpr void dec(T& x) { x = x - UInt(1); }

// nil
struct Nil {};
inline constexpr Nil nil{};

// selector helper: Go's `x.f` dereferences pointers automatically
template <class T> constexpr T& deref(T& x) { return x; }
template <class T> constexpr const T& deref(const T& x) { return x; }
template <class T> constexpr T& deref(T* x) {
    if (!x) panic_msg("runtime error: invalid memory address or nil pointer dereference");
    return *x;
}
template <class T> constexpr T& star(T* x) { return deref(x); }
template <class T> constexpr T* addr(T& x) { return &x; }

// ---------------------------------------------------------------------------------------------------- slices
template <class X> long long index_of(X i, long long limit, const char* what) {
    long long k = to_i64(i);
    if (k < 0 || k >= limit) panic_msg(std::string("runtime error: index out of range [") + std::to_string(k) + "] with " + what + " " + std::to_string(limit));
    return k;
}

struct NoBound {};
inline constexpr NoBound nobound{};

template <class T> struct Slice {
    T* ptr = nullptr;
    long long len = 0, cap = 0;
    Slice() = default;
    Slice(Nil) {}
    Slice(T* p, long long l, long long c) : ptr(p), len(l), cap(c) {}
    // a defined type whose underlying type is a slice converts back and forth
    template <class D, class = std::enable_if_t<std::is_base_of_v<Slice<T>, D> && !std::is_same_v<D, Slice<T>>>>
    Slice(const D& d) : Slice(static_cast<const Slice<T>&>(d)) {}
    static Slice make(long long l, long long c) {
        if (l < 0 || c < l) panic_msg("runtime error: makeslice: len out of range");
        T* p = c ? new T[c]() : nullptr;
        return Slice(p, l, c);
    }
    static Slice of(std::initializer_list<T> il) {
        Slice s = make((long long)il.size(), (long long)il.size());
        std::copy(il.begin(), il.end(), s.ptr);
        return s;
    }
    template <class I> T& operator[](I i) const { return ptr[index_of(i, len, "length")]; }
    friend bool operator==(const Slice& s, Nil) { return s.ptr == nullptr; }
    friend bool operator!=(const Slice& s, Nil) { return s.ptr != nullptr; }
};
template <class T> struct is_slice : std::false_type {};
template <class T> struct is_slice<Slice<T>> : std::true_type {};

template <class T, long long N> struct Array {
    using go_array_elem = T;
    T a[N > 0 ? N : 1]{};
    template <class I> T& operator[](I i) { return a[index_of(i, N, "length")]; }
    template <class I> const T& operator[](I i) const { return a[index_of(i, N, "length")]; }
    friend bool operator==(const Array& x, const Array& y) { return std::equal(x.a, x.a + N, y.a); }
    friend bool operator!=(const Array& x, const Array& y) { return !(x == y); }
};

template <class L, class H> std::pair<long long, long long> bounds(L lo, H hi, long long cap, long long len) {
    long long l = 0, h = len;
    if constexpr (!std::is_same_v<L, NoBound>) l = to_i64(lo);
    if constexpr (!std::is_same_v<H, NoBound>) h = to_i64(hi);
    if (h < 0 || h > cap) panic_msg("runtime error: slice bounds out of range [:" + std::to_string(h) + "] with capacity " + std::to_string(cap));
    if (l < 0 || l > h) panic_msg("runtime error: slice bounds out of range [" + std::to_string(l) + ":" + std::to_string(h) + "]");
    return {l, h};
}
template <class T, class L, class H> Slice<T> slice(const Slice<T>& s, L lo, H hi) {
    auto [l, h] = bounds(lo, hi, s.cap, s.len);
    return Slice<T>(s.ptr + l, h - l, s.cap - l);
}
template <class T, class L, class H, class M> Slice<T> slice(const Slice<T>& s, L lo, H hi, M mx) {
    long long m = to_i64(mx);
    if (m < 0 || m > s.cap) panic_msg("runtime error: slice bounds out of range [::" + std::to_string(m) + "]");
    auto [l, h] = bounds(lo, hi, m, s.len);
    return Slice<T>(s.ptr + l, h - l, m - l);
}
template <class T, long long N, class L, class H> Slice<T> slice(Array<T, N>& a, L lo, H hi) {  // the array must be addressable
    auto [l, h] = bounds(lo, hi, N, N);
    return Slice<T>(a.a + l, h - l, N - l);
}
template <class T, long long N, class L, class H> Slice<T> slice(Array<T, N>* a, L lo, H hi) { return slice(deref(a), lo, hi); }

template <class T> int_ len(const Slice<T>& s) { return int_(s.len); }
template <class T> int_ cap(const Slice<T>& s) { return int_(s.cap); }
template <class T, long long N> constexpr int_ len(const Array<T, N>&) { return int_(N); }
template <class T, long long N> constexpr int_ cap(const Array<T, N>&) { return int_(N); }

template <class T> Slice<T> grow(Slice<T> s, long long need) {
    if (s.len + need <= s.cap) return s;
    long long nc = s.cap < 256 ? std::max<long long>(2 * s.cap, 4) : s.cap + s.cap / 4 + 192;
    nc = std::max(nc, s.len + need);
    Slice<T> r = Slice<T>::make(s.len, nc);
    std::copy(s.ptr, s.ptr + s.len, r.ptr);
    return r;
}
template <class S, class... A> auto append(const S& s0, A... a);

// ---------------------------------------------------------------------------------------------------- strings
struct String {
    std::shared_ptr<const std::string> owner;  // keeps substrings alive; Go strings are immutable
    const char* ptr = "";
    long long n = 0;
    String() = default;
    String(const char* lit) : owner(std::make_shared<const std::string>(lit)) { ptr = owner->data(); n = (long long)owner->size(); }
    String(const char* p, size_t len) : owner(std::make_shared<const std::string>(p, len)) { ptr = owner->data(); n = (long long)len; }
    String(const std::string& s) : owner(std::make_shared<const std::string>(s)) { ptr = owner->data(); n = (long long)s.size(); }
    explicit String(const Slice<byte>& b) : String(std::string((const char*)b.ptr, (size_t)b.len)) {}
    template <class D, class = std::enable_if_t<std::is_base_of_v<Slice<byte>, D> && !std::is_same_v<D, Slice<byte>>>>
    explicit String(const D& d) : String(static_cast<const Slice<byte>&>(d)) {}
    template <class T, class = std::enable_if_t<std::is_integral_v<T>>> explicit String(Num<T> r) : String(utf8((long long)r.v)) {}
    explicit String(UInt r) : String(utf8((long long)r.v)) {}
    static std::string utf8(long long r) {
        std::string o;
        if (r < 0 || r > 0x10FFFF || (r >= 0xD800 && r < 0xE000)) r = 0xFFFD;
        if (r < 0x80) o += (char)r;
        else if (r < 0x800) { o += (char)(0xC0 | (r >> 6)); o += (char)(0x80 | (r & 63)); }
        else if (r < 0x10000) { o += (char)(0xE0 | (r >> 12)); o += (char)(0x80 | ((r >> 6) & 63)); o += (char)(0x80 | (r & 63)); }
        else { o += (char)(0xF0 | (r >> 18)); o += (char)(0x80 | ((r >> 12) & 63)); o += (char)(0x80 | ((r >> 6) & 63)); o += (char)(0x80 | (r & 63)); }
        return o;
    }
    std::string str() const { return std::string(ptr, (size_t)n); }
    template <class I> byte operator[](I i) const { return byte((uint8_t)ptr[index_of(i, n, "length")]); }
    friend String operator+(const String& a, const String& b) { return String(a.str() + b.str()); }
    String& operator+=(const String& b) { *this = *this + b; return *this; }
    int cmp(const String& o) const {
        int c = std::memcmp(ptr, o.ptr, (size_t)std::min(n, o.n));
        return c ? c : (n < o.n ? -1 : n > o.n ? 1 : 0);
    }
    friend bool operator==(const String& a, const String& b) { return a.cmp(b) == 0; }
    friend bool operator!=(const String& a, const String& b) { return a.cmp(b) != 0; }
    friend bool operator<(const String& a, const String& b) { return a.cmp(b) < 0; }
    friend bool operator<=(const String& a, const String& b) { return a.cmp(b) <= 0; }
    friend bool operator>(const String& a, const String& b) { return a.cmp(b) > 0; }
    friend bool operator>=(const String& a, const String& b) { return a.cmp(b) >= 0; }
};
using string = String;
inline int_ len(const String& s) { return int_(s.n); }
template <class L, class H> String slice(const String& s, L lo, H hi) {
    auto [l, h] = bounds(lo, hi, s.n, s.n);
    String r;
    r.owner = s.owner;
    r.ptr = s.ptr + l;
    r.n = h - l;
    return r;
}
// []byte(s), []rune not supported
template <class T> Slice<T> to_slice(const String& s);
template <> inline Slice<byte> to_slice<byte>(const String& s) {
    Slice<byte> r = Slice<byte>::make(s.n, s.n);
    std::memcpy((void*)r.ptr, s.ptr, (size_t)s.n);
    return r;
}

// conversion T(x) for the cases C++'s functional cast cannot spell: go::conv<T>(x)
template <class T, class X> T conv(const X& x) {
    if constexpr (is_slice<T>::value && std::is_same_v<X, String>) return to_slice<typename std::remove_pointer_t<decltype(T().ptr)>>(x);
    else return T(x);
}

// ---------------------------------------------------------------------------------------------------- maps
template <class K> struct KeyLess {
    bool operator()(const K& a, const K& b) const {
        if constexpr (is_num<K>::value) return a.v < b.v;
        else return a < b;
    }
};
template <class K, class V> struct Map;
template <class K, class V> struct MapRef {  // m[k] as an expression: readable (zero value when absent) and assignable
    const Map<K, V>* m;
    K k;
    operator V() const;
    V get() const { return (V)(*this); }
    const MapRef& operator=(const V& v) const;
    template <class X> const MapRef& operator+=(X x) const { return *this = get() + x; }
    template <class X> const MapRef& operator-=(X x) const { return *this = get() - x; }
    template <class X> const MapRef& operator|=(X x) const { return *this = get() | x; }
};
template <class K, class V> struct Map {
    std::shared_ptr<std::map<K, V, KeyLess<K>>> p;
    Map() = default;
    Map(Nil) {}
    static Map make() {
        Map m;
        m.p = std::make_shared<std::map<K, V, KeyLess<K>>>();
        return m;
    }
    static Map of(std::initializer_list<std::pair<K, V>> il) {
        Map m = make();
        for (auto& kv : il) (*m.p)[kv.first] = kv.second;
        return m;
    }
    template <class X> MapRef<K, V> operator[](const X& k) const { return MapRef<K, V>{this, K(k)}; }
    friend bool operator==(const Map& m, Nil) { return !m.p; }
    friend bool operator!=(const Map& m, Nil) { return !!m.p; }
};
template <class K, class V> MapRef<K, V>::operator V() const {
    if (!m->p) return V{};
    auto it = m->p->find(k);
    return it == m->p->end() ? V{} : it->second;
}
template <class K, class V> const MapRef<K, V>& MapRef<K, V>::operator=(const V& v) const {
    if (!m->p) panic_msg("assignment to entry in nil map");
    (*m->p)[k] = v;
    return *this;
}
template <class K, class V> int_ len(const Map<K, V>& m) { return int_(m.p ? (long long)m.p->size() : 0); }
template <class K, class V, class X> void delete_(const Map<K, V>& m, const X& k) {
    if (m.p) m.p->erase(K(k));
}
template <class K, class V> std::tuple<V, bool> lookup(const MapRef<K, V>& r) {  // v, ok := m[k]
    if (!r.m->p) return {V{}, false};
    auto it = r.m->p->find(r.k);
    if (it == r.m->p->end()) return {V{}, false};
    return {it->second, true};
}

// an index expression used as a value: slices give the element, maps the value or zero
template <class T> constexpr T& rv(T& x) { return x; }
template <class T> constexpr const T& rv(const T& x) { return x; }
template <class K, class V> V rv(const MapRef<K, V>& r) { return r.get(); }
template <class K, class V> V def(const MapRef<K, V>& r) { return r.get(); }

// ---------------------------------------------------------------------------------------------------- append / copy
template <class S> struct slice_elem;
template <class T> struct slice_elem<Slice<T>> { using type = T; };
template <class S> auto as_slice(const S& s) {
    if constexpr (is_slice<S>::value) return s;
    else return rv(s);
}
template <class S, class... A> auto append(const S& s0, A... a) {
    auto s = as_slice(rv(s0));
    using SL = decltype(s);
    using T = typename slice_elem<SL>::type;
    SL r = grow(s, (long long)sizeof...(A));
    ((r.ptr[r.len++] = T(a)), ...);
    return r;
}
struct Spread {};  // append(a, b...)
template <class S, class B> auto append_spread(const S& s0, const B& b0) {
    auto s = as_slice(rv(s0));
    using SL = decltype(s);
    using T = typename slice_elem<SL>::type;
    if constexpr (std::is_same_v<B, String>) {
        SL r = grow(s, b0.n);
        for (long long i = 0; i < b0.n; i++) r.ptr[r.len++] = T((uint8_t)b0.ptr[i]);
        return r;
    } else {
        auto b = as_slice(rv(b0));
        SL r = grow(s, b.len);
        std::copy(b.ptr, b.ptr + b.len, r.ptr + r.len);  // distinct or same backing array: source precedes the write
        r.len += b.len;
        return r;
    }
}
template <class T> int_ copy(const Slice<T>& dst, const Slice<T>& src) {
    long long n = std::min(dst.len, src.len);
    if (n > 0) std::memmove((void*)dst.ptr, (const void*)src.ptr, (size_t)n * sizeof(T));
    return int_(n);
}
inline int_ copy(const Slice<byte>& dst, const String& src) {
    long long n = std::min(dst.len, src.n);
    if (n > 0) std::memmove((void*)dst.ptr, src.ptr, (size_t)n);
    return int_(n);
}
template <class T, class... A> Slice<T> make_slice(A... a) {
    long long v[] = {to_i64(a)...};
    return Slice<T>::make(v[0], sizeof...(A) > 1 ? v[sizeof...(A) - 1] : v[0]);
}
template <class A, class B> constexpr auto min(A a, B b) { return b < a ? decltype(a + b)(b) : decltype(a + b)(a); }
template <class A, class B> constexpr auto max(A a, B b) { return b > a ? decltype(a + b)(b) : decltype(a + b)(a); }

template <class X> [[noreturn]] void panic(const X& x) {
    if constexpr (std::is_same_v<X, String>) panic_msg(x.str());
    else if constexpr (std::is_convertible_v<X, String>) panic_msg(String(x).str());
    else panic_msg("(value)");
}

// ---------------------------------------------------------------------------------------------------- channels, goroutines
template <class T> struct ChanState {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<T> q;
    long long capacity = 0;
    long long receivers_waiting = 0;
    long long taken = 0, sent = 0;
    bool closed = false;
};
template <class T> struct Chan {
    std::shared_ptr<ChanState<T>> s;
    Chan() = default;
    Chan(Nil) {}
    static Chan make(long long capacity = 0) {
        Chan c;
        c.s = std::make_shared<ChanState<T>>();
        c.s->capacity = capacity;
        return c;
    }
    void send(const T& v) const {
        if (!s) for (;;) std::this_thread::sleep_for(std::chrono::hours(1));  // a nil channel blocks forever
        std::unique_lock<std::mutex> lk(s->mu);
        if (s->closed) panic_msg("send on closed channel");
        if (s->capacity > 0) {
            s->cv.wait(lk, [&] { return (long long)s->q.size() < s->capacity || s->closed; });
            if (s->closed) panic_msg("send on closed channel");
            s->q.push_back(v);
            s->cv.notify_all();
            return;
        }
        // unbuffered: hand over and wait until a receiver has taken THIS value
        s->q.push_back(v);
        long long ticket = ++s->sent;
        s->cv.notify_all();
        s->cv.wait(lk, [&] { return s->taken >= ticket; });
    }
    std::tuple<T, bool> recv2() const {
        if (!s) for (;;) std::this_thread::sleep_for(std::chrono::hours(1));
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return !s->q.empty() || s->closed; });
        if (s->q.empty()) return {T{}, false};
        T v = s->q.front();
        s->q.pop_front();
        s->taken++;
        s->cv.notify_all();
        return {v, true};
    }
    T recv() const { return std::get<0>(recv2()); }
    friend bool operator==(const Chan& c, Nil) { return !c.s; }
    friend bool operator!=(const Chan& c, Nil) { return !!c.s; }
};
template <class T> void close(const Chan<T>& c) {
    if (!c.s) panic_msg("close of nil channel");
    std::unique_lock<std::mutex> lk(c.s->mu);
    if (c.s->closed) panic_msg("close of closed channel");
    c.s->closed = true;
    c.s->cv.notify_all();
}
template <class F> void spawn(F f) { std::thread(std::move(f)).detach(); }

struct Defer {  // `defer` statements of one function call, run in reverse at exit
    std::vector<std::function<void()>> fs;
    void add(std::function<void()> f) { fs.push_back(std::move(f)); }
    ~Defer() {
        for (size_t i = fs.size(); i-- > 0;) fs[i]();
    }
};

// ---------------------------------------------------------------------------------------------------- range
template <class T> struct RangeSlice {
    Slice<T> s;  // the range expression is evaluated once; the length is fixed at loop entry
    long long i = -1;
    bool next() { return ++i < s.len; }
    int_ key() const { return int_(i); }
    T val() const { return s.ptr[i]; }
};
template <class T, long long N> struct RangeArray {
    Array<T, N> a;  // ranging over an array VALUE with a second variable copies the array
    long long i = -1;
    bool next() { return ++i < N; }
    int_ key() const { return int_(i); }
    T val() const { return a.a[i]; }
};
struct RangeString {
    String s;
    long long i = 0, start = 0;
    long long r = 0;
    bool next() {
        if (i >= s.n) return false;
        start = i;
        unsigned char c = (unsigned char)s.ptr[i];
        int need = c < 0x80 ? 0 : (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : -1;
        if (need < 0 || i + need >= s.n + 0 && need > 0 && i + need > s.n - 1 + 0) {
            if (need < 0 || i + need > s.n - 1) { r = 0xFFFD; i += 1; return true; }
        }
        if (need == 0) { r = c; i += 1; return true; }
        long long v = c & (0xFF >> (need + 2));
        for (int k = 1; k <= need; k++) {
            unsigned char cc = (unsigned char)s.ptr[i + k];
            if ((cc >> 6) != 2) { r = 0xFFFD; i += 1; return true; }
            v = (v << 6) | (cc & 63);
        }
        r = v;
        i += need + 1;
        return true;
    }
    int_ key() const { return int_(start); }
    rune val() const { return rune((int32_t)r); }
};
template <class K, class V> struct RangeMap {
    Map<K, V> m;
    std::vector<K> keys;  // snapshot: entries removed during the loop are skipped, added ones are not visited
    long long i = -1;
    bool next() {
        while (++i < (long long)keys.size())
            if (m.p->count(keys[i])) return true;
        return false;
    }
    K key() const { return keys[i]; }
    V val() const { return m.p->find(keys[i])->second; }
};
template <class T> struct RangeChan {
    Chan<T> c;
    T cur{};
    bool next() {
        auto [v, ok] = c.recv2();
        cur = v;
        return ok;
    }
    T key() const { return cur; }
};
struct RangeInt {
    long long n, i = -1;
    bool next() { return ++i < n; }
    int_ key() const { return int_(i); }
};
template <class T> RangeSlice<T> range(const Slice<T>& s) { return {s}; }
template <class T, long long N> RangeArray<T, N> range(const Array<T, N>& a) { return {a}; }
template <class T, long long N> RangeSlice<T> range(Array<T, N>* a) { return {slice(a, nobound, nobound)}; }
inline RangeString range(const String& s) { return {s}; }
template <class K, class V> RangeMap<K, V> range(const Map<K, V>& m) {
    RangeMap<K, V> r{m};
    if (m.p)
        for (auto& kv : *m.p) r.keys.push_back(kv.first);
    return r;
}
template <class K, class V> auto range(const MapRef<K, V>& r) { return range(r.get()); }
template <class T> RangeChan<T> range(const Chan<T>& c) { return {c}; }
template <class T> RangeInt range(Num<T> n) { return {(long long)n.v}; }
inline RangeInt range(UInt n) { return {(long long)n.v}; }

// ---------------------------------------------------------------------------------------------------- error, any
struct error {
    struct I {
        virtual String Error() = 0;
        virtual ~I() {}
    };
    std::shared_ptr<I> p;
    error() = default;
    error(Nil) {}
    String Error() const {
        if (!p) panic_msg("nil error");
        return p->Error();
    }
    friend bool operator==(const error& e, Nil) { return !e.p; }
    friend bool operator!=(const error& e, Nil) { return !!e.p; }
};
struct StringError : error::I {
    String s;
    explicit StringError(String x) : s(std::move(x)) {}
    String Error() override { return s; }
};
inline error make_error(const String& s) {
    error e;
    e.p = std::make_shared<StringError>(s);
    return e;
}

// x.(T) for a CONCRETE type T: the dynamic type stored in the interface value must be exactly T
template <class T, class W> std::tuple<T, bool> type_assert2(const W& w) {
    static_assert(!requires { typename T::I; }, "type assertion to an interface type is not supported");
    auto* m = dynamic_cast<typename W::template M<T>*>(w.p.get());
    if (!m) return {T{}, false};
    return {m->v, true};
}
template <class T, class W> T type_assert(const W& w) {
    auto [v, ok] = type_assert2<T>(w);
    if (!ok) panic_msg("interface conversion: dynamic type is not the asserted type");
    return v;
}

}  // namespace go

// ============================================================================================ standard library
namespace P_math {
using namespace go;
inline uint32 Float32bits(float32 f) {
    uint32_t u;
    std::memcpy(&u, &f.v, 4);
    return uint32(u);
}
inline float32 Float32frombits(uint32 b) {
    float f;
    std::memcpy(&f, &b.v, 4);
    return float32(f);
}
inline uint64 Float64bits(float64 f) {
    unsigned long long u;
    std::memcpy(&u, &f.v, 8);
    return uint64(u);
}
inline float64 Ceil(float64 x) { return float64(std::ceil(x.v)); }
inline float64 Floor(float64 x) { return float64(std::floor(x.v)); }
inline float64 Abs(float64 x) { return float64(std::fabs(x.v)); }
inline float64 Sqrt(float64 x) { return float64(std::sqrt(x.v)); }
inline float64 Log(float64 x) { return float64(std::log(x.v)); }
// math.Log2 as the Go library computes it (src/math/log10.go): Frexp, exact for powers of two
inline float64 Log2(float64 x) {
    int e;
    double frac = std::frexp(x.v, &e);
    if (frac == 0.5) return float64((double)(e - 1));
    return float64((double)e + std::log(frac) * (1 / 0.693147180559945309417232121458176568));
}
inline float64 Pow(float64 x, float64 y) { return float64(std::pow(x.v, y.v)); }
inline constexpr UFloat Pi{3.14159265358979323846264338327950288419716939937510582097494459};
inline constexpr UInt MaxInt32{2147483647}, MaxUint8{255}, MaxUint16{65535};
}  // namespace P_math

namespace P_sync {
struct WaitGroup {
    std::mutex mu;
    std::condition_variable cv;
    long long n = 0;
    template <class X> void Add(X d) {
        std::unique_lock<std::mutex> lk(mu);
        n += go::to_i64(d);
        if (n < 0) go::panic_msg("sync: negative WaitGroup counter");
        if (n == 0) cv.notify_all();
    }
    void Done() { Add(go::UInt(-1)); }
    void Wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return n == 0; });
    }
};
struct Mutex {
    std::mutex mu;
    void Lock() { mu.lock(); }
    void Unlock() { mu.unlock(); }
};
struct Once {
    std::once_flag flag;
    template <class F> void Do(F f) { std::call_once(flag, f); }
};
}  // namespace P_sync

namespace P_strconv {
using namespace go;
inline String Itoa(int_ i) { return String(std::to_string(i.v)); }
inline std::string format_base(unsigned long long v, int base) {
    if (base < 2 || base > 36) panic_msg("strconv: illegal AppendInt/FormatInt base");
    if (!v) return "0";
    std::string s;
    while (v) {
        s += "0123456789abcdefghijklmnopqrstuvwxyz"[v % base];
        v /= base;
    }
    std::reverse(s.begin(), s.end());
    return s;
}
template <class B> String FormatUint(uint64 v, B base) { return String(format_base(v.v, (int)to_i64(base))); }
template <class B> String FormatInt(int64 v, B base) {
    if (v.v < 0) return String("-" + format_base(0ULL - (unsigned long long)v.v, (int)to_i64(base)));
    return String(format_base((unsigned long long)v.v, (int)to_i64(base)));
}
// strconv.ParseUint / ParseInt for an explicit base 2..36 (base 0 prefixes and underscores are not implemented: panic)
inline std::tuple<uint64, error> ParseUintImpl(const String& s, long long base, long long bits, const char* fn) {
    auto fail = [&](const char* why, unsigned long long v) {
        return std::tuple<uint64, error>(uint64(v), make_error(String(std::string("strconv.") + fn + ": parsing \"" + s.str() + "\": " + why)));
    };
    if (base < 2 || base > 36) panic_msg("go2cxx runtime: strconv base 0 / out of range is not implemented");
    if (bits == 0) bits = 64;
    if (s.n == 0) return fail("invalid syntax", 0);
    unsigned long long maxv = bits >= 64 ? ~0ULL : (1ULL << bits) - 1;
    unsigned long long v = 0;
    for (long long i = 0; i < s.n; i++) {
        char c = s.ptr[i];
        int d = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'z' ? c - 'a' + 10 : c >= 'A' && c <= 'Z' ? c - 'A' + 10 : 99;
        if (d >= base) return fail("invalid syntax", 0);
        if (v > (~0ULL - d) / base) return fail("value out of range", maxv);
        v = v * base + d;
        if (v > maxv) return fail("value out of range", maxv);
    }
    return {uint64(v), error()};
}
template <class B, class W> std::tuple<uint64, error> ParseUint(const String& s, B base, W bits) {
    return ParseUintImpl(s, to_i64(base), to_i64(bits), "ParseUint");
}
template <class B, class W> std::tuple<int64, error> ParseInt(const String& s, B base, W bits0) {
    long long bits = to_i64(bits0);
    if (bits == 0) bits = 64;
    bool neg = false;
    String t = s;
    if (s.n > 0 && (s.ptr[0] == '+' || s.ptr[0] == '-')) {
        neg = s.ptr[0] == '-';
        t = slice(s, UInt(1), nobound);
    }
    auto [u, e] = ParseUintImpl(t, to_i64(base), 64, "ParseInt");
    if (e != nil && u.v == 0) return {int64(0), e};
    unsigned long long cutoff = 1ULL << (bits - 1);
    if (!neg && u.v >= cutoff) return {int64((long long)(cutoff - 1)), make_error(String("strconv.ParseInt: parsing \"" + s.str() + "\": value out of range"))};
    if (neg && u.v > cutoff) return {int64(-(long long)cutoff), make_error(String("strconv.ParseInt: parsing \"" + s.str() + "\": value out of range"))};
    return {int64(neg ? -(long long)u.v : (long long)u.v), error()};
}
}  // namespace P_strconv

namespace P_fmt {
using namespace go;
template <class X> struct has_String_method {
    template <class Y> static auto test(int) -> decltype(std::declval<const Y&>().String(), std::true_type());
    template <class> static std::false_type test(...);
    static constexpr bool value = decltype(test<X>(0))::value;
};
struct Arg {
    enum { INT, UINT, FLOAT, STR, BOOL, OTHER } kind = OTHER;
    long long i = 0;
    unsigned long long u = 0;
    double f = 0;
    std::string s;
    bool b = false;
};
template <class X> Arg to_arg(const X& x) {
    Arg a;
    if constexpr (std::is_same_v<X, UInt>) { a.kind = Arg::INT; a.i = (long long)x.v; a.u = (unsigned long long)x.v; }
    else if constexpr (std::is_same_v<X, UFloat>) { a.kind = Arg::FLOAT; a.f = x.v; }
    else if constexpr (std::is_same_v<X, bool>) { a.kind = Arg::BOOL; a.b = x; }
    else if constexpr (std::is_same_v<X, String>) { a.kind = Arg::STR; a.s = x.str(); }
    else if constexpr (is_num<X>::value) {
        if constexpr (X::is_float) { a.kind = Arg::FLOAT; a.f = x.v; }
        else if constexpr (std::is_signed_v<typename X::raw>) { a.kind = Arg::INT; a.i = x.v; a.u = (unsigned long long)x.v; }
        else { a.kind = Arg::UINT; a.u = x.v; a.i = (long long)x.v; }
    } else if constexpr (has_String_method<X>::value) { a.kind = Arg::STR; a.s = x.String().str(); }
    else if constexpr (std::is_same_v<X, error>) { a.kind = Arg::STR; a.s = x.p ? x.Error().str() : "<nil>"; }
    else if constexpr (std::is_same_v<X, Slice<byte>>) { a.kind = Arg::STR; a.s = std::string((const char*)x.ptr, (size_t)x.len); }
    else { a.kind = Arg::STR; a.s = "?"; }
    return a;
}
inline std::string pad(std::string body, long long width, bool left, bool zero, bool numeric) {
    if ((long long)body.size() >= width) return body;
    size_t n = (size_t)width - body.size();
    if (left) return body + std::string(n, ' ');
    if (zero && numeric) {
        size_t sign = (!body.empty() && (body[0] == '-' || body[0] == '+')) ? 1 : 0;
        return body.substr(0, sign) + std::string(n, '0') + body.substr(sign);
    }
    return std::string(n, ' ') + body;
}
inline std::string quote(const std::string& s) {
    std::string o = "\"";
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
        else if (c == '\n') o += "\\n";
        else if (c == '\t') o += "\\t";
        else if (c == '\r') o += "\\r";
        else if (c < 0x20 || c == 0x7f) { char b[8]; std::snprintf(b, sizeof b, "\\x%02x", c); o += b; }
        else o += (char)c;
    }
    return o + "\"";
}
inline std::string format(const std::string& f, const std::vector<Arg>& args) {
    std::string o;
    size_t ai = 0;
    for (size_t i = 0; i < f.size(); i++) {
        if (f[i] != '%') { o += f[i]; continue; }
        i++;
        if (i >= f.size()) { o += "%!(NOVERB)"; break; }
        if (f[i] == '%') { o += '%'; continue; }
        bool left = false, zero = false, plus = false, sharp = false, space = false;
        for (;; i++) {
            if (f[i] == '-') left = true;
            else if (f[i] == '0') zero = true;
            else if (f[i] == '+') plus = true;
            else if (f[i] == '#') sharp = true;
            else if (f[i] == ' ') space = true;
            else break;
        }
        long long width = 0, prec = -1;
        while (i < f.size() && std::isdigit((unsigned char)f[i])) width = width * 10 + (f[i++] - '0');
        if (i < f.size() && f[i] == '.') {
            prec = 0;
            i++;
            while (i < f.size() && std::isdigit((unsigned char)f[i])) prec = prec * 10 + (f[i++] - '0');
        }
        char verb = f[i];
        if (ai >= args.size()) { o += std::string("%!") + verb + "(MISSING)"; continue; }
        const Arg& a = args[ai++];
        std::string body;
        bool numeric = true;
        auto intbody = [&](int base, bool upper) {
            bool neg = a.kind == Arg::INT && a.i < 0;
            unsigned long long mag = neg ? 0ULL - (unsigned long long)a.i : a.u;
            std::string d = P_strconv::format_base(mag, base);
            if (upper) for (auto& c : d) c = (char)std::toupper((unsigned char)c);
            if (prec >= 0 && (long long)d.size() < prec) d = std::string((size_t)prec - d.size(), '0') + d;
            if (sharp && base == 16) d = (upper ? "0X" : "0x") + d;
            if (sharp && base == 8) d = "0" + d;
            if (sharp && base == 2) d = "0b" + d;
            return (neg ? "-" : plus ? "+" : space ? " " : "") + d;
        };
        bool is_int = a.kind == Arg::INT || a.kind == Arg::UINT;
        switch (verb) {
        case 'd': if (!is_int) goto bad; body = intbody(10, false); break;
        case 'b': if (!is_int) goto bad; body = intbody(2, false); break;
        case 'o': if (!is_int) goto bad; body = intbody(8, false); break;
        case 'x': case 'X':
            if (is_int) body = intbody(16, verb == 'X');
            else if (a.kind == Arg::STR) {
                for (unsigned char c : a.s) { char b[4]; std::snprintf(b, sizeof b, verb == 'x' ? "%02x" : "%02X", c); body += b; }
                numeric = false;
            } else goto bad;
            break;
        case 'c': if (!is_int) goto bad; body = String::utf8(a.i); numeric = false; break;
        case 's':
            numeric = false;
            if (a.kind == Arg::STR) body = prec >= 0 ? a.s.substr(0, (size_t)prec) : a.s;
            else goto bad;
            break;
        case 'q': numeric = false; if (a.kind != Arg::STR) goto bad; body = quote(a.s); break;
        case 't': numeric = false; if (a.kind != Arg::BOOL) goto bad; body = a.b ? "true" : "false"; break;
        case 'f': case 'e': case 'g': {
            if (a.kind != Arg::FLOAT) goto bad;
            char b[512];
            std::string spec = std::string("%") + (plus ? "+" : "") + "." + std::to_string(prec < 0 ? 6 : prec) + verb;
            if (verb == 'g' && prec < 0) go::panic_msg("go2cxx runtime: %g without precision (shortest formatting) is not implemented");
            std::snprintf(b, sizeof b, spec.c_str(), a.f);
            body = b;
            break;
        }
        case 'v':
            if (is_int) body = intbody(10, false);
            else if (a.kind == Arg::STR) { body = a.s; numeric = false; }
            else if (a.kind == Arg::BOOL) { body = a.b ? "true" : "false"; numeric = false; }
            else go::panic_msg("go2cxx runtime: %v of this operand is not implemented");
            break;
        default:
        bad:
            go::panic_msg(std::string("go2cxx runtime: fmt verb %") + verb + " with this operand is not implemented");
        }
        o += pad(body, width, left, zero, numeric);
    }
    if (ai < args.size()) o += "%!(EXTRA)";
    return o;
}
template <class... A> String Sprintf(const String& f, const A&... a) { return String(format(f.str(), std::vector<Arg>{to_arg(a)...})); }
template <class... A> error Errorf(const String& f, const A&... a) { return make_error(Sprintf(f, a...)); }
template <class... A> String Sprint(const A&... a) {
    std::string o;
    ((o += format("%v", {to_arg(a)})), ...);
    return String(o);
}
template <class... A> void Println(const A&... a) {
    std::string o;
    size_t k = 0;
    ((o += (k++ ? " " : "") + format("%v", {to_arg(a)})), ...);
    std::printf("%s\n", o.c_str());
}
template <class... A> void Printf(const String& f, const A&... a) { std::fputs(Sprintf(f, a...).str().c_str(), stdout); }
}  // namespace P_fmt

namespace P_log {
template <class... A> void Println(const A&... a) {
    std::string o;
    size_t k = 0;
    ((o += (k++ ? " " : "") + P_fmt::format("%v", {P_fmt::to_arg(a)})), ...);
    std::fprintf(stderr, "%s\n", o.c_str());
}
template <class... A> void Printf(const go::String& f, const A&... a) { std::fprintf(stderr, "%s\n", P_fmt::Sprintf(f, a...).str().c_str()); }
template <class... A> [[noreturn]] void Fatalf(const go::String& f, const A&... a) {
    std::fprintf(stderr, "%s\n", P_fmt::Sprintf(f, a...).str().c_str());
    std::exit(1);
}
template <class... A> [[noreturn]] void Fatal(const A&... a) {
    Println(a...);
    std::exit(1);
}
}  // namespace P_log

namespace P_bytes {
struct Reader {
    go::Slice<go::byte> s;
    long long pos = 0;
};
inline Reader* NewReader(const go::Slice<go::byte>& b) { return new Reader{b, 0}; }
inline bool Equal(const go::Slice<go::byte>& a, const go::Slice<go::byte>& b) {
    return a.len == b.len && (a.len == 0 || std::memcmp((const void*)a.ptr, (const void*)b.ptr, (size_t)a.len) == 0);
}
}  // namespace P_bytes

namespace P_strings {
using namespace go;
inline String Join(const Slice<String>& parts, const String& sep) {
    std::string o;
    for (long long i = 0; i < parts.len; i++) {
        if (i) o += sep.str();
        o += parts.ptr[i].str();
    }
    return String(o);
}
inline String Repeat(const String& s, int_ n) {
    std::string o;
    for (long long i = 0; i < n.v; i++) o += s.str();
    return String(o);
}
inline bool Contains(const String& s, const String& sub) { return s.str().find(sub.str()) != std::string::npos; }
inline bool HasPrefix(const String& s, const String& p) { return s.n >= p.n && std::memcmp(s.ptr, p.ptr, (size_t)p.n) == 0; }
inline String ToLower(const String& s) {
    std::string o = s.str();
    for (auto& c : o) c = (char)std::tolower((unsigned char)c);
    return String(o);
}
}  // namespace P_strings

namespace P_binary {  // encoding/binary: the ByteOrder values
using namespace go;
struct BigEndianOrder {
    uint16 Uint16(const Slice<byte>& b) const { (void)b[UInt(1)]; return uint16((uint16_t)((b.ptr[0].v << 8) | b.ptr[1].v)); }
    uint32 Uint32(const Slice<byte>& b) const {
        (void)b[UInt(3)];
        return uint32(((uint32_t)b.ptr[0].v << 24) | ((uint32_t)b.ptr[1].v << 16) | ((uint32_t)b.ptr[2].v << 8) | b.ptr[3].v);
    }
    uint64 Uint64(const Slice<byte>& b) const {
        (void)b[UInt(7)];
        unsigned long long v = 0;
        for (int i = 0; i < 8; i++) v = (v << 8) | b.ptr[i].v;
        return uint64(v);
    }
    void PutUint16(const Slice<byte>& b, uint16 v) const { (void)b[UInt(1)]; b.ptr[0] = byte((uint8_t)(v.v >> 8)); b.ptr[1] = byte((uint8_t)v.v); }
    void PutUint32(const Slice<byte>& b, uint32 v) const {
        (void)b[UInt(3)];
        for (int i = 0; i < 4; i++) b.ptr[i] = byte((uint8_t)(v.v >> (24 - 8 * i)));
    }
};
struct LittleEndianOrder {
    uint16 Uint16(const Slice<byte>& b) const { (void)b[UInt(1)]; return uint16((uint16_t)((b.ptr[1].v << 8) | b.ptr[0].v)); }
    uint32 Uint32(const Slice<byte>& b) const {
        (void)b[UInt(3)];
        return uint32(((uint32_t)b.ptr[3].v << 24) | ((uint32_t)b.ptr[2].v << 16) | ((uint32_t)b.ptr[1].v << 8) | b.ptr[0].v);
    }
    void PutUint16(const Slice<byte>& b, uint16 v) const { (void)b[UInt(1)]; b.ptr[1] = byte((uint8_t)(v.v >> 8)); b.ptr[0] = byte((uint8_t)v.v); }
};
// binary.Read(r, order, &data) for fixed-size data: numbers, arrays and structs of them, in declaration order
template <class X> struct is_array : std::false_type {};
template <class T, long long N> struct is_array<Array<T, N>> : std::true_type {};
template <class O, class X> bool read_value(P_bytes::Reader* r, const O& order, X& x) {
    if constexpr (is_num<X>::value) {
        constexpr long long n = (long long)sizeof(typename X::raw);
        static_assert(!X::is_float, "binary.Read of floats is not implemented");
        if (r->pos + n > r->s.len) return false;
        unsigned long long v = 0;
        for (long long i = 0; i < n; i++) {
            unsigned long long b = r->s.ptr[r->pos + i].v;
            if constexpr (std::is_same_v<O, LittleEndianOrder>) v |= b << (8 * i);
            else v = (v << 8) | b;
        }
        r->pos += n;
        x = X((typename X::raw)v);
        return true;
    } else if constexpr (std::is_same_v<X, bool>) {
        if (r->pos + 1 > r->s.len) return false;
        x = r->s.ptr[r->pos++].v != 0;
        return true;
    } else if constexpr (requires { typename X::go_array_elem; }) {
        for (auto& e : x.a)
            if (!read_value(r, order, e)) return false;
        return true;
    } else {
        bool ok = true;
        x.go_fields_([&](auto& f) { ok = ok && read_value(r, order, f); });
        return ok;
    }
}
template <class O, class X> error Read(P_bytes::Reader* r, const O& order, X* data) {
    if (!read_value(r, order, *data)) return make_error(String("unexpected EOF"));
    return error();
}
inline constexpr BigEndianOrder BigEndian{};
inline constexpr LittleEndianOrder LittleEndian{};
}  // namespace P_binary
