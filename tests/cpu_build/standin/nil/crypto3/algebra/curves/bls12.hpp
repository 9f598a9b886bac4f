// STAND-IN for crypto3-algebra's BLS12-381 value types -- SHAPES ONLY, no arithmetic.  The real headers are absent submodules
// (/root/reference/.gitmodules:5-9); what the reference's own code shows of them is reproduced here so that the overlay headers and a
// libsnark-shaped prover loop can be compile-tested:
//     field element            e.data                          (common.hpp:92)           element of Fp2   e.data[0].data, e.data[1].data   (:101)
//     curve point              p.X, p.Y, p.Z, to_affine()      (common.hpp:107-129)      curve            curves::bls12<381> (:148: curves::bls12_381)
//     typename curve::scalar_field_type::value_type            (common.hpp:169)          g1_type<> / g2_type<> ::value_type
// Operators that a prover loop uses on these types (+, *, scalar * point) exist and do NOTHING meaningful: this file is never linked
// into the product, and the test that includes it only compiles and checks what went through libvsp_hip.so.
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>

namespace nil { namespace crypto3 {
namespace multiprecision {
template <std::size_t Words> struct number {                      // a fixed-width unsigned integer with the operators the traits use
    std::uint64_t w[Words];
    number() : w{} {}
    number(std::uint64_t v) : w{} { w[0] = v; }
    number operator>>(unsigned s) const { number r; for (std::size_t i = 0; i < Words; i++) { std::size_t j = i + s / 64; r.w[i] = j < Words ? w[j] >> (s % 64) : 0; if (s % 64 && j + 1 < Words) r.w[i] |= w[j + 1] << (64 - s % 64); } return r; }
    number operator<<(unsigned s) const { number r; for (std::size_t i = Words; i-- > 0;) { std::size_t k = s / 64; r.w[i] = i >= k ? w[i - k] << (s % 64) : 0; if (s % 64 && i >= k + 1) r.w[i] |= w[i - k - 1] >> (64 - s % 64); } return r; }
    number operator&(const number &o) const { number r; for (std::size_t i = 0; i < Words; i++) r.w[i] = w[i] & o.w[i]; return r; }
    number operator|(const number &o) const { number r; for (std::size_t i = 0; i < Words; i++) r.w[i] = w[i] | o.w[i]; return r; }
    explicit operator std::uint64_t() const { return w[0]; }
    bool operator==(const number &o) const { for (std::size_t i = 0; i < Words; i++) if (w[i] != o.w[i]) return false; return true; }
};
}  // namespace multiprecision
namespace algebra {
namespace fields {
template <std::size_t Bits> struct params { static constexpr std::size_t modulus_bits = Bits; typedef multiprecision::number<(Bits + 63) / 64> integral_type; };
namespace detail {
template <typename FieldParams> struct element_fp {
    typedef FieldParams field_type;
    typedef typename FieldParams::integral_type integral_type;
    integral_type data;
    element_fp() {}
    element_fp(const integral_type &d) : data(d) {}
    static element_fp zero() { return element_fp(integral_type(0)); }
    static element_fp one() { return element_fp(integral_type(1)); }
    element_fp operator*(const element_fp &) const { return *this; }      // shapes only
    element_fp &operator*=(const element_fp &) { return *this; }
    element_fp operator+(const element_fp &) const { return *this; }
    element_fp operator-(const element_fp &) const { return *this; }
    element_fp inversed() const { return *this; }
    bool operator==(const element_fp &o) const { return data == o.data; }
};
template <typename FieldParams> struct element_fp2 {
    typedef element_fp<FieldParams> underlying_type;
    std::array<underlying_type, 2> data;
    element_fp2() {}
    element_fp2(const underlying_type &a, const underlying_type &b) : data{{a, b}} {}
    static element_fp2 zero() { return element_fp2(underlying_type::zero(), underlying_type::zero()); }
    static element_fp2 one() { return element_fp2(underlying_type::one(), underlying_type::zero()); }
};
}  // namespace detail
template <std::size_t Bits> struct field { static constexpr std::size_t modulus_bits = Bits; typedef detail::element_fp<params<Bits>> value_type; };
}  // namespace fields
namespace curves {
namespace coordinates { struct affine {}; struct jacobian_with_a4_0 {}; }
namespace detail {
template <typename FieldValue, typename Coordinates> struct curve_element;
template <typename FieldValue> struct curve_element<FieldValue, coordinates::affine> { FieldValue X, Y; };
template <typename FieldValue> struct curve_element<FieldValue, coordinates::jacobian_with_a4_0> {
    typedef FieldValue field_value_type;
    FieldValue X, Y, Z;
    bool inf = true;
    curve_element() {}
    curve_element(const FieldValue &x, const FieldValue &y, const FieldValue &z) : X(x), Y(y), Z(z), inf(false) {}
    static curve_element zero() { return curve_element(); }
    bool is_zero() const { return inf; }
    curve_element<FieldValue, coordinates::affine> to_affine() const { return {X, Y}; }      // the stand-in only ever holds Z = 1
    curve_element operator+(const curve_element &) const { return *this; }                   // shapes only
    curve_element operator-(const curve_element &) const { return *this; }
    template <typename S> friend curve_element operator*(const S &, const curve_element &p) { return p; }
};
}  // namespace detail
template <std::size_t Version> struct bls12 {
    typedef fields::field<381> base_field_type;
    typedef fields::field<255> scalar_field_type;
    template <typename Coordinates = coordinates::jacobian_with_a4_0> struct g1_type { typedef detail::curve_element<typename base_field_type::value_type, Coordinates> value_type; };
    template <typename Coordinates = coordinates::jacobian_with_a4_0> struct g2_type { typedef detail::curve_element<fields::detail::element_fp2<fields::params<381>>, Coordinates> value_type; };
};
typedef bls12<381> bls12_381;
}  // namespace curves
}  // namespace algebra
}}  // namespace nil::crypto3
