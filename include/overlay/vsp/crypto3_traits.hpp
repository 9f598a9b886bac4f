// crypto3_traits.hpp -- vsp::limb_traits / vsp::point_traits for crypto3-SHAPED value types, found by their members (no crypto3 header is
// needed to read this file, none is included):
//   field element of Fp / Fr      .data                    an integer (crypto3-multiprecision number) with >>, &, |, << and an explicit
//                                                          conversion to std::uint64_t; constructible from that integer type
//                                                          (bin/cli/include/nil/vote_saver/common.hpp:92: `os << e.data`)
//   element of Fp2                .data[0], .data[1]       two Fp elements (common.hpp:101: `e.data[0].data << ", " << e.data[1].data`)
//   curve point                   .X .Y (.Z), to_affine(), is_zero(), static zero(); field_type::value_type the coordinate type
//                                                          (common.hpp:107-129 prints X, Y, Z of these)
// [UPSTREAM-KNOWLEDGE] the submodules are absent from the reference tree, so these shapes are taken from the reference's own uses of them;
// tests/cpu_build/standin/ holds a stand-in with exactly these members, and tests/test_abi.py compiles a libsnark-shaped prover loop
// against it through the overlay headers.  An integrator whose crypto3 revision differs specialises the two traits by hand.
#pragma once
#include <cstddef>
#include <cstdint>
#include <type_traits>
#include <utility>

#include "../../vsp/limb_traits.hpp"

namespace vsp {
namespace c3detail {
template <class...> using void_t = void;
template <class T, class = void> struct has_scalar_data : std::false_type {};
template <class T> struct has_scalar_data<T, void_t<decltype(std::declval<const T &>().data >> 64)>> : std::true_type {};
template <class T, class = void> struct has_pair_data : std::false_type {};
template <class T> struct has_pair_data<T, void_t<decltype(std::declval<const T &>().data[0].data >> 64)>> : std::true_type {};
template <class T, class = void> struct is_point : std::false_type {};
template <class T> struct is_point<T, void_t<decltype(std::declval<const T &>().to_affine().X), decltype(std::declval<const T &>().is_zero())>> : std::true_type {};
// limbs of a field element: from its policy's modulus_bits where the type names one, else 4 below 2^256 and 6 above
template <class T, class = void> struct limb_count { static constexpr std::size_t value = sizeof(decltype(std::declval<T>().data)) > 40 ? 6 : 4; };
template <class T> struct limb_count<T, void_t<decltype(T::field_type::modulus_bits)>> { static constexpr std::size_t value = (T::field_type::modulus_bits + 63) / 64; };
}  // namespace c3detail

// Fp / Fr element
template <class T>
struct limb_traits<T, typename std::enable_if<c3detail::has_scalar_data<T>::value && !c3detail::has_pair_data<T>::value>::type> {
    using integral_type = typename std::decay<decltype(std::declval<T>().data)>::type;
    static constexpr std::size_t limbs = c3detail::limb_count<T>::value;
    static void to_limbs(const T &v, std::uint64_t *out) {
        const integral_type mask = integral_type(0xFFFFFFFFFFFFFFFFull);
        for (std::size_t i = 0; i < limbs; i++) out[i] = static_cast<std::uint64_t>(integral_type(v.data >> (64 * i)) & mask);
    }
    static T from_limbs(const std::uint64_t *in) {
        integral_type t = integral_type(0);
        for (std::size_t i = limbs; i-- > 0;) t = integral_type(integral_type(t << 64) | integral_type(in[i]));
        return T(t);
    }
};
// Fp2 element: c0 | c1
template <class T>
struct limb_traits<T, typename std::enable_if<c3detail::has_pair_data<T>::value>::type> {
    using base = typename std::decay<decltype(std::declval<T>().data[0])>::type;
    static constexpr std::size_t limbs = 2 * limb_traits<base>::limbs;
    static void to_limbs(const T &v, std::uint64_t *out) { limb_traits<base>::to_limbs(v.data[0], out); limb_traits<base>::to_limbs(v.data[1], out + limb_traits<base>::limbs); }
    static T from_limbs(const std::uint64_t *in) { return T(limb_traits<base>::from_limbs(in), limb_traits<base>::from_limbs(in + limb_traits<base>::limbs)); }
};
// curve point (any coordinate system that offers to_affine())
template <class G>
struct point_traits<G, typename std::enable_if<c3detail::is_point<G>::value>::type> {
    using field_type = typename std::decay<decltype(std::declval<const G &>().to_affine().X)>::type;
    static constexpr int group = limb_traits<field_type>::limbs == 6 ? 1 : 2;
    static bool is_zero(const G &p) { return p.is_zero(); }
    static void to_affine_xy(const G &p, field_type &x, field_type &y) { auto a = p.to_affine(); x = a.X; y = a.Y; }
    static G from_affine_xy(const field_type &x, const field_type &y) { return G(x, y, field_type::one()); }      // Z = 1: the affine point in any projective system
    static G zero() { return G::zero(); }
};
}  // namespace vsp
