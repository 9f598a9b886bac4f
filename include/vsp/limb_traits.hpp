// limb_traits.hpp -- how the shims move a caller's field / curve value types across the C ABI.
//
// crypto3 value types keep their integer in a `.data` member backed by crypto3-multiprecision's
// modular_adaptor (bin/cli/include/nil/vote_saver/common.hpp:92,101 read it exactly so).  The shims need only
// "value -> canonical little-endian uint64 limbs" and back, which an integrator provides by specialising
// vsp::limb_traits<T> for the three value types involved (Fr, Fp, Fp2 element).  INTEGRATION.md shows the
// specialisation for crypto3's element_fp / element_fp2.
#pragma once
#include <cstddef>
#include <cstdint>

namespace vsp {

// Primary template: intentionally undefined members -> a clear compile error if a type was not adapted.
template <class T, class Enable = void>
struct limb_traits {
    // static constexpr std::size_t limbs;                       number of uint64 limbs (Fr: 4, Fp: 6, Fp2: 12)
    // static void to_limbs(const T &v, std::uint64_t *out);     canonical (non-Montgomery) little-endian
    // static T from_limbs(const std::uint64_t *in);
};

// Curve points: affine coordinates through the field traits; infinity <-> all limbs zero.
template <class G, class Enable = void>
struct point_traits {
    // using field_type = ...;                                   coordinate value type (Fp or Fp2 element)
    // static bool is_zero(const G &p);
    // static void to_affine_xy(const G &p, field_type &x, field_type &y);
    // static G from_affine_xy(const field_type &x, const field_type &y);
    // static G zero();
    // static constexpr int group;                               1 = G1, 2 = G2
};

}  // namespace vsp
