// OVERLAY of crypto3-math <nil/crypto3/math/algorithms/make_evaluation_domain.hpp> (absent submodule; r1cs_to_qap reaches it from
// bin/cli/include/nil/vote_saver/common.hpp:916-917 and :1132-1135): make_evaluation_domain<FieldType>(m) -> shared_ptr<evaluation_domain<FieldType>>
// -- the first of basic_radix2(m), step_radix2(m), then the same at big + rounded_small, as upstream's factory tries them (its extended and
// sequence domains cannot be selected for BLS12-381 Fr below 2^28 elements).
#pragma once
#include <memory>
#include "../domains/basic_radix2_domain.hpp"
#include "../domains/step_radix2_domain.hpp"

namespace nil { namespace crypto3 { namespace math {
template <typename FieldType>
std::shared_ptr<evaluation_domain<FieldType>> make_evaluation_domain(std::size_t m) {
    return ::vsp::make_evaluation_domain<typename FieldType::value_type>(m);
}
}}}  // namespace nil::crypto3::math
