// OVERLAY of crypto3-math <nil/crypto3/math/domains/step_radix2_domain.hpp> (absent submodule): the concrete domain in upstream's namespace, templated on the field type.
#pragma once
#include "evaluation_domain.hpp"

namespace nil { namespace crypto3 { namespace math {
template <typename FieldType> using step_radix2_domain = ::vsp::step_radix2_domain<typename FieldType::value_type>;
}}}  // namespace nil::crypto3::math
