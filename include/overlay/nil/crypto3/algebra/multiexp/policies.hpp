// OVERLAY of crypto3-algebra <nil/crypto3/algebra/multiexp/policies.hpp> (absent submodule, /root/reference/.gitmodules:8-9): the method tags,
// in upstream's namespace.  Put include/overlay ahead of crypto3-algebra on the include path and libs/zk picks this up unchanged.
#pragma once
#include "../../../../vsp/crypto3_traits.hpp"
#include "../../../../../vsp/multiexp.hpp"

namespace nil { namespace crypto3 { namespace algebra { namespace policies {
using multiexp_method_BDLO12 = ::vsp::policies::multiexp_method_BDLO12;
using multiexp_method_naive_plain = ::vsp::policies::multiexp_method_naive_plain;
using multiexp_method_bos_coster = ::vsp::policies::multiexp_method_bos_coster;
}}}}  // namespace nil::crypto3::algebra::policies
