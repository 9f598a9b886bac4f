// OVERLAY of crypto3-algebra <nil/crypto3/algebra/multiexp/multiexp.hpp> (absent submodule; its parameter table is included at
// bin/cli/include/nil/vote_saver/common.hpp:38; reached from common.hpp:1132-1135 through the r1cs_gg_ppzksnark prover, README.md:272-273):
//     algebra::multiexp<Method>(vec_start, vec_end, scalar_start, scalar_end, chunks_count)
//     algebra::multiexp_with_mixed_addition<Method>(...)
// in upstream's namespace and with upstream's five arguments, forwarding to the vsp:: shims (include/vsp/multiexp.hpp -> vsp_msm_g1 / vsp_msm_g2
// of libvsp_hip.so).  With include/overlay ahead of crypto3-algebra on the include path the prover's translation unit compiles unchanged.
#pragma once
#include "policies.hpp"

namespace nil { namespace crypto3 { namespace algebra {
template <typename MultiexpMethod, typename InputBaseIterator, typename InputFieldIterator>
typename std::iterator_traits<InputBaseIterator>::value_type
multiexp(InputBaseIterator vec_start, InputBaseIterator vec_end, InputFieldIterator scalar_start, InputFieldIterator scalar_end, const std::size_t chunks_count) {
    return ::vsp::multiexp<MultiexpMethod>(vec_start, vec_end, scalar_start, scalar_end, chunks_count);
}
template <typename MultiexpMethod, typename InputBaseIterator, typename InputFieldIterator>
typename std::iterator_traits<InputBaseIterator>::value_type
multiexp_with_mixed_addition(InputBaseIterator vec_start, InputBaseIterator vec_end, InputFieldIterator scalar_start, InputFieldIterator scalar_end,
                             const std::size_t chunks_count) {
    return ::vsp::multiexp_with_mixed_addition<MultiexpMethod>(vec_start, vec_end, scalar_start, scalar_end, chunks_count);
}
}}}  // namespace nil::crypto3::algebra
