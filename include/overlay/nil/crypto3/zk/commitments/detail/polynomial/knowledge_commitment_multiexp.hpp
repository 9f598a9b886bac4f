// OVERLAY of crypto3-zk <nil/crypto3/zk/commitments/detail/polynomial/knowledge_commitment_multiexp.hpp> (absent submodule,
// /root/reference/.gitmodules:11-12): the prover's B_query call
//     commitments::kc_multiexp_with_mixed_addition<Method>(vec, min_idx, max_idx, scalar_start, scalar_end, chunks)
// over upstream's knowledge_commitment_vector (members .indices, .values[k].g / .h), forwarding to vsp::kc_multiexp_with_mixed_addition.
#pragma once
#include "../../../../algebra/multiexp/multiexp.hpp"

namespace nil { namespace crypto3 { namespace zk { namespace commitments {
template <typename MultiexpMethod, typename KCVector, typename InputFieldIterator>
auto kc_multiexp_with_mixed_addition(const KCVector &vec, std::size_t min_idx, std::size_t max_idx, InputFieldIterator scalar_start,
                                     InputFieldIterator scalar_end, std::size_t chunks)
    -> decltype(::vsp::kc_multiexp_with_mixed_addition<MultiexpMethod>(vec, min_idx, max_idx, scalar_start, scalar_end, chunks)) {
    return ::vsp::kc_multiexp_with_mixed_addition<MultiexpMethod>(vec, min_idx, max_idx, scalar_start, scalar_end, chunks);
}
}}}}  // namespace nil::crypto3::zk::commitments
