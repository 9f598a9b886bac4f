// STAND-IN for crypto3-zk's knowledge_commitment / knowledge_commitment_vector (libsnark lineage: a sparse vector of (g, h) pairs with
// `indices`, `values`, `domain_size_`): the members kc_multiexp_with_mixed_addition reads.  Shapes only (absent submodule).
#pragma once
#include <cstddef>
#include <vector>
namespace nil { namespace crypto3 { namespace zk { namespace commitments {
template <typename T1, typename T2> struct knowledge_commitment { T1 g; T2 h; };
template <typename T1, typename T2> struct knowledge_commitment_vector {
    std::vector<std::size_t> indices; std::vector<knowledge_commitment<T1, T2>> values; std::size_t domain_size_ = 0;
};
}}}}
