// multiexp.hpp -- algebra::multiexp / multiexp_with_mixed_addition call shape over the vsp C ABI.
//
// Upstream signature (crypto3-algebra multiexp/multiexp.hpp, absent submodule; parameter table included at
// bin/cli/include/nil/vote_saver/common.hpp:38; libff multi_exp lineage):
//     template<typename MultiexpMethod, typename InputBaseIterator, typename InputFieldIterator>
//     typename std::iterator_traits<InputBaseIterator>::value_type
//     multiexp(InputBaseIterator vec_start, InputBaseIterator vec_end,
//              InputFieldIterator scalar_start, InputFieldIterator scalar_end, std::size_t chunks_count);
// The shim keeps the five arguments and the value-returning form; `chunks_count` is accepted and ignored (the
// GPU pipeline parallelises over buckets, not over chunks).  The method tag is a template parameter as upstream.
#pragma once
#include <cstdint>
#include <iterator>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../vsp.h"
#include "limb_traits.hpp"

namespace vsp {

namespace policies { struct multiexp_method_BDLO12 {}; struct multiexp_method_naive_plain {}; struct multiexp_method_bos_coster {}; }

// One process-wide context per device for callers that do not manage one (the reference is single-threaded).  A failed creation
// is not cached: the next call tries again.  Contexts live until the process ends.
inline vsp_ctx *default_context(int device = 0) {
    static std::mutex mu;
    static std::map<int, vsp_ctx *> contexts;
    std::lock_guard<std::mutex> lock(mu);
    vsp_ctx *&slot = contexts[device];
    if (!slot) slot = vsp_create(device);
    if (!slot) throw std::runtime_error("vsp: no such HIP device / libvsp_hip.so context could not be created (device " + std::to_string(device) + ")");
    return slot;
}

template <typename MultiexpMethod = policies::multiexp_method_BDLO12, typename InputBaseIterator, typename InputFieldIterator>
typename std::iterator_traits<InputBaseIterator>::value_type
multiexp(InputBaseIterator vec_start, InputBaseIterator vec_end, InputFieldIterator scalar_start, InputFieldIterator scalar_end,
         std::size_t /*chunks_count*/ = 1, vsp_ctx *ctx = nullptr) {
    using G = typename std::iterator_traits<InputBaseIterator>::value_type;
    using S = typename std::iterator_traits<InputFieldIterator>::value_type;
    using PT = point_traits<G>;
    using FT = limb_traits<typename PT::field_type>;
    using ST = limb_traits<S>;
    static_assert(ST::limbs == 4, "scalars must be BLS12-381 Fr elements");
    const std::size_t n = static_cast<std::size_t>(std::distance(vec_start, vec_end));
    if (static_cast<std::size_t>(std::distance(scalar_start, scalar_end)) != n)
        throw std::invalid_argument("multiexp: bases and scalars differ in length");       // upstream asserts this
    if (!ctx) ctx = default_context();
    constexpr std::size_t PL = 2 * FT::limbs;                                               // 12 (G1) or 24 (G2)
    std::vector<std::uint64_t> bases(n * PL, 0), scalars(n * 4, 0);
    std::size_t i = 0;
    for (auto it = vec_start; it != vec_end; ++it, ++i) {
        if (PT::is_zero(*it)) continue;                                                     // infinity = all-zero record
        typename PT::field_type x, y;
        PT::to_affine_xy(*it, x, y);
        FT::to_limbs(x, &bases[i * PL]);
        FT::to_limbs(y, &bases[i * PL + FT::limbs]);
    }
    i = 0;
    for (auto it = scalar_start; it != scalar_end; ++it, ++i) ST::to_limbs(*it, &scalars[i * 4]);
    std::uint64_t out[24] = {0};
    int is_inf = 0;
    int rc = PT::group == 1 ? vsp_msm_g1(ctx, bases.data(), scalars.data(), n, out, &is_inf)
                            : vsp_msm_g2(ctx, bases.data(), scalars.data(), n, out, &is_inf);
    if (rc != VSP_OK) throw std::runtime_error(std::string("vsp multiexp failed: ") + vsp_last_error(ctx));
    if (is_inf) return PT::zero();
    return PT::from_affine_xy(FT::from_limbs(out), FT::from_limbs(out + FT::limbs));
}

// multiexp_with_mixed_addition: same value; zero and one scalars are special-cased inside the bucket pipeline.
template <typename MultiexpMethod = policies::multiexp_method_BDLO12, typename InputBaseIterator, typename InputFieldIterator>
typename std::iterator_traits<InputBaseIterator>::value_type
multiexp_with_mixed_addition(InputBaseIterator vec_start, InputBaseIterator vec_end, InputFieldIterator scalar_start,
                             InputFieldIterator scalar_end, std::size_t chunks_count = 1, vsp_ctx *ctx = nullptr) {
    return multiexp<MultiexpMethod>(vec_start, vec_end, scalar_start, scalar_end, chunks_count, ctx);
}

// ---- kc_multiexp_with_mixed_addition: the prover's B_query call -------------------------------------------------------------
// Upstream (crypto3-zk commitments/knowledge_commitment_multiexp.hpp, absent submodule; libsnark kc_multi_exp_with_mixed_addition):
//     template<typename MultiexpMethod, typename T1, typename T2, typename InputFieldIterator>
//     knowledge_commitment<T1, T2> kc_multiexp_with_mixed_addition(const knowledge_commitment_vector<T1, T2> &vec,
//             std::size_t min_idx, std::size_t max_idx, InputFieldIterator scalar_start, InputFieldIterator scalar_end, std::size_t chunks);
// `vec` is a sparse vector of pairs (g in T1 = G2, h in T2 = G1): `indices` (ascending), `values`, `domain_size_`.  The entries with
// min_idx <= index < max_idx are multiplied by scalar_start[index - min_idx]; the result is the pair (sum in T1, sum in T2).
// The shim needs only the members named below, so upstream's knowledge_commitment_vector / knowledge_commitment plug in unchanged:
//     KCVector: .indices (random access of integers), .values (random access of pairs with .g and .h);  value type = decltype(values[0])
// The two halves run as two multi-exponentiations over the same gathered scalars (a resident key shares one digit sort between
// them: vsp_groth16_prove does exactly that for B_query).
template <typename MultiexpMethod = policies::multiexp_method_BDLO12, typename KCVector, typename InputFieldIterator>
auto kc_multiexp_with_mixed_addition(const KCVector &vec, std::size_t min_idx, std::size_t max_idx, InputFieldIterator scalar_start,
                                     InputFieldIterator scalar_end, std::size_t chunks_count = 1, vsp_ctx *ctx = nullptr)
    -> typename std::decay<decltype(vec.values[0])>::type {
    using KC = typename std::decay<decltype(vec.values[0])>::type;
    using T1 = typename std::decay<decltype(std::declval<KC>().g)>::type;
    using T2 = typename std::decay<decltype(std::declval<KC>().h)>::type;
    using S = typename std::iterator_traits<InputFieldIterator>::value_type;
    const std::size_t span = static_cast<std::size_t>(std::distance(scalar_start, scalar_end));
    if (max_idx < min_idx || span < max_idx - min_idx) throw std::invalid_argument("kc_multiexp: scalar range shorter than the index range");
    std::vector<T1> g; std::vector<T2> h; std::vector<S> sc;
    for (std::size_t k = 0; k < vec.indices.size(); k++) {
        const std::size_t idx = static_cast<std::size_t>(vec.indices[k]);
        if (idx < min_idx) continue;
        if (idx >= max_idx) break;                                                           // indices ascend
        g.push_back(vec.values[k].g); h.push_back(vec.values[k].h);
        sc.push_back(*(scalar_start + static_cast<std::ptrdiff_t>(idx - min_idx)));
    }
    KC out;
    out.g = multiexp_with_mixed_addition<MultiexpMethod>(g.begin(), g.end(), sc.begin(), sc.end(), chunks_count, ctx);
    out.h = multiexp_with_mixed_addition<MultiexpMethod>(h.begin(), h.end(), sc.begin(), sc.end(), chunks_count, ctx);
    return out;
}

}  // namespace vsp
