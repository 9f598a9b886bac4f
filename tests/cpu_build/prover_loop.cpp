// A libsnark-shaped r1cs_gg_ppzksnark prover translation unit written against UPSTREAM's include paths and namespaces only
// (README.md:272-273 points at crypto3-zk prover.hpp#L73; reached from bin/cli/include/nil/vote_saver/common.hpp:1132-1135):
//     nil::crypto3::math::make_evaluation_domain<FieldType>(m), domain->inverse_fft / fft / multiply_by_coset / divide_by_z_on_coset,
//     nil::crypto3::algebra::multiexp / multiexp_with_mixed_addition<policies::multiexp_method_BDLO12>(begin, end, begin, end, chunks),
//     nil::crypto3::zk::commitments::kc_multiexp_with_mixed_addition<...>(B_query, 0, n, begin, end, chunks)
// It names nothing of this repository.  tests/test_abi.py compiles it with
//     -I include/overlay   (this repository's overlay of those four crypto3 headers)   -I tests/cpu_build/standin   (stand-in value types)
// i.e. "an include-path change and nothing else" is what moves the prover's hot loops onto libvsp_hip.so.  With a GPU the program also
// runs one multiexp through the overlay (2 * generator) and checks the result's x coordinate.
#include <cstdio>
#include <memory>
#include <vector>

#include <nil/crypto3/algebra/curves/bls12.hpp>
#include <nil/crypto3/algebra/multiexp/multiexp.hpp>
#include <nil/crypto3/algebra/multiexp/policies.hpp>
#include <nil/crypto3/math/algorithms/make_evaluation_domain.hpp>
#include <nil/crypto3/math/domains/evaluation_domain.hpp>
#include <nil/crypto3/zk/commitments/knowledge_commitment.hpp>
#include <nil/crypto3/zk/commitments/detail/polynomial/knowledge_commitment_multiexp.hpp>

using namespace nil::crypto3;

template <typename CurveType> struct proving_key {
    typedef typename CurveType::template g1_type<>::value_type g1_value_type;
    typedef typename CurveType::template g2_type<>::value_type g2_value_type;
    g1_value_type alpha_g1, beta_g1, delta_g1;
    g2_value_type beta_g2, delta_g2;
    std::vector<g1_value_type> A_query, H_query, L_query;
    zk::commitments::knowledge_commitment_vector<g2_value_type, g1_value_type> B_query;
    std::size_t num_constraints = 0, num_inputs = 0;
};
template <typename CurveType> struct proof { typename proving_key<CurveType>::g1_value_type g_A, g_C; typename proving_key<CurveType>::g2_value_type g_B; };

// r1cs_to_qap::witness_map tail, as upstream spells it: three inverse transforms, the coset shift, three transforms, H = (A B - C) / Z on the coset
template <typename FieldType>
std::vector<typename FieldType::value_type> witness_map_tail(std::size_t min_size, std::vector<typename FieldType::value_type> &aA,
                                                             std::vector<typename FieldType::value_type> &aB, std::vector<typename FieldType::value_type> &aC) {
    typedef typename FieldType::value_type value_type;
    const std::shared_ptr<math::evaluation_domain<FieldType>> domain = math::make_evaluation_domain<FieldType>(min_size);
    aA.resize(domain->m, value_type::zero()); aB.resize(domain->m, value_type::zero()); aC.resize(domain->m, value_type::zero());
    domain->inverse_fft(aA); domain->inverse_fft(aB); domain->inverse_fft(aC);
    const value_type g = value_type(typename value_type::integral_type(7));          // the field's multiplicative generator
    math::detail::multiply_by_coset(aA, g); domain->fft(aA);                           // upstream's spelling ...
    domain->cosetFFT(aB, g); domain->cosetFFT(aC, g);                                  // ... and libfqfft's
    std::vector<value_type> H(domain->m + 1, value_type::zero());
    for (std::size_t i = 0; i < domain->m; ++i) H[i] = aA[i] * aB[i] - aC[i];
    std::vector<value_type> Hm(H.begin(), H.begin() + domain->m);
    domain->divide_by_z_on_coset(Hm);
    domain->icosetFFT(Hm, g);
    const value_type Zt = domain->compute_vanishing_polynomial(g);
    domain->add_poly_z(Zt, H);
    (void)domain->get_domain_element(1); (void)domain->evaluate_all_lagrange_polynomials(g);
    return Hm;
}

template <typename CurveType>
proof<CurveType> prover_process(const proving_key<CurveType> &pk, const std::vector<typename CurveType::scalar_field_type::value_type> &full_variable_assignment,
                                const std::vector<typename CurveType::scalar_field_type::value_type> &coefficients_for_H,
                                const typename CurveType::scalar_field_type::value_type &r, const typename CurveType::scalar_field_type::value_type &s) {
    typedef algebra::policies::multiexp_method_BDLO12 method;
    const std::size_t chunks = 1, num_variables = full_variable_assignment.size() - 1;
    auto evaluation_At = algebra::multiexp_with_mixed_addition<method>(pk.A_query.begin(), pk.A_query.begin() + num_variables + 1,
                                                                       full_variable_assignment.begin(), full_variable_assignment.begin() + num_variables + 1, chunks);
    auto evaluation_Bt = zk::commitments::kc_multiexp_with_mixed_addition<method>(pk.B_query, 0, num_variables + 1, full_variable_assignment.begin(),
                                                                                  full_variable_assignment.begin() + num_variables + 1, chunks);
    auto evaluation_Ht = algebra::multiexp<method>(pk.H_query.begin(), pk.H_query.begin() + coefficients_for_H.size(), coefficients_for_H.begin(),
                                                   coefficients_for_H.end(), chunks);
    auto evaluation_Lt = algebra::multiexp_with_mixed_addition<method>(pk.L_query.begin(), pk.L_query.end(), full_variable_assignment.begin() + pk.num_inputs + 1,
                                                                       full_variable_assignment.begin() + num_variables + 1, chunks);
    proof<CurveType> pf;
    pf.g_A = pk.alpha_g1 + evaluation_At + r * pk.delta_g1;
    pf.g_B = pk.beta_g2 + evaluation_Bt.g + s * pk.delta_g2;
    auto g1_B = pk.beta_g1 + evaluation_Bt.h + s * pk.delta_g1;
    pf.g_C = evaluation_Ht + evaluation_Lt + s * pf.g_A + r * g1_B - (r * s) * pk.delta_g1;
    return pf;
}

int main() {
    typedef algebra::curves::bls12<381> curve;
    typedef curve::scalar_field_type::value_type fr;
    typedef curve::base_field_type::value_type fp;
    typedef fp::integral_type big;
    auto mk = [](std::initializer_list<std::uint64_t> l) { big b; std::size_t i = 0; for (auto x : l) b.w[i++] = x; return fp(b); };
    proving_key<curve>::g1_value_type g(mk({0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL, 0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL}),
                                        mk({0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL, 0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL}), fp::one());
    try {
        std::vector<proving_key<curve>::g1_value_type> bases{g, g};
        std::vector<fr> scalars{fr(fr::integral_type(1)), fr(fr::integral_type(1))};
        auto two_g = algebra::multiexp<algebra::policies::multiexp_method_BDLO12>(bases.begin(), bases.end(), scalars.begin(), scalars.end(), 1);
        const bool ok = static_cast<std::uint64_t>(two_g.to_affine().X.data) == 0xc39a8c5529bf0f4eULL;      // low word of (2 G).x; the caller compares the printed value with the oracle's as well
        std::printf("2G.x[0] = %016llx\n", (unsigned long long)static_cast<std::uint64_t>(two_g.to_affine().X.data));
        // instantiate the whole loop (never run here: the stand-in has no arithmetic to make a satisfying instance from)
        if (bases.empty()) {
            proving_key<curve> pk; std::vector<fr> z, h;
            (void)prover_process<curve>(pk, z, h, fr::one(), fr::one());
            std::vector<fr> a, b, c; (void)witness_map_tail<curve::scalar_field_type>(8, a, b, c);
        }
        return ok ? 0 : 1;
    } catch (const std::exception &e) { std::printf("no GPU path: %s\n", e.what()); return 77; }
}
