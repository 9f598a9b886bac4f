// OVERLAY of crypto3-math <nil/crypto3/math/domains/evaluation_domain.hpp> (absent submodule, /root/reference/.gitmodules:47-48): the abstract
// evaluation_domain<FieldType> r1cs_to_qap works through -- m, fft, inverse_fft, evaluate_all_lagrange_polynomials, get_domain_element,
// compute_vanishing_polynomial, add_poly_z, divide_by_z_on_coset (and the libfqfft spellings cosetFFT / icosetFFT) -- in upstream's
// namespace, templated on the FIELD type as upstream is (the vsp:: class is templated on its value_type), served by libvsp_hip.so.
#pragma once
#include <vector>
#include "../../../../vsp/crypto3_traits.hpp"
#include "../../../../../vsp/evaluation_domain.hpp"

namespace nil { namespace crypto3 { namespace math {
template <typename FieldType> using evaluation_domain = ::vsp::evaluation_domain<typename FieldType::value_type>;
namespace detail {
// a[i] *= g^i: the coset shift r1cs_to_qap applies between inverse_fft and fft in upstream's spelling (host arithmetic of the caller's own
// value type; the fused form is cosetFFT / icosetFFT)
template <typename Range, typename FieldValueType>
void multiply_by_coset(Range &a, const FieldValueType &g) {
    FieldValueType u = g;
    for (std::size_t i = 1; i < a.size(); ++i) { a[i] *= u; u *= g; }
}
}  // namespace detail
}}}  // namespace nil::crypto3::math
