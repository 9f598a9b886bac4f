// evaluation_domain.hpp -- math::evaluation_domain<FieldType> call shape (radix-2 family) over the vsp C ABI.
//
// Upstream (crypto3-math domains/evaluation_domain.hpp, basic_radix2_domain.hpp, algorithms/make_evaluation_domain.hpp;
// absent submodule, /root/reference/.gitmodules:47-48; libfqfft lineage): an abstract class with members
//     std::size_t m;  fft(std::vector<value_type>&), inverse_fft(...), cosetFFT/icosetFFT in libfqfft,
//     divide_by_z_on_coset(std::vector<value_type>&), get_domain_element(idx), compute_vanishing_polynomial(t)
// and a factory make_evaluation_domain<FieldType>(min_size) returning std::shared_ptr<evaluation_domain>.
// Only the power-of-two domain is served by the GPU path; other sizes throw, as basic_radix2_domain's ctor does.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <vector>

#include "../vsp.h"
#include "limb_traits.hpp"
#include "multiexp.hpp"   // default_context

namespace vsp {

template <typename FieldValueType>
class basic_radix2_domain {
    using T = limb_traits<FieldValueType>;
    vsp_ctx *ctx_;
    unsigned log_m_;

    void run(std::vector<FieldValueType> &a, int inverse, const FieldValueType *g) const {
        if (a.size() != m) throw std::invalid_argument("basic_radix2: expected a.size() == this->m");
        std::vector<std::uint64_t> buf(m * 4);
        for (std::size_t i = 0; i < m; i++) T::to_limbs(a[i], &buf[4 * i]);
        std::uint64_t g4[4];
        if (g) T::to_limbs(*g, g4);
        if (vsp_ntt_fr(ctx_, buf.data(), log_m_, inverse, g ? g4 : nullptr) != VSP_OK)
            throw std::runtime_error(std::string("vsp fft failed: ") + vsp_last_error(ctx_));
        for (std::size_t i = 0; i < m; i++) a[i] = T::from_limbs(&buf[4 * i]);
    }

public:
    typedef FieldValueType value_type;
    std::size_t m;

    explicit basic_radix2_domain(std::size_t m_, vsp_ctx *ctx = nullptr) : ctx_(ctx ? ctx : default_context()), m(m_) {
        static_assert(T::limbs == 4, "evaluation_domain is served for BLS12-381 Fr");
        if (m < 1 || (m & (m - 1))) throw std::invalid_argument("basic_radix2(): expected m a power of two");
        log_m_ = 0;
        while ((std::size_t(1) << log_m_) < m) log_m_++;
        if (log_m_ > 28) throw std::invalid_argument("basic_radix2(): m too large");
    }
    void fft(std::vector<value_type> &a) const { run(a, 0, nullptr); }
    void inverse_fft(std::vector<value_type> &a) const { run(a, 1, nullptr); }
    void cosetFFT(std::vector<value_type> &a, const value_type &g) const { run(a, 0, &g); }
    void icosetFFT(std::vector<value_type> &a, const value_type &g) const { run(a, 1, &g); }
};

template <typename FieldValueType>
std::shared_ptr<basic_radix2_domain<FieldValueType>> make_evaluation_domain(std::size_t min_size, vsp_ctx *ctx = nullptr) {
    std::size_t m = 1;
    while (m < min_size) m <<= 1;
    return std::make_shared<basic_radix2_domain<FieldValueType>>(m, ctx);
}

}  // namespace vsp
