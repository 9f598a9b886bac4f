// evaluation_domain.hpp -- math::evaluation_domain<FieldType> call shape (radix-2 family) over the vsp C ABI.
//
// Upstream (crypto3-math domains/evaluation_domain.hpp, basic_radix2_domain.hpp, step_radix2_domain.hpp,
// algorithms/make_evaluation_domain.hpp; absent submodule, /root/reference/.gitmodules:47-48; libfqfft lineage): an abstract
// class with members
//     std::size_t m;  fft(std::vector<value_type>&), inverse_fft(...), evaluate_all_lagrange_polynomials(t),
//     get_domain_element(idx), compute_vanishing_polynomial(t), add_poly_z(coeff, H), divide_by_z_on_coset(P)
// (cosetFFT / icosetFFT in the libfqfft spelling), concrete domains constructed from m, and a factory
// make_evaluation_domain<FieldType>(min_size) returning std::shared_ptr<evaluation_domain>.
// The GPU path serves the two domains that factory can select for BLS12-381 Fr up to 2^28 elements: basic_radix2_domain and
// step_radix2_domain.  Constructors throw std::invalid_argument for sizes their upstream counterparts reject.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../vsp.h"
#include "limb_traits.hpp"
#include "multiexp.hpp"   // default_context

namespace vsp {

template <typename FieldValueType>
class evaluation_domain {
protected:
    using T = limb_traits<FieldValueType>;
    vsp_ctx *ctx_;
    vsp_domain *dom_ = nullptr;

    [[noreturn]] void fail(const char *what) const { throw std::runtime_error(std::string(what) + ": " + vsp_last_error(ctx_)); }
    std::vector<std::uint64_t> flatten(const std::vector<FieldValueType> &a, std::size_t expect) const {
        if (a.size() != expect) throw std::invalid_argument("evaluation_domain: expected a vector of the domain's size");
        std::vector<std::uint64_t> buf(a.size() * 4);
        for (std::size_t i = 0; i < a.size(); i++) T::to_limbs(a[i], &buf[4 * i]);
        return buf;
    }
    static void rebuild(std::vector<FieldValueType> &a, const std::vector<std::uint64_t> &buf) {
        for (std::size_t i = 0; i < a.size(); i++) a[i] = T::from_limbs(&buf[4 * i]);
    }
    void run(std::vector<FieldValueType> &a, int inverse, const FieldValueType *g) const {
        std::vector<std::uint64_t> buf = flatten(a, m);
        std::uint64_t g4[4];
        if (g) T::to_limbs(*g, g4);
        if (vsp_domain_fft(ctx_, dom_, buf.data(), inverse, g ? g4 : nullptr) != VSP_OK) fail("vsp fft failed");
        rebuild(a, buf);
    }
    // kind: -1 whatever make_evaluation_domain selects for `size`, 0 basic radix-2 of exactly `size`, 1 step radix-2 of exactly `size`
    evaluation_domain(std::size_t size, int kind, vsp_ctx *ctx) : ctx_(ctx ? ctx : default_context()) {
        static_assert(T::limbs == 4, "evaluation_domain is served for BLS12-381 Fr");
        dom_ = vsp_domain_create(ctx_, size);
        if (!dom_) throw std::invalid_argument(std::string("evaluation_domain: ") + vsp_last_error(ctx_));
        m = vsp_domain_size(dom_);
        if (kind >= 0 && (m != size || vsp_domain_kind(dom_) != kind)) {
            vsp_domain_free(ctx_, dom_);
            throw std::invalid_argument(kind == 0 ? "basic_radix2(): expected m == 1ul<<log2(m)" : "step_radix2(): expected small_m == 1ul<<log2(small_m)");
        }
    }

public:
    typedef FieldValueType value_type;
    std::size_t m;

    evaluation_domain(const evaluation_domain &) = delete;
    evaluation_domain &operator=(const evaluation_domain &) = delete;
    virtual ~evaluation_domain() { if (dom_) vsp_domain_free(ctx_, dom_); }

    void fft(std::vector<value_type> &a) const { run(a, 0, nullptr); }
    void inverse_fft(std::vector<value_type> &a) const { run(a, 1, nullptr); }
    void cosetFFT(std::vector<value_type> &a, const value_type &g) const { run(a, 0, &g); }
    void icosetFFT(std::vector<value_type> &a, const value_type &g) const { run(a, 1, &g); }

    std::vector<value_type> evaluate_all_lagrange_polynomials(const value_type &t) const {
        std::uint64_t t4[4]; T::to_limbs(t, t4);
        std::vector<std::uint64_t> buf(m * 4);
        if (vsp_domain_lagrange(ctx_, dom_, t4, buf.data()) != VSP_OK) fail("vsp lagrange failed");
        std::vector<value_type> out(m); rebuild(out, buf); return out;
    }
    value_type get_domain_element(std::size_t idx) const {
        std::uint64_t o4[4];
        if (vsp_domain_element(ctx_, dom_, idx, o4) != VSP_OK) throw std::invalid_argument("get_domain_element: index out of range");
        return T::from_limbs(o4);
    }
    value_type compute_vanishing_polynomial(const value_type &t) const {
        std::uint64_t t4[4], o4[4]; T::to_limbs(t, t4);
        if (vsp_domain_vanishing(ctx_, dom_, t4, o4) != VSP_OK) fail("vsp vanishing failed");
        return T::from_limbs(o4);
    }
    void add_poly_z(const value_type &coeff, std::vector<value_type> &H) const {
        std::vector<std::uint64_t> buf = flatten(H, m + 1);
        std::uint64_t c4[4]; T::to_limbs(coeff, c4);
        if (vsp_domain_add_poly_z(ctx_, dom_, c4, buf.data()) != VSP_OK) fail("vsp add_poly_z failed");
        rebuild(H, buf);
    }
    void divide_by_z_on_coset(std::vector<value_type> &P) const {
        std::vector<std::uint64_t> buf = flatten(P, m);
        if (vsp_domain_divide_by_z_on_coset(ctx_, dom_, buf.data()) != VSP_OK) fail("vsp divide_by_z_on_coset failed");
        rebuild(P, buf);
    }
    bool is_step_radix2() const { return vsp_domain_kind(dom_) == 1; }
    const vsp_domain *handle() const { return dom_; }
};

template <typename FieldValueType>
class basic_radix2_domain : public evaluation_domain<FieldValueType> {
public:
    explicit basic_radix2_domain(std::size_t m_, vsp_ctx *ctx = nullptr) : evaluation_domain<FieldValueType>(m_, 0, ctx) {}
};

template <typename FieldValueType>
class step_radix2_domain : public evaluation_domain<FieldValueType> {
public:
    explicit step_radix2_domain(std::size_t m_, vsp_ctx *ctx = nullptr) : evaluation_domain<FieldValueType>(m_, 1, ctx) {}
};

namespace detail {
template <typename FieldValueType>
class selected_domain : public evaluation_domain<FieldValueType> {
public:
    selected_domain(std::size_t min_size, vsp_ctx *ctx) : evaluation_domain<FieldValueType>(min_size, -1, ctx) {}
};
}  // namespace detail

// math::make_evaluation_domain<FieldType>(min_size): basic_radix2(min_size), else step_radix2(min_size), else the same at
// big + rounded_small (the order of upstream's factory; its extended_radix2 and sequence domains cannot be selected for this field)
template <typename FieldValueType>
std::shared_ptr<evaluation_domain<FieldValueType>> make_evaluation_domain(std::size_t min_size, vsp_ctx *ctx = nullptr) {
    return std::make_shared<detail::selected_domain<FieldValueType>>(min_size, ctx);
}

}  // namespace vsp
