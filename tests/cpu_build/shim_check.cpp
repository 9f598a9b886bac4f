// Compile-and-link check of the C++ shims against a stand-in value type defined HERE (crypto3 headers are absent,
// SURVEY.md F1/F3).  With a GPU it also runs: 2*G via multiexp and an fft/inverse_fft round trip.
#include <cstdio>
#include <cstring>
#include "../../include/vsp/evaluation_domain.hpp"
#include "../../include/vsp/multiexp.hpp"

struct Fr4 { std::uint64_t l[4]; bool operator==(const Fr4 &o) const { return !memcmp(l, o.l, 32); } };
struct Fp6 { std::uint64_t l[6]; };
struct G1pt { Fp6 x, y; bool inf; };
namespace vsp {
template <> struct limb_traits<Fr4> { static constexpr std::size_t limbs = 4; static void to_limbs(const Fr4 &v, std::uint64_t *o) { memcpy(o, v.l, 32); } static Fr4 from_limbs(const std::uint64_t *i) { Fr4 r; memcpy(r.l, i, 32); return r; } };
template <> struct limb_traits<Fp6> { static constexpr std::size_t limbs = 6; static void to_limbs(const Fp6 &v, std::uint64_t *o) { memcpy(o, v.l, 48); } static Fp6 from_limbs(const std::uint64_t *i) { Fp6 r; memcpy(r.l, i, 48); return r; } };
template <> struct point_traits<G1pt> {
    using field_type = Fp6; static constexpr int group = 1;
    static bool is_zero(const G1pt &p) { return p.inf; }
    static void to_affine_xy(const G1pt &p, Fp6 &x, Fp6 &y) { x = p.x; y = p.y; }
    static G1pt from_affine_xy(const Fp6 &x, const Fp6 &y) { return G1pt{x, y, false}; }
    static G1pt zero() { G1pt p; memset(&p, 0, sizeof p); p.inf = true; return p; }
};
}
// stand-ins for knowledge_commitment<T1, T2> and knowledge_commitment_vector<T1, T2> (members as upstream names them)
struct KC { G1pt g; G1pt h; };
struct KCVec { std::vector<std::size_t> indices; std::vector<KC> values; std::size_t domain_size_; };
int main() {
    G1pt g{{{0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL, 0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL}},
           {{0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL, 0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL}}, false};
    std::vector<G1pt> bases{g, g};
    std::vector<Fr4> scalars{Fr4{{1, 0, 0, 0}}, Fr4{{1, 0, 0, 0}}};
    try {
        G1pt r = vsp::multiexp<vsp::policies::multiexp_method_BDLO12>(bases.begin(), bases.end(), scalars.begin(), scalars.end(), 1);
        std::printf("2G.x[0] = %016llx\n", (unsigned long long)r.x.l[0]);
        // kc_multiexp_with_mixed_addition over a sparse pair vector: entries 1, 2, 4 of a domain of 6, index window [1, 5)
        KCVec kv{{0, 1, 2, 4, 5}, {KC{g, g}, KC{g, r}, KC{r, g}, KC{g, g}, KC{g, g}}, 6};
        std::vector<Fr4> ksc{Fr4{{1, 0, 0, 0}}, Fr4{{0, 0, 0, 0}}, Fr4{{5, 0, 0, 0}}, Fr4{{2, 0, 0, 0}}};     // scalars for indices 1, 2, 3, 4
        KC kc = vsp::kc_multiexp_with_mixed_addition<vsp::policies::multiexp_method_BDLO12>(kv, 1, 5, ksc.begin(), ksc.end(), 1);
        // g-half: 1*G + 0*2G + 2*G = 3G;  h-half: 1*2G + 0*G + 2*G = 4G
        std::vector<G1pt> b3{g}; std::vector<Fr4> s3{Fr4{{3, 0, 0, 0}}}, s4{Fr4{{4, 0, 0, 0}}};
        G1pt g3 = vsp::multiexp(b3.begin(), b3.end(), s3.begin(), s3.end()), g4 = vsp::multiexp(b3.begin(), b3.end(), s4.begin(), s4.end());
        bool kc_ok = !memcmp(&kc.g.x, &g3.x, 48) && !memcmp(&kc.g.y, &g3.y, 48) && !memcmp(&kc.h.x, &g4.x, 48) && !memcmp(&kc.h.y, &g4.y, 48);
        std::printf("kc_multiexp %s\n", kc_ok ? "ok" : "MISMATCH");
        if (!kc_ok) return 1;
        bool ok = true;
        for (std::size_t min_size : {5, 8, 11, 70}) {          // 5 -> 5 (step 4+1), 8 -> 8 (basic), 11 -> 12 (step 8+4), 70 -> 72 (step 64+8)
            auto dom = vsp::make_evaluation_domain<Fr4>(min_size);
            std::vector<Fr4> a(dom->m);
            for (std::size_t i = 0; i < a.size(); i++) a[i] = Fr4{{i + 1, 0, 0, 0}};
            auto b = a; dom->fft(b); dom->inverse_fft(b);
            auto c = a; Fr4 g{{7, 0, 0, 0}}; dom->cosetFFT(c, g); dom->icosetFFT(c, g);
            // L_i(x_k) is the indicator of i == k;  Z vanishes on the domain
            Fr4 x3 = dom->get_domain_element(3);
            auto u = dom->evaluate_all_lagrange_polynomials(x3);
            bool ind = true;
            for (std::size_t i = 0; i < u.size(); i++) ind = ind && (u[i] == Fr4{{i == 3 ? 1ull : 0ull, 0, 0, 0}});
            Fr4 z = dom->compute_vanishing_polynomial(x3);
            // add_poly_z then evaluate nothing: just the shape (m + 1 coefficients); divide_by_z_on_coset keeps the size
            std::vector<Fr4> H(dom->m + 1, Fr4{{0, 0, 0, 0}}); dom->add_poly_z(Fr4{{1, 0, 0, 0}}, H);
            auto P = a; dom->divide_by_z_on_coset(P);
            bool good = b == a && c == a && ind && z == Fr4{{0, 0, 0, 0}} && H[dom->m] == Fr4{{1, 0, 0, 0}} && P.size() == dom->m;
            std::printf("domain(min %zu) m=%zu %s: %s\n", min_size, dom->m, dom->is_step_radix2() ? "step_radix2" : "basic_radix2", good ? "ok" : "MISMATCH");
            ok = ok && good;
        }
        bool threw = false;
        try { vsp::basic_radix2_domain<Fr4> bad(12); } catch (const std::invalid_argument &) { threw = true; }
        try { vsp::step_radix2_domain<Fr4> s12(12); ok = ok && s12.m == 12; } catch (const std::invalid_argument &) { ok = false; }
        std::printf("roundtrip %s\n", ok && threw ? "ok" : "MISMATCH");
        return ok && threw ? 0 : 1;
    } catch (const std::exception &e) { std::printf("no GPU path: %s\n", e.what()); return 77; }
}
