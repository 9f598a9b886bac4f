// Lane view of the group arithmetic, shared by the MSM and fixed-base kernels.
#pragma once
#include "curve.h"
#include "fp28.h"

namespace vsp {

// Lane view of the group arithmetic.  G1: one lane per point, element type Fp.  G2: every Fp2 value is split over a lane pair
// (field.h Fp2L: even lane c0, odd lane c1), so each lane carries the register footprint of the G1 kernels; one lane holding
// whole Fp2 values needs > 256 VGPRs and is confined to one wave per SIMD, where v_mad_u64_u32 issues at half rate
// (2^18-point G2 MSM: 11.9 ms that way).  Kernels are written over LOGICAL threads: lid() of them per block of
// NT physical threads; at(i) maps a logical LDS slot to this lane's physical slot; memory keeps the standard XYZZ<F> layout.
template <class F> struct LaneView {
    using E = F;
    static constexpr unsigned LANES = 1;
    __device__ __forceinline__ static unsigned comp() { return 0; }
    __device__ __forceinline__ static Affine<E> load(const Affine<F> *p) { return *p; }
    __device__ __forceinline__ static XYZZ<E> load(const XYZZ<F> *p) { return *p; }
    __device__ __forceinline__ static void store(XYZZ<F> *p, const XYZZ<E> &v) { *p = v; }
};
template <> struct LaneView<Fp2> {
    using E = Fp2L;
    static constexpr unsigned LANES = 2;
    __device__ __forceinline__ static unsigned comp() { return threadIdx.x & 1; }
    __device__ __forceinline__ static Affine<E> load(const Affine<Fp2> *p) {
        const Fp *row = reinterpret_cast<const Fp *>(p);                 // x.c0, x.c1, y.c0, y.c1
        Affine<E> r; r.x.v = row[comp()]; r.y.v = row[2 + comp()]; return r;
    }
    __device__ __forceinline__ static XYZZ<E> load(const XYZZ<Fp2> *p) {
        const Fp *row = reinterpret_cast<const Fp *>(p);                 // X.c0, X.c1, Y.c0, Y.c1, ZZ.c0, ...
        XYZZ<E> r; r.X.v = row[comp()]; r.Y.v = row[2 + comp()]; r.ZZ.v = row[4 + comp()]; r.ZZZ.v = row[6 + comp()]; return r;
    }
    __device__ __forceinline__ static void store(XYZZ<Fp2> *p, const XYZZ<E> &v) {
        Fp *row = reinterpret_cast<Fp *>(p);
        row[comp()] = v.X.v; row[2 + comp()] = v.Y.v; row[4 + comp()] = v.ZZ.v; row[6 + comp()] = v.ZZZ.v;
    }
};
// the 28-bit form of the same split (fp28.h): memory XYZZ<Fp2x28> = X.c0, X.c1, Y.c0, ... (56 bytes each), a lane holds XYZZ<Fp28L>
template <> struct LaneView<Fp2x28> {
    using E = Fp28L;
    static constexpr unsigned LANES = 2;
    __device__ __forceinline__ static unsigned comp() { return threadIdx.x & 1; }
    __device__ __forceinline__ static XYZZ<E> load(const XYZZ<Fp2x28> *p) {
        const Fp28 *row = reinterpret_cast<const Fp28 *>(p);
        XYZZ<E> r; r.X.v = row[comp()]; r.Y.v = row[2 + comp()]; r.ZZ.v = row[4 + comp()]; r.ZZZ.v = row[6 + comp()]; return r;
    }
    __device__ __forceinline__ static void store(XYZZ<Fp2x28> *p, const XYZZ<E> &v) {
        Fp28 *row = reinterpret_cast<Fp28 *>(p);
        row[comp()] = v.X.v; row[2 + comp()] = v.Y.v; row[4 + comp()] = v.ZZ.v; row[6 + comp()] = v.ZZZ.v;
    }
};
template <class F> __device__ __forceinline__ unsigned lid() { return threadIdx.x / LaneView<F>::LANES; }                       // logical thread in block
template <class F> __device__ __forceinline__ size_t gid() { return ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LaneView<F>::LANES; }
template <class F> __device__ __forceinline__ unsigned at(unsigned logical) { return logical * LaneView<F>::LANES + LaneView<F>::comp(); }


}  // namespace vsp
