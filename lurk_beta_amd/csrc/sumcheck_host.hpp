// sumcheck_host.hpp - the arithmetic of one sum-check round, shared by the round kernel (sumcheck.hip) and by the host: the binding
// of a table's top variable, the combination functions, and the whole round over host tables that sumcheck_prove switches to once the
// tables are short (LURK_SUMCHECK_HOST_TAIL_LOG).  Host-callable, so the CPU harness runs it against the oracle (tests/host_harness).
#pragma once
#include <cstddef>
#include <vector>

#include "field.cuh"

namespace lurk {

template <class F>
LURK_HD Fe<F> sc_bind(const Fe<F>& lo, const Fe<F>& hi, const Fe<F>& r) {
    return fe_add<F>(lo, fe_mul<F>(r, fe_sub<F>(hi, lo)));
}
template <class F>
LURK_HD Fe<F> sc_comb_cubic(const Fe<F>& a, const Fe<F>& b, const Fe<F>& c, const Fe<F>& d) {
    return fe_mul<F>(a, fe_sub<F>(fe_mul<F>(b, c), d));  // comb_func_outer: a * (b * c - d)
}

// one round on host tables P[0 .. np) of `len` elements: bind (r != NULL: len -> len / 2, in place) then the evaluation sums at
// 0, 2 (, 3) over the bound tables - sumcheck_round_kernel's arithmetic, element for element
template <class F>
inline void sumcheck_host_round(int np, std::vector<Fe<F>>* P, size_t& len, const Fe<F>* r, Fe<F>* ev) {
    if (r) {
        const size_t m = len / 2;
        for (int k = 0; k < np; k++) {
            for (size_t i = 0; i < m; i++) P[k][i] = sc_bind<F>(P[k][i], P[k][m + i], *r);
            P[k].resize(m);
        }
        len = m;
    }
    const int nv = np == 4 ? 3 : 2;
    for (int k = 0; k < nv; k++) ev[k] = fe_zero<F>();
    const size_t h = len / 2;
    for (size_t i = 0; i < h; i++) {
        Fe<F> lo[4], b2[4], b3[4];
        for (int k = 0; k < np; k++) {
            lo[k] = P[k][i];
            const Fe<F> hi = P[k][h + i], d = fe_sub<F>(hi, lo[k]);
            b2[k] = fe_add<F>(hi, d);
            b3[k] = fe_add<F>(b2[k], d);
        }
        if (np == 4) {
            ev[0] = fe_add<F>(ev[0], sc_comb_cubic<F>(lo[0], lo[1], lo[2], lo[3]));
            ev[1] = fe_add<F>(ev[1], sc_comb_cubic<F>(b2[0], b2[1], b2[2], b2[3]));
            ev[2] = fe_add<F>(ev[2], sc_comb_cubic<F>(b3[0], b3[1], b3[2], b3[3]));
        } else {
            ev[0] = fe_add<F>(ev[0], fe_mul<F>(lo[0], lo[1]));
            ev[1] = fe_add<F>(ev[1], fe_mul<F>(b2[0], b2[1]));
        }
    }
}

}  // namespace lurk
