// One cell of ksw_extd2 (ksw2_extd2_sse.c:168-321: the score of the base pair, the five-way maximum with its traceback code,
// the new u / v differences and the four gap-extension states with their continuation flags) on plain 32-bit integers -- the
// int8 lanes of the reference never wrap for the scoring parameters the callers admit.  Shared by the row-by-row kernels:
// the grouped DP service (align_kernel_dpg.hip) and the wave-wide form of the wave tiers (aln_ksw_rows.hpp).
//   in : sq / qb      target and query base (nt4 codes, 4 = ambiguous)
//        xl vl x2l    x, v, x2 of the left neighbour (same query row, previous target column)
//        ut yt y2t    u, y, y2 of the upper neighbour (same target column, previous query row)
//   out: un vn        H(t,q) - H(t-1,q) and H(t,q) - H(t,q-1)
//        xn yn x2n y2n, d = traceback byte (bits 0-2: which of the five won; 0x08 / 0x10 / 0x20 / 0x40: x / y / x2 / y2 continue)
#pragma once
#include <stdint.h>

#include "../device/pmx_math.h"   // PMX_HD (the host build of the unit tests includes this header too)

namespace pmx {
namespace aln {

struct KswCellParams {
    int q, q2, qe, qe2;          // gap open costs, open + extend
    int sc_mch, sc_mis, sc_N;
};

template <bool RIGHT>
PMX_HD void ksw_cell(const KswCellParams& P, int sq, int qb, int xl, int vl, int x2l, int ut, int yt, int y2t, int& un, int& vn, int& xn,
                                         int& yn, int& x2n, int& y2n, uint32_t& d) {
    int z = sq == qb ? P.sc_mch : P.sc_mis;
    if (sq == 4 || qb == 4) z = P.sc_N;
    int a = xl + vl, b = yt + ut, a2 = x2l + vl, b2 = y2t + ut;
    if (!RIGHT) {   // gaps left-aligned: the first of equal candidates wins (:228-256)
        d = a > z ? 1u : 0u;
        z = z > a ? z : a;
        d = b > z ? 2u : d;
        z = z > b ? z : b;
        d = a2 > z ? 3u : d;
        z = z > a2 ? z : a2;
        d = b2 > z ? 4u : d;
        z = z > b2 ? z : b2;
    } else {        // gaps right-aligned: the last one does (:283-311)
        d = z > a ? 0u : 1u;
        z = z > a ? z : a;
        d = z > b ? d : 2u;
        z = z > b ? z : b;
        d = z > a2 ? d : 3u;
        z = z > a2 ? z : a2;
        d = z > b2 ? d : 4u;
        z = z > b2 ? z : b2;
    }
    z = z < P.sc_mch ? z : P.sc_mch;
    un = z - vl;
    vn = z - ut;
    int tmp = z - P.q;
    a -= tmp;
    b -= tmp;
    tmp = z - P.q2;
    a2 -= tmp;
    b2 -= tmp;
    if (!RIGHT) {
        xn = (a > 0 ? a : 0) - P.qe;    d |= a > 0 ? 0x08u : 0u;
        yn = (b > 0 ? b : 0) - P.qe;    d |= b > 0 ? 0x10u : 0u;
        x2n = (a2 > 0 ? a2 : 0) - P.qe2; d |= a2 > 0 ? 0x20u : 0u;
        y2n = (b2 > 0 ? b2 : 0) - P.qe2; d |= b2 > 0 ? 0x40u : 0u;
    } else {
        xn = (0 > a ? 0 : a) - P.qe;    d |= 0 > a ? 0u : 0x08u;
        yn = (0 > b ? 0 : b) - P.qe;    d |= 0 > b ? 0u : 0x10u;
        x2n = (0 > a2 ? 0 : a2) - P.qe2; d |= 0 > a2 ? 0u : 0x20u;
        y2n = (0 > b2 ? 0 : b2) - P.qe2; d |= 0 > b2 ? 0u : 0x40u;
    }
}

}  // namespace aln
}  // namespace pmx
