// ALIGN stage, long reads: the second chaining pass of mm_map_frag (map.c:296-305) -- when the first chain of a
// long read leaves more than rmq_rescue_size bases of it uncovered, or spans more than rmq_rescue_ratio of it while
// other chains exist, the anchors that made it into chains are sorted again and chained with mg_lchain_rmq
// (lchain.c:232-369) under the long bandwidth, so that chains across large gaps join.
//
// mg_lchain_rmq finds the best predecessor of an anchor with a range-minimum query on a balanced tree of the anchors
// in range (krmq.h: an AVL tree keyed by (query position, anchor index) whose nodes know the member of least
// priority below them).  Where two members of a range have the SAME priority the one returned depends on the shape
// of the tree, so the tree is restated operation for operation: the same insertions, deletions and rotations leave
// the same shape.  What is different is the storage -- the reference allocates pointer nodes from a pool; here a node
// IS its anchor (every anchor enters a tree at most once), the links are 16-bit anchor indices packed with the
// subtree size into one 8-byte word per tree, and both trees (outer range max_dist, inner range max_dist_inner) live
// in the two anchor-sized scratch arrays that are idle during a fill (W.z, W.a2):
//   z[j].x  = priority (double bits), shared by the two trees (the reference copies the node)
//   z[j].y  = links of the outer tree      a2[j].x = links of the inner tree      a2[j].y = the two balance factors
// The fill runs sequentially (in the wave kernels uniformly on the 64 lanes, like chain_dp's wide fill): a few
// thousand anchors x log n steps for the reads that take this path at all.
#pragma once
#include "aln_chain.hpp"

namespace pmx {
namespace aln {

namespace rmq {

constexpr uint32_t kNil = 0xffffu;    // no node
constexpr uint32_t kFake = 0xfffeu;   // krmq_erase's stand-in above the root (krmq.h:226)
constexpr int kDepth = 40;            // AVL height for < 65,535 nodes is below 24

struct Links {
    uint32_t c[2], s, size;           // children, member of least priority in the subtree, subtree size
};

struct Tree {
    Ptr<A128> z, a2;
    Ptr<A128> a;                      // the anchors (keys)
    int which;                        // 0 outer, 1 inner
    uint32_t root;

    PMX_HD Links get(uint32_t j) const {
        const uint64_t w = which ? a2[j].x : z[j].y;
        Links L;
        L.c[0] = (uint32_t)(w & 0xffffu); L.c[1] = (uint32_t)(w >> 16 & 0xffffu);
        L.s = (uint32_t)(w >> 32 & 0xffffu); L.size = (uint32_t)(w >> 48);
        return L;
    }
    PMX_HD void put(uint32_t j, const Links& L) {
        const uint64_t w = (uint64_t)L.c[0] | (uint64_t)L.c[1] << 16 | (uint64_t)L.s << 32 | (uint64_t)L.size << 48;
        if (which) a2[j].x = w; else z[j].y = w;
    }
    PMX_HD int bal(uint32_t j) const { return (int)(int8_t)(a2[j].y >> (8 * which) & 0xffu); }
    PMX_HD void set_bal(uint32_t j, int b) {
        const uint64_t w = a2[j].y;
        a2[j].y = (w & ~(0xffULL << (8 * which))) | (uint64_t)(uint8_t)(int8_t)b << (8 * which);
    }
    PMX_HD double pri(uint32_t j) const {
        const uint64_t w = z[j].x;
        double d;
        memcpy(&d, &w, 8);
        return d;
    }
    PMX_HD uint32_t child(uint32_t j, int d) const { return j == kFake ? (d == 0 ? root : kNil) : get(j).c[d]; }
    PMX_HD void set_child(uint32_t j, int d, uint32_t v) {
        if (j == kFake) { if (d == 0) root = v; return; }
        Links L = get(j);
        L.c[d] = v;
        put(j, L);
    }
    PMX_HD uint32_t size_of(uint32_t j) const { return j == kNil ? 0u : get(j).size; }
    // lc_elem_cmp (lchain.c:226) of a key (y, i) against node p
    PMX_HD int cmp(int32_t y, int64_t i, uint32_t p) const {
        const int32_t py = (int32_t)a[p].y;
        return y < py ? -1 : y > py ? 1 : (int)(i > (int64_t)p) - (int)(i < (int64_t)p);
    }
    PMX_HD bool lt2(uint32_t x, uint32_t y) const { return pri(x) < pri(y); }
    // krmq_update_min (krmq.h:137-140): p's least member from itself and its children q, r
    PMX_HD void update_min(uint32_t p, uint32_t q, uint32_t r) {
        if (p == kFake) return;   // the stand-in's fields are thrown away
        Links L = get(p);
        uint32_t s = p;
        if (q != kNil) { const uint32_t qs = get(q).s; if (!lt2(p, qs)) s = qs; }
        if (r != kNil) { const uint32_t rs = get(r).s; if (!lt2(s, rs)) s = rs; }
        L.s = s;
        put(p, L);
    }
    // krmq_rotate1 (krmq.h:142-153): (a,(b,c)q)p => ((a,b)p,c)q for dir = 0
    PMX_HD uint32_t rotate1(uint32_t p, int dir) {
        const int opp = 1 - dir;
        Links P = get(p);
        const uint32_t q = P.c[opp], s = P.s;
        Links Q = get(q);
        const uint32_t size_p = P.size;
        P.size -= Q.size - size_of(Q.c[dir]);
        Q.size = size_p;
        put(p, P);
        update_min(p, P.c[dir], Q.c[dir]);
        P = get(p);
        Q.s = s;
        P.c[opp] = Q.c[dir];
        Q.c[dir] = p;
        put(p, P);
        put(q, Q);
        return q;
    }
    // krmq_rotate2 (krmq.h:155-177): (a,((b,c)r,d)q)p => ((a,b)p,(c,d)q)r for dir = 0
    PMX_HD uint32_t rotate2(uint32_t p, int dir) {
        const int opp = 1 - dir;
        Links P = get(p);
        const uint32_t q = P.c[opp];
        Links Q = get(q);
        const uint32_t r = Q.c[dir], s = P.s;
        Links R = get(r);
        const uint32_t size_x_dir = size_of(R.c[dir]);
        R.size = P.size;
        P.size -= Q.size - size_x_dir;
        Q.size -= size_x_dir + 1;
        put(p, P); put(q, Q); put(r, R);
        update_min(p, P.c[dir], R.c[dir]);
        update_min(q, Q.c[opp], R.c[opp]);
        P = get(p); Q = get(q);
        R.s = s;
        P.c[opp] = R.c[dir];
        R.c[dir] = p;
        Q.c[dir] = R.c[opp];
        R.c[opp] = q;
        put(p, P); put(q, Q); put(r, R);
        const int b1 = dir == 0 ? +1 : -1, br = bal(r);
        if (br == b1) { set_bal(q, 0); set_bal(p, -b1); }
        else if (br == 0) { set_bal(q, 0); set_bal(p, 0); }
        else { set_bal(q, b1); set_bal(p, 0); }
        set_bal(r, 0);
        return r;
    }
    // krmq_insert (krmq.h:179-224); the key of x is never in the tree already (anchor indices are distinct)
    PMX_HDN void insert(uint32_t x) {
        uint8_t stack[kDepth];
        uint32_t path[kDepth];
        uint32_t bp = root, bq = kNil, p, q;
        int which_ = 0, top = 0, path_len = 0;
        const int32_t xy = (int32_t)a[x].y;
        for (p = bp, q = bq; p != kNil; q = p, p = get(p).c[which_]) {
            const int c_ = cmp(xy, (int64_t)x, p);
            if (bal(p) != 0) { bq = q; bp = p; top = 0; }
            stack[top++] = (uint8_t)(which_ = (c_ > 0));
            path[path_len++] = p;
        }
        Links X;
        X.c[0] = X.c[1] = kNil; X.s = x; X.size = 1;
        put(x, X);
        set_bal(x, 0);
        if (q == kNil) root = x;
        else set_child(q, which_, x);
        if (bp == kNil) return;
        for (int i = 0; i < path_len; ++i) { Links L = get(path[i]); ++L.size; put(path[i], L); }
        for (int i = path_len - 1; i >= 0; --i) {
            const Links L = get(path[i]);
            update_min(path[i], L.c[0], L.c[1]);
            if (get(path[i]).s != x) break;
        }
        top = 0;
        for (p = bp; p != x; p = get(p).c[stack[top]], ++top)
            set_bal(p, bal(p) + (stack[top] == 0 ? -1 : +1));
        const int bb = bal(bp);
        if (bb > -2 && bb < 2) return;
        which_ = bb < 0;
        const int b1 = which_ == 0 ? +1 : -1;
        q = get(bp).c[1 - which_];
        uint32_t r;
        if (bal(q) == b1) {
            r = rotate1(bp, which_);
            set_bal(q, 0);
            set_bal(bp, 0);
        } else r = rotate2(bp, which_);
        if (bq == kNil) root = r;
        else set_child(bq, bp != get(bq).c[0], r);
    }
    // krmq_find + krmq_erase (krmq.h:226-311) of the node with key (y, i); absent: nothing happens (the reference's
    // krmq_find returns NULL first, lchain.c:284-288)
    PMX_HDN void erase(int32_t y, int64_t i_key) {
        if (root == kNil) return;
        uint32_t path[kDepth];
        uint8_t dir[kDepth];
        int d = 0;
        uint32_t p = kFake;
        for (int c_ = -1; c_; c_ = cmp(y, i_key, p)) {
            const int w = c_ > 0;
            dir[d] = (uint8_t)w;
            path[d++] = p;
            p = child(p, w);
            if (p == kNil) return;
        }
        for (int i = 1; i < d; ++i) { Links L = get(path[i]); --L.size; put(path[i], L); }
        const Links P = get(p);
        if (P.c[1] == kNil) {
            set_child(path[d - 1], dir[d - 1], P.c[0]);
        } else {
            uint32_t q = P.c[1];
            Links Q = get(q);
            if (Q.c[0] == kNil) {
                Q.c[0] = P.c[0];
                Q.size = P.size - 1;
                put(q, Q);
                set_bal(q, bal(p));
                set_child(path[d - 1], dir[d - 1], q);
                path[d] = q; dir[d++] = 1;
            } else {
                uint32_t r;
                const int e = d++;
                for (;;) {
                    dir[d] = 0;
                    path[d++] = q;
                    r = get(q).c[0];
                    if (get(r).c[0] == kNil) break;
                    q = r;
                }
                Links R = get(r);
                Q = get(q);
                R.c[0] = P.c[0];
                Q.c[0] = R.c[1];
                R.c[1] = P.c[1];
                put(q, Q);
                put(r, R);
                set_bal(r, bal(p));
                set_child(path[e - 1], dir[e - 1], r);
                path[e] = r; dir[e] = 1;
                for (int i = e + 1; i < d; ++i) { Links L = get(path[i]); --L.size; put(path[i], L); }
                R = get(r);
                R.size = P.size - 1;
                put(r, R);
            }
        }
        for (int i = d - 1; i >= 0; --i) {
            if (path[i] == kFake) continue;
            const Links L = get(path[i]);
            update_min(path[i], L.c[0], L.c[1]);
        }
        while (--d > 0) {
            const uint32_t q = path[d];
            int b1 = 1, b2 = 2;
            const int w = dir[d], other = 1 - w;
            if (w) { b1 = -b1; b2 = -b2; }
            const int nb = bal(q) + b1;
            set_bal(q, nb);
            if (nb == b1) break;
            else if (nb == b2) {
                const uint32_t r = get(q).c[other];
                if (bal(r) == -b1) {
                    set_child(path[d - 1], dir[d - 1], rotate2(q, w));
                } else {
                    set_child(path[d - 1], dir[d - 1], rotate1(q, w));
                    if (bal(r) == 0) {
                        set_bal(r, -b1);
                        set_bal(q, b1);
                        break;
                    } else { set_bal(r, 0); set_bal(q, 0); }
                }
            }
        }
    }
    // krmq_rmq (krmq.h:111-134): the member of least priority with lo <= key <= up (closed), kNil without one
    PMX_HDN uint32_t range_min(int32_t lo_y, int64_t lo_i, int32_t up_y, int64_t up_i) const {
        if (root == kNil) return kNil;
        uint32_t path0[kDepth], path1[kDepth];
        int8_t pc0[kDepth], pc1[kDepth];
        int n0 = 0, n1 = 0;
        for (uint32_t p = root; p != kNil;) {
            const int c_ = cmp(lo_y, lo_i, p);
            path0[n0] = p; pc0[n0++] = (int8_t)c_;
            if (c_ < 0) p = get(p).c[0];
            else if (c_ > 0) p = get(p).c[1];
            else break;
        }
        for (uint32_t p = root; p != kNil;) {
            const int c_ = cmp(up_y, up_i, p);
            path1[n1] = p; pc1[n1++] = (int8_t)c_;
            if (c_ < 0) p = get(p).c[0];
            else if (c_ > 0) p = get(p).c[1];
            else break;
        }
        int i;
        for (i = 0; i < n0 && i < n1; ++i)
            if (path0[i] == path1[i] && pc0[i] <= 0 && pc1[i] >= 0) break;
        if (i == n0 || i == n1) return kNil;
        const int lca = i;
        uint32_t mn = path0[lca];
        for (i = lca + 1; i < n0; ++i) {
            if (pc0[i] <= 0) {
                if (lt2(path0[i], mn)) mn = path0[i];
                const uint32_t rc = get(path0[i]).c[1];
                if (rc != kNil) { const uint32_t rs = get(rc).s; if (lt2(rs, mn)) mn = rs; }
            }
        }
        for (i = lca + 1; i < n1; ++i) {
            if (pc1[i] >= 0) {
                if (lt2(path1[i], mn)) mn = path1[i];
                const uint32_t lc = get(path1[i]).c[0];
                if (lc != kNil) { const uint32_t ls = get(lc).s; if (lt2(ls, mn)) mn = ls; }
            }
        }
        return mn;
    }
    // krmq_interval (krmq.h:96-108), lower bound only: the last member with key <= (y, i)
    PMX_HD uint32_t lower(int32_t y, int64_t i_key) const {
        uint32_t p = root, l = kNil;
        while (p != kNil) {
            const int c_ = cmp(y, i_key, p);
            if (c_ < 0) p = get(p).c[0];
            else if (c_ > 0) { l = p; p = get(p).c[1]; }
            else { l = p; break; }
        }
        return l;
    }
};

// krmq_itr_find on a member + krmq_itr_prev (krmq.h:344-381): in-order walk towards smaller keys
struct Iter {
    uint32_t stack[kDepth];
    int top;   // index of the top entry, -1 = exhausted
    PMX_HD void find(const Tree& T, uint32_t x) {
        const int32_t y = (int32_t)T.a[x].y;
        top = -1;
        for (uint32_t p = T.root; p != kNil;) {
            stack[++top] = p;
            const int c_ = T.cmp(y, (int64_t)x, p);
            if (c_ < 0) p = T.get(p).c[0];
            else if (c_ > 0) p = T.get(p).c[1];
            else break;
        }
    }
    PMX_HD uint32_t at() const { return top < 0 ? kNil : stack[top]; }
    PMX_HD bool prev(const Tree& T) {
        if (top < 0) return false;
        uint32_t p = T.get(stack[top]).c[0];
        if (p != kNil) {
            for (; p != kNil; p = T.get(p).c[1]) stack[++top] = p;
            return true;
        }
        do {
            p = stack[top--];
        } while (top >= 0 && p == T.get(stack[top]).c[0]);
        return top >= 0;
    }
};

// comput_sc_simple (lchain.c:232-248)
PMX_HD int32_t score_simple(const A128 ai, const A128 aj, float chn_pen_gap, float chn_pen_skip, int32_t* exact, int32_t* width) {
    const int32_t dq = (int32_t)ai.y - (int32_t)aj.y;
    const int32_t dr = (int32_t)(ai.x - aj.x);
    const int32_t dd = dr > dq ? dr - dq : dq - dr;
    *width = dd;
    const int32_t dg = dr < dq ? dr : dq;
    const int32_t q_span = (int32_t)(aj.y >> 32 & 0xff);
    int32_t sc = q_span < dg ? q_span : dg;
    if (exact) *exact = (dd == 0 && dg <= q_span);
    if (dd || dq > q_span) {
        const float lin_pen = chn_pen_gap * (float)dd + chn_pen_skip * (float)dg;
        const float log_pen = dd >= 1 ? mg_log2f((float)(dd + 1)) : 0.0f;
        sc -= (int)(lin_pen + .5f * log_pen);
    }
    return sc;
}

}  // namespace rmq

// mg_lchain_rmq (lchain.c:250-369).  In: W.a[0..n_a) sorted anchors.  Out: as chain_dp.
PMX_HDN void chain_rmq(Work& W, int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc,
                       float chn_pen_gap, float chn_pen_skip) {
    const int64_t n = W.n_a;
    PMX_LDS(&W);
    W.n_u = 0;
    if (n == 0) return;
    if (n >= 65534) {   // 16-bit links
        W.status |= PMX_ST_UNSUPPORTED;
        W.n_a = 0;
        return;
    }
    Ptr<A128> a = W.a; PMX_LDS(a);
    Ptr<ChainCell> c = W.cc; PMX_LDS(c);
    const int32_t max_drop = bw;
    if (max_dist < bw) max_dist = bw;
    if (max_dist_inner <= 0 || max_dist_inner >= max_dist) max_dist_inner = 0;
    rmq::Tree T, TI;
    T.z = TI.z = W.z; T.a2 = TI.a2 = W.a2; T.a = TI.a = a;
    T.which = 0; TI.which = 1;
    T.root = TI.root = rmq::kNil;
    for (int64_t i = 0; i < n; ++i) c[i].t = 0;
    wave_sync();
    int64_t i0 = 0, st = 0, st_inner = 0;
    for (int64_t i = 0; i < n; ++i) {
        const A128 ai = a[i];
        int64_t max_j = -1;
        const int32_t q_span = (int32_t)(ai.y >> 32 & 0xff);
        int32_t max_f = q_span;
        // add in-range anchors (lchain.c:271-283)
        if (i0 < i && a[i0].x != ai.x) {
            for (int64_t j = i0; j < i; ++j) {
                const A128 aj = a[j];
                const double half_gap = 0.5 * chn_pen_gap;
                const double diag = half_gap * ((int32_t)aj.x + (int32_t)aj.y);
                const double pri = -(c[j].f + diag);
                uint64_t bits;
                memcpy(&bits, &pri, 8);
                W.z[j].x = bits;
                T.insert((uint32_t)j);
                if (max_dist_inner > 0) TI.insert((uint32_t)j);
            }
            i0 = i;
        }
        // get rid of active chains out of range (lchain.c:284-301)
        while (st < i && (ai.x >> 32 != a[st].x >> 32 || ai.x > a[st].x + (uint64_t)max_dist || T.size_of(T.root) > (uint32_t)cap_rmq_size)) {
            T.erase((int32_t)a[st].y, st);
            ++st;
        }
        if (max_dist_inner > 0) {
            while (st_inner < i && (ai.x >> 32 != a[st_inner].x >> 32 || ai.x > a[st_inner].x + (uint64_t)max_dist_inner ||
                                    TI.size_of(TI.root) > (uint32_t)cap_rmq_size)) {
                TI.erase((int32_t)a[st_inner].y, st_inner);
                ++st_inner;
            }
        }
        // RMQ (lchain.c:302-338)
        const uint32_t q = T.range_min((int32_t)ai.y - max_dist, (int64_t)INT32_MAX, (int32_t)ai.y, 0);
        if (q != rmq::kNil) {
            int32_t exact, width, n_skip = 0;
            int64_t j = (int64_t)q;
            int32_t sc = c[j].f + rmq::score_simple(ai, a[j], chn_pen_gap, chn_pen_skip, &exact, &width);
            if (width <= bw && sc > max_f) { max_f = sc; max_j = j; }
            if (!exact && TI.root != rmq::kNil && (int32_t)ai.y > 0) {
                const uint32_t lo = TI.lower((int32_t)ai.y - 1, n);
                if (lo != rmq::kNil) {
                    rmq::Iter it;
                    it.find(TI, lo);
                    uint32_t qq;
                    while ((qq = it.at()) != rmq::kNil) {
                        if ((int32_t)a[qq].y < (int32_t)ai.y - max_dist_inner) break;
                        j = (int64_t)qq;
                        sc = c[j].f + rmq::score_simple(ai, a[j], chn_pen_gap, chn_pen_skip, nullptr, &width);
                        if (width <= bw) {
                            if (sc > max_f) {
                                max_f = sc; max_j = j;
                                if (n_skip > 0) --n_skip;
                            } else if (c[j].t == (int32_t)i) {
                                if (++n_skip > max_chn_skip) break;
                            }
                            const int32_t pj = c[j].p;
                            if (pj >= 0) c[pj].t = (int32_t)i;
                        }
                        if (!it.prev(TI)) break;
                    }
                }
            }
        }
        // set max (lchain.c:339-343)
        const int32_t vm = max_j >= 0 ? c[max_j].v : 0;
        c[i].f = max_f;
        c[i].p = (int32_t)max_j;
        c[i].v = max_j >= 0 && vm > max_f ? vm : max_f;
    }
    wave_sync();
    chain_finish(W, n, min_cnt, min_sc, max_drop);
}

}  // namespace aln
}  // namespace pmx
