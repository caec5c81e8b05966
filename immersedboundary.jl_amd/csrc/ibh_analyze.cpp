// libibhip: host-side analysis that recovers the block structure of a Partition from the
// reference's own data (part.domain global ids + face lists) and classifies every block side
// for the block fast path.  Nothing is assumed: a side gets a fast class only if the faces the
// reference actually registered for its 8 boundary cells are exactly the faces that class
// implies (this also covers the reference's search-radius rule that can drop faces at >2:1
// jumps, ImmersedBoundary.jl:85,114-116).  Everything else falls back to the face-list kernels.
//
// Cell numbering inside a block is x-fastest (mesher.jl:1064-1112): local = i + 8*j.
#include <algorithm>
#include <array>
#include <unordered_map>

#include "ibh_common.h"

namespace {

inline int pos_own(int s, int t) { return s == 0 ? 8 * t : s == 1 ? 7 + 8 * t : s == 2 ? t : t + 56; }
inline int pos_opp(int s, int tt) { return s == 0 ? 7 + 8 * tt : s == 1 ? 8 * tt : s == 2 ? tt + 56 : tt; }
inline int tang_of_pos(int s, int pos) { return s < 2 ? pos / 8 : pos % 8; }
inline bool on_opp_edge(int s, int pos) {
    return s == 0 ? (pos % 8 == 7) : s == 1 ? (pos % 8 == 0) : s == 2 ? (pos / 8 == 7) : (pos / 8 == 0);
}

}  // namespace

void ibh_analyze_blocks2(const HostPartView& v, std::vector<BlockDesc2>& blocks, std::vector<int32_t>& irr,
                         int64_t* info, const int32_t* image_in_domain, int32_t n_image, int32_t* n_phase1,
                         std::vector<int32_t>& htab, std::vector<int32_t>& etab, std::vector<char>& fus,
                         std::vector<char>& needg, std::vector<int32_t>& dtab) {
    const int32_t nc = v.nc;
    const int NPB = 64;
    const float* hx = v.spacing;
    const float* hy = v.spacing + nc;
    auto gid = [&](int32_t c) { return (int64_t)v.domain[c] - v.index_base; };

    // 1. full blocks: 64 consecutive local cells = one whole global block, uniform spacing
    std::unordered_map<int64_t, int32_t> blockbase;
    std::vector<int32_t> bases;
    std::vector<char> in_full(nc, 0);
    for (int32_t c = 0; c + NPB <= nc;) {
        int64_t g = gid(c);
        bool ok = (g % NPB == 0) && gid(c + NPB - 1) == g + NPB - 1;
        if (ok)
            for (int k = 1; k < NPB && ok; ++k)
                ok = gid(c + k) == g + k && hx[c + k] == hx[c] && hy[c + k] == hy[c];
        if (ok) {
            blockbase[g / NPB] = c;
            bases.push_back(c);
            for (int k = 0; k < NPB; ++k) in_full[c + k] = 1;
            c += NPB;
        } else {
            ++c;
        }
    }

    auto single = [&](const std::vector<int32_t>& off, const std::vector<int32_t>& idx, int32_t c) -> int32_t {
        return (off[c + 1] - off[c] == 1) ? idx[off[c]] : -1;
    };

    std::vector<char> cell_irr(nc, 0);
    std::vector<std::array<int32_t, 64>> halos;  // halo cell ids per block, in the order blocks are found
    for (int32_t c = 0; c < nc; ++c) cell_irr[c] = !in_full[c];
    int64_t counts[5] = {0, 0, 0, 0, 0};

    for (int32_t base : bases) {
        // 2. interior faces must be the implicit ones
        bool good = true;
        for (int pos = 0; pos < NPB && good; ++pos) {
            int i = pos % 8, j = pos / 8;
            int32_t c = base + pos;
            for (int d = 0; d < 2 && good; ++d) {
                int step = d == 0 ? 1 : 8;
                int ij = d == 0 ? i : j;
                if (ij < 7) {
                    int32_t f = single(v.roff[d], v.ridx[d], c);
                    good = f >= 0 && v.owners[d][f] == c && v.neighbors[d][f] == c + step;
                }
                if (good && ij > 0) {
                    int32_t f = single(v.loff[d], v.lidx[d], c);
                    good = f >= 0 && v.owners[d][f] == c - step && v.neighbors[d][f] == c;
                }
            }
        }
        if (!good) {
            for (int k = 0; k < NPB; ++k) cell_irr[base + k] = 1;
            continue;
        }
        BlockDesc2 b;
        b.base = base;
        b.h[0] = hx[base];
        b.h[1] = hy[base];
        b.rh[0] = 1.0f / b.h[0];
        b.rh[1] = 1.0f / b.h[1];
        // 3. sides.  The class of a side follows from the geometry of the faces the reference registered (global
        //    ids give block and position of every neighbour cell); the neighbour block itself need not be complete
        //    in this partition (skirt): the halo table takes the local ids of the cells that are there.
        std::array<int32_t, 64> hal;
        for (int s = 0; s < 4; ++s) {
            const int d = s / 2;
            const bool low = (s % 2) == 0;
            const float* h = d == 0 ? hx : hy;
            const float hc = h[base];
            const std::vector<int32_t>& off = low ? v.loff[d] : v.roff[d];
            const std::vector<int32_t>& idx = low ? v.lidx[d] : v.ridx[d];
            int type = -1;
            int64_t nbg[2] = {-1, -1};   // global block id of the neighbour block(s)
            bool nbfull[2] = {true, true};
            int sub = -1;
            bool ok = true;
            for (int t = 0; t < 8 && ok; ++t) {
                int32_t c = base + pos_own(s, t);
                int nfc = off[c + 1] - off[c];
                int32_t other[2] = {-1, -1};
                if (nfc < 1 || nfc > 2) { ok = false; break; }
                for (int k = 0; k < nfc; ++k) {
                    int32_t f = idx[off[c] + k];
                    int32_t me = low ? v.neighbors[d][f] : v.owners[d][f];
                    if (me != c) ok = false;
                    other[k] = low ? v.owners[d][f] : v.neighbors[d][f];
                }
                if (!ok) break;
                int ty;
                int32_t* slot = &hal[(s * 8 + t) * 2];
                if (nfc == 1 && other[0] == c) {
                    ty = SIDE_MIRROR;
                    slot[0] = slot[1] = c;
                } else if (nfc == 1) {
                    int32_t o = other[0];
                    int64_t g = gid(o);
                    int64_t gb = g / NPB;
                    int pos = (int)(g % NPB);
                    if (!on_opp_edge(s, pos)) { ok = false; break; }
                    int tt = tang_of_pos(s, pos);
                    if (h[o] == hc) {
                        ty = SIDE_SAME;
                        if (tt != t) { ok = false; break; }
                        if (t == 0) nbg[0] = gb; else if (nbg[0] != gb) { ok = false; break; }
                    } else if (h[o] == hc * 2.0f) {
                        ty = SIDE_COARSE;
                        int q = (tt - t / 2);
                        if (q != 0 && q != 4) { ok = false; break; }
                        q /= 4;
                        if (t == 0) { nbg[0] = gb; sub = q; } else if (nbg[0] != gb || sub != q) { ok = false; break; }
                    } else { ok = false; break; }
                    nbfull[0] = nbfull[0] && in_full[o];
                    slot[0] = slot[1] = o;
                } else {
                    ty = SIDE_FINE;
                    int64_t gbk = -1;
                    bool seen[2] = {false, false};
                    for (int k = 0; k < 2; ++k) {
                        int32_t o = other[k];
                        if (o == c || h[o] != hc * 0.5f) { ok = false; break; }
                        int64_t g = gid(o);
                        int64_t gb = g / NPB;
                        int pos = (int)(g % NPB);
                        if (!on_opp_edge(s, pos)) { ok = false; break; }
                        int tt = tang_of_pos(s, pos) - 2 * (t & 3);
                        if (tt != 0 && tt != 1) { ok = false; break; }
                        if (seen[tt]) { ok = false; break; }
                        seen[tt] = true;
                        slot[tt] = o;
                        if (gbk < 0) gbk = gb; else if (gbk != gb) { ok = false; break; }
                        nbfull[t >> 2] = nbfull[t >> 2] && in_full[o];
                    }
                    if (!ok) break;
                    int kb = t >> 2;
                    if ((t & 3) == 0) nbg[kb] = gbk; else if (nbg[kb] != gbk) { ok = false; break; }
                }
                if (t == 0) type = ty; else if (type != ty) { ok = false; break; }
            }
            if (!ok) {
                type = SIDE_GENERAL;
                for (int t = 0; t < 8; ++t) hal[(s * 8 + t) * 2] = hal[(s * 8 + t) * 2 + 1] = base + pos_own(s, t);
            }
            b.type[s] = type;
            for (int k = 0; k < 2; ++k) {  // local base of the neighbour block, -1 when it is not complete here
                b.nb[s][k] = -1;
                if (type != SIDE_GENERAL && nbg[k] >= 0 && nbfull[k]) {
                    auto it = blockbase.find(nbg[k]);
                    if (it != blockbase.end()) b.nb[s][k] = it->second;
                }
            }
            b.sub[s] = sub < 0 ? 0 : sub;
            b.q[s] = type == SIDE_COARSE ? (1.0f / 3.0f) : type == SIDE_FINE ? (2.0f / 3.0f) : 0.5f;
            b.rt[s] = type == SIDE_COARSE ? 2.0f : type == SIDE_FINE ? 0.5f : 1.0f;
            counts[type]++;
            if (type == SIDE_GENERAL)
                for (int t = 0; t < 8; ++t) cell_irr[base + pos_own(s, t)] = 1;
        }
        halos.push_back(hal);
        blocks.push_back(b);
    }
    for (int32_t c = 0; c < nc; ++c)
        if (cell_irr[c]) irr.push_back(c);

    // 4. overlap phases: a cell's gradients depend on the skirt if the cell or a face
    //    neighbour is a skirt cell (not in `image`).  A1 = blocks without such a cell (pass A can run before
    //    the halo exchange lands); B1 = A1 blocks whose side neighbours are all A1 (pass B as well).
    //    Block table order: [B1 | A1 only | rest].
    n_phase1[0] = n_phase1[1] = 0;
    {
        std::vector<char> is_image(nc, 0), dep(nc, 0);
        for (int32_t k = 0; k < n_image; ++k) is_image[image_in_domain[k] - v.index_base] = 1;
        for (int32_t c = 0; c < nc; ++c) dep[c] = !is_image[c];
        for (int d = 0; d < v.nd; ++d)
            for (size_t f = 0; f < v.owners[d].size(); ++f) {
                int32_t o = v.owners[d][f], n = v.neighbors[d][f];
                if (!is_image[o]) dep[n] = 1;
                if (!is_image[n]) dep[o] = 1;
            }
        std::unordered_map<int32_t, int> base2a1;
        std::vector<char> a1(blocks.size(), 0), b1(blocks.size(), 0);
        for (size_t b = 0; b < blocks.size(); ++b) {
            bool ok = true;
            for (int k = 0; k < NPB && ok; ++k) ok = !dep[blocks[b].base + k] && !cell_irr[blocks[b].base + k];
            a1[b] = ok;
            base2a1[blocks[b].base] = ok;
        }
        for (size_t b = 0; b < blocks.size(); ++b) {
            bool ok = a1[b];
            for (int s = 0; s < 4 && ok; ++s) {
                int ty = blocks[b].type[s];
                if (ty == SIDE_MIRROR) continue;
                if (ty == SIDE_GENERAL) { ok = false; break; }
                int nn = ty == SIDE_FINE ? 2 : 1;
                for (int k = 0; k < nn && ok; ++k) {
                    auto it = base2a1.find(blocks[b].nb[s][k]);
                    ok = it != base2a1.end() && it->second;
                }
            }
            b1[b] = ok;
        }
        std::vector<BlockDesc2> ordered;
        std::vector<std::array<int32_t, 64>> hordered;
        ordered.reserve(blocks.size());
        hordered.reserve(blocks.size());
        auto take = [&](size_t b) { ordered.push_back(blocks[b]); hordered.push_back(halos[b]); };
        for (size_t b = 0; b < blocks.size(); ++b) if (b1[b]) take(b);
        n_phase1[1] = (int32_t)ordered.size();
        for (size_t b = 0; b < blocks.size(); ++b) if (a1[b] && !b1[b]) take(b);
        n_phase1[0] = (int32_t)ordered.size();
        for (size_t b = 0; b < blocks.size(); ++b) if (!a1[b]) take(b);
        blocks.swap(ordered);
        halos.swap(hordered);
    }
    // 5. halo cell table in the final block order: the local ids found in step 3 (for complete neighbour blocks
    //    these are base + position; MIRROR and GENERAL sides name the boundary cell itself)
    htab.resize(blocks.size() * 64);
    for (size_t bi = 0; bi < blocks.size(); ++bi)
        for (int k = 0; k < 64; ++k) htab[bi * 64 + k] = halos[bi][k];
    // 6. single-kernel sweep (blk2::sweep_adv / sweep_euler): the wave of a block also computes slope + sensor of
    //    its 64 halo cells.  For a halo cell h across side s (normal dim d, tangential dim td) that takes
    //      (a) its faces towards this block: this block's boundary cells (verified in step 3, re-checked from h's side),
    //      (b) the cell one step deeper: h's single far face along d, same level       -> dtab[blk][slot]
    //      (c) its lateral neighbours along td: the adjacent halo cells of the side,
    //      (d) at the two ends of the side whatever h's outer lateral faces name (one or two cells, h itself on a
    //          mirror face)                                                            -> etab[blk][(s*2 + end)*2 + k]
    //    all read off the face lists of h itself, so the neighbour block may be a skirt fragment.  A block is
    //    eligible when every halo cell passes.  `dt` of the descriptor: -1 when the deeper cells are simply
    //    halo id -/+ 1 (x sides) or -/+ 8 (y sides) -- complete neighbour blocks -- else the row of dtab to read.
    etab.assign(blocks.size() * 16, 0);
    fus.assign(blocks.size(), 0);
    dtab.clear();
    {
        auto faces_of = [&](int32_t c, int dim, bool right, int32_t* out) -> int {  // cells across the faces of c
            const std::vector<int32_t>& off = right ? v.roff[dim] : v.loff[dim];
            const std::vector<int32_t>& idx = right ? v.ridx[dim] : v.lidx[dim];
            const int n = off[c + 1] - off[c];
            if (n < 1 || n > 2) return -1;
            for (int k = 0; k < n; ++k) {
                const int32_t f = idx[off[c] + k];
                const int32_t o = v.owners[dim][f], nn = v.neighbors[dim][f];
                if ((right ? o : nn) != c) return -1;
                out[k] = right ? nn : o;
            }
            return n;
        };
        int64_t nfus = 0;
        for (size_t bi = 0; bi < blocks.size(); ++bi) {
            BlockDesc2& b = blocks[bi];
            b.dt = -1;
            bool ok = true;
            for (int k = 0; k < NPB && ok; ++k) ok = !cell_irr[b.base + k];
            int32_t deep[64];
            bool arithmetic = true;
            for (int s = 0; s < 4; ++s) {
                int32_t* e = &etab[bi * 16 + s * 4];
                e[0] = e[1] = e[2] = e[3] = b.base + pos_own(s, 0);
                for (int k = 0; k < 16; ++k) deep[s * 16 + k] = halos[bi][s * 16 + k];
                if (!ok) continue;
                const int ty = b.type[s];
                if (ty == SIDE_MIRROR) continue;
                if (ty == SIDE_GENERAL) { ok = false; continue; }
                const int d = s / 2, td = 1 - d;
                const bool low = (s % 2) == 0;
                const float* hn = d == 0 ? hx : hy;
                const float* ht = d == 0 ? hy : hx;
                const float rt = b.rt[s];
                const int delta = s == 0 ? -1 : s == 1 ? 1 : s == 2 ? -8 : 8;
                // the distinct halo cells of the side in order: 8 (SAME), 4 (COARSE), 16 (FINE)
                const int step = ty == SIDE_FINE ? 1 : ty == SIDE_COARSE ? 4 : 2;
                const int ncell = 16 / step;
                const int near_expected = ty == SIDE_COARSE ? 2 : 1;
                for (int q = 0; q < ncell && ok; ++q) {
                    const int32_t h = halos[bi][s * 16 + q * step];
                    if (hn[h] != hn[b.base] * rt || ht[h] != ht[b.base] * rt) { ok = false; break; }
                    int32_t cells[2];
                    // (a) faces towards this block
                    int n = faces_of(h, d, low, cells);
                    if (n != near_expected) { ok = false; break; }
                    for (int k = 0; k < n; ++k)
                        if (cells[k] < b.base || cells[k] >= b.base + NPB) ok = false;
                    // (b) the deeper cell
                    n = faces_of(h, d, !low, cells);
                    if (!ok || n != 1 || cells[0] == h || hn[cells[0]] != hn[h]) { ok = false; break; }
                    for (int k = 0; k < step; ++k) deep[s * 16 + q * step + k] = cells[0];
                    if (cells[0] != h + delta) arithmetic = false;
                    // (c), (d) lateral neighbours
                    for (int side = 0; side < 2 && ok; ++side) {
                        n = faces_of(h, td, side == 1, cells);
                        if (n < 1) { ok = false; break; }
                        const int qn = q + (side ? 1 : -1);
                        if (qn >= 0 && qn < ncell) {
                            if (n != 1 || cells[0] != halos[bi][s * 16 + qn * step]) ok = false;
                        } else {
                            e[side * 2] = cells[0];
                            e[side * 2 + 1] = n == 2 ? cells[1] : cells[0];
                        }
                    }
                }
            }
            if (ok && !arithmetic) {
                b.dt = (int32_t)(dtab.size() / 64);
                dtab.insert(dtab.end(), deep, deep + 64);
            }
            fus[bi] = ok;
            nfus += ok;
        }
        info[8] = nfus;
    }
    // 7. mixed launches (partitions with skirt blocks): eligible blocks take the single kernel, the others the
    //    two-kernel form.  The gradient workspace is then needed only where somebody reads it: in the blocks
    //    that are not eligible, in their halo cells and in the face neighbours of the face-list cells.
    needg.assign(blocks.size(), 0);
    {
        std::vector<char> mark(nc, 0);
        for (int32_t c : irr) {
            mark[c] = 1;
            for (int d = 0; d < v.nd; ++d) {
                for (int32_t k = v.loff[d][c]; k < v.loff[d][c + 1]; ++k) mark[v.owners[d][v.lidx[d][k]]] = 1;
                for (int32_t k = v.roff[d][c]; k < v.roff[d][c + 1]; ++k) mark[v.neighbors[d][v.ridx[d][k]]] = 1;
            }
        }
        for (size_t bi = 0; bi < blocks.size(); ++bi)
            if (!fus[bi])
                for (int k = 0; k < 64; ++k) mark[htab[bi * 64 + k]] = 1;
        for (size_t bi = 0; bi < blocks.size(); ++bi) {
            bool need = !fus[bi];
            for (int k = 0; k < NPB && !need; ++k) need = mark[blocks[bi].base + k];
            needg[bi] = need;
        }
    }
    info[7] = n_phase1[1];
    info[0] = (int64_t)blocks.size();
    info[1] = (int64_t)irr.size();
    for (int k = 0; k < 5; ++k) info[2 + k] = counts[k];
}

// ------------------------------------------------------------------------------------------
// Quads: 2x2 groups of complete same-level blocks with consecutive bases whose inner sides are SAME sides onto each
// other and whose 8 outer half-sides are SAME / COARSE / FINE with arithmetic deeper cells (dt < 0).  89.6 % of the
// blocks of the 0.87 M-cell RAE2822 mesh sit in such groups (the four leaf children of a quadtree node).
// ------------------------------------------------------------------------------------------
// companion row of a quad / pair row (IBH_QAUX): end ids + per half-side the origin of arithmetic halo ids or -1
static void quad_aux_row(const int32_t* row, uint32_t cls, std::vector<int32_t>& aux) {
    const size_t a0 = aux.size();
    aux.resize(a0 + IBH_QAUX);
    for (int e = 0; e < 32; ++e) aux[a0 + e] = row[128 + e];
    for (int l = 0; l < 8; ++l) {
        const int g = l >> 1, half = l & 1;
        const uint32_t ty = (cls >> (4 * l)) & 15u;
        const int stride = (g == 0 || g == 3) ? 8 : 1;
        const int32_t* ids = row + 2 * (16 * g + 8 * half);
        int32_t orig = -1;
        if (ty == SIDE_SAME || ty == SIDE_COARSE) {
            orig = ids[0];
            for (int t = 0; t < 8 && orig >= 0; ++t) {
                const int32_t want = ids[0] + stride * (ty == SIDE_COARSE ? (t >> 1) : t);
                if (ids[2 * t] != want || ids[2 * t + 1] != want) orig = -1;
            }
        }
        aux[a0 + 32 + l] = orig;
    }
}

void ibh_build_quads2(const std::vector<BlockDesc2>& blocks, const std::vector<int32_t>& htab,
                      const std::vector<int32_t>& etab, const std::vector<char>& cand, int32_t nB1, QuadSet2& out,
                      int32_t nc) {
    out = QuadSet2();
    const int32_t nb = (int32_t)blocks.size();
    std::unordered_map<int32_t, int32_t> bybase;
    bybase.reserve(nb * 2);
    for (int32_t b = 0; b < nb; ++b) bybase[blocks[b].base] = b;
    std::vector<int32_t> order(nb);
    for (int32_t b = 0; b < nb; ++b) order[b] = b;
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return blocks[a].base < blocks[b].base; });
    std::vector<char> used(nb, 0);
    struct Q { int32_t b[4]; bool interior; };
    std::vector<Q> quads;
    auto same_to = [&](const BlockDesc2& a, int s, int32_t base) {
        return a.type[s] == SIDE_SAME && a.nb[s][0] == base;
    };
    // outer sides of the block at position k (0 LL, 1 LR, 2 UL, 3 UR)
    static const int outer[4][2] = {{0, 2}, {1, 2}, {0, 3}, {1, 3}};
    for (int32_t o : order) {
        if (used[o] || !cand[o]) continue;
        const BlockDesc2& b0 = blocks[o];
        int32_t idx[4] = {o, -1, -1, -1};
        bool ok = true;
        for (int k = 1; k < 4 && ok; ++k) {
            auto it = bybase.find(b0.base + 64 * k);
            ok = it != bybase.end() && !used[it->second] && cand[it->second];
            if (ok) idx[k] = it->second;
        }
        if (!ok) continue;
        const BlockDesc2 &b1 = blocks[idx[1]], &b2 = blocks[idx[2]], &b3 = blocks[idx[3]];
        ok = same_to(b0, 1, b1.base) && same_to(b1, 0, b0.base) && same_to(b0, 3, b2.base) && same_to(b2, 2, b0.base) &&
             same_to(b1, 3, b3.base) && same_to(b3, 2, b1.base) && same_to(b2, 1, b3.base) && same_to(b3, 0, b2.base);
        for (int k = 0; k < 4 && ok; ++k) {
            const BlockDesc2& b = blocks[idx[k]];
            ok = b.dt < 0 && b.h[0] == b0.h[0] && b.h[1] == b0.h[1];
            for (int e = 0; e < 2 && ok; ++e) {
                const int s = outer[k][e];
                const int ty = b.type[s];
                ok = ty == SIDE_SAME || ty == SIDE_COARSE || ty == SIDE_FINE;
                // paired gathers of the quad sweep (quad_load_halo_paired): on bottom / top sides the 8-byte load at a
                // halo cell must stay inside the arrays, and the two sub-face cells of a FINE side must be x neighbours
                if (ok && s >= 2)
                    for (int t = 0; t < 8 && ok; ++t) {
                        const int32_t h0 = htab[(size_t)idx[k] * 64 + (s * 8 + t) * 2], h1 = htab[(size_t)idx[k] * 64 + (s * 8 + t) * 2 + 1];
                        const int32_t dl = s == 2 ? -8 : 8;
                        ok = h0 + 1 < nc && h0 + dl >= 0 && h0 + dl + 1 < nc && (ty != SIDE_FINE || h1 == h0 + 1);
                    }
            }
        }
        if (!ok) continue;
        Q q;
        q.interior = true;
        for (int k = 0; k < 4; ++k) {
            q.b[k] = idx[k];
            used[idx[k]] = 1;
            q.interior = q.interior && idx[k] < nB1;
        }
        quads.push_back(q);
    }
    // side of lane row g, and the block holding half-side (g, half)
    static const int side_of_g[4] = {0, 2, 3, 1};
    static const int blk_of[4][2] = {{0, 2}, {0, 1}, {2, 3}, {1, 3}};
    for (int pass = 0; pass < 2; ++pass)
        for (const Q& q : quads) {
            if (q.interior != (pass == 0)) continue;
            QuadDesc2 d;
            d.base = blocks[q.b[0]].base;
            d.rh[0] = blocks[q.b[0]].rh[0];
            d.rh[1] = blocks[q.b[0]].rh[1];
            d.cls = 0;
            const size_t row = out.qtab.size();
            out.qtab.resize(row + IBH_QROW);
            for (int g = 0; g < 4; ++g)
                for (int half = 0; half < 2; ++half) {
                    const int s = side_of_g[g], l = 2 * g + half;
                    const int32_t bi = q.b[blk_of[g][half]];
                    d.cls |= (uint32_t)blocks[bi].type[s] << (4 * l);
                    for (int t = 0; t < 8; ++t)
                        for (int k = 0; k < 2; ++k)
                            out.qtab[row + 2 * (16 * g + 8 * half + t) + k] = htab[(size_t)bi * 64 + (s * 8 + t) * 2 + k];
                    for (int e = 0; e < 4; ++e) out.qtab[row + 128 + 4 * l + e] = etab[(size_t)bi * 16 + s * 4 + e];
                }
            out.qd.push_back(d);
            quad_aux_row(out.qtab.data() + row, d.cls, out.qaux);
            if (pass == 0) out.nq_int++;
        }
    for (int pass = 0; pass < 2; ++pass)
        for (int32_t b = 0; b < nb; ++b)
            if (cand[b] && !used[b] && (b < nB1) == (pass == 0)) {
                out.singles.push_back(b);
                if (pass == 0) out.ns_int++;
            }
    // ---- pairs among the blocks left over (only where no phase split exists: every block an interior-phase block)
    if (nB1 != nb) {
        out.singles2 = out.singles;
        return;
    }
    std::vector<char> paired(nb, 0);
    auto side_ok = [&](int32_t bi, int s) {
        const BlockDesc2& b = blocks[bi];
        const int ty = b.type[s];
        bool ok = ty == SIDE_SAME || ty == SIDE_COARSE || ty == SIDE_FINE;
        if (ok && s >= 2)
            for (int t = 0; t < 8 && ok; ++t) {
                const int32_t h0 = htab[(size_t)bi * 64 + (s * 8 + t) * 2], h1 = htab[(size_t)bi * 64 + (s * 8 + t) * 2 + 1];
                const int32_t dl = s == 2 ? -8 : 8;
                ok = h0 + 1 < nc && h0 + dl >= 0 && h0 + dl + 1 < nc && (ty != SIDE_FINE || h1 == h0 + 1);
            }
        return ok;
    };
    for (int32_t o : order) {
        if (used[o] || paired[o] || !cand[o]) continue;
        const BlockDesc2& b0 = blocks[o];
        auto it = bybase.find(b0.base + 64);
        if (it == bybase.end()) continue;
        const int32_t i1 = it->second;
        if (used[i1] || paired[i1] || !cand[i1]) continue;
        const BlockDesc2& b1 = blocks[i1];
        bool ok = same_to(b0, 1, b1.base) && same_to(b1, 0, b0.base) && b0.dt < 0 && b1.dt < 0 && b1.h[0] == b0.h[0] &&
                  b1.h[1] == b0.h[1];
        ok = ok && side_ok(o, 0) && side_ok(o, 2) && side_ok(o, 3) && side_ok(i1, 1) && side_ok(i1, 2) && side_ok(i1, 3);
        if (!ok) continue;
        paired[o] = paired[i1] = 1;
        // lane rows g = 0..3 = sides left, bottom, top, right; half-sides (g, half): the block that holds them, or the
        // half-side they repeat where a pair has none (left / right upper halves)
        const int32_t blk_of_pair[4][2] = {{o, o}, {o, i1}, {o, i1}, {i1, i1}};
        QuadDesc2 d;
        d.base = b0.base;
        d.rh[0] = b0.rh[0];
        d.rh[1] = b0.rh[1];
        d.cls = 0;
        const size_t row = out.ptab.size();
        out.ptab.resize(row + IBH_QROW);
        for (int g = 0; g < 4; ++g)
            for (int half = 0; half < 2; ++half) {
                const int s = side_of_g[g], l = 2 * g + half;
                const int32_t bi = blk_of_pair[g][half];
                d.cls |= (uint32_t)blocks[bi].type[s] << (4 * l);
                for (int t = 0; t < 8; ++t)
                    for (int k = 0; k < 2; ++k)
                        out.ptab[row + 2 * (16 * g + 8 * half + t) + k] = htab[(size_t)bi * 64 + (s * 8 + t) * 2 + k];
                for (int e = 0; e < 4; ++e) out.ptab[row + 128 + 4 * l + e] = etab[(size_t)bi * 16 + s * 4 + e];
            }
        out.pd.push_back(d);
        quad_aux_row(out.ptab.data() + row, d.cls, out.paux);
    }
    for (int32_t b : out.singles)
        if (!paired[b]) out.singles2.push_back(b);
}
