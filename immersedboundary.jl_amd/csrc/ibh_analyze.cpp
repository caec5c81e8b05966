// libibhip: host-side analysis that recovers the block structure of a Partition from the
// reference's own data (part.domain global ids + face lists) and classifies every block side
// for the block fast path.  Nothing is assumed: a side gets a fast class only if the faces the
// reference actually registered for its 8 boundary cells are exactly the faces that class
// implies (this also covers the reference's search-radius rule that can drop faces at >2:1
// jumps, ImmersedBoundary.jl:85,114-116).  Everything else falls back to the face-list kernels.
//
// Cell numbering inside a block is x-fastest (mesher.jl:1064-1112): local = i + 8*j.
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
                         std::vector<char>& needg) {
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
    // 6. single-kernel sweep (blk2::sweep_adv): the wave of a block also computes gradient + sensor of its 64
    //    halo cells, from (a) the halo values, (b) the cell one step deeper inside the neighbour block
    //    (halo id -/+ 1 or 8: interior to a verified block), (c) the lateral neighbours along the side = the
    //    adjacent halo slots, and (d) at the two ends of every side the cells across the neighbour block's
    //    perpendicular side: etab[blk][(s*2 + end)*2 + k], taken from that block's own (verified) halo row.
    //    A block is eligible when none of its sides is GENERAL, every neighbour block is in the table with a
    //    verified side facing back, the perpendicular sides used at the ends are not GENERAL, and the two fine
    //    blocks of a FINE side are SAME-level neighbours of each other.
    etab.assign(blocks.size() * 16, 0);
    fus.assign(blocks.size(), 0);
    {
        std::unordered_map<int32_t, int32_t> base2idx;
        for (size_t bi = 0; bi < blocks.size(); ++bi) base2idx[blocks[bi].base] = (int32_t)bi;
        auto find = [&](int32_t base) -> int32_t {
            auto it = base2idx.find(base);
            return it == base2idx.end() ? -1 : it->second;
        };
        int64_t nfus = 0;
        for (size_t bi = 0; bi < blocks.size(); ++bi) {
            const BlockDesc2& b = blocks[bi];
            bool ok = true;
            for (int k = 0; k < NPB && ok; ++k) ok = !cell_irr[b.base + k];
            for (int s = 0; s < 4; ++s) {
                int32_t* e = &etab[bi * 16 + s * 4];
                e[0] = e[1] = e[2] = e[3] = b.base + pos_own(s, 0);
                if (!ok) continue;
                const int ty = b.type[s];
                if (ty == SIDE_MIRROR) continue;
                if (ty == SIDE_GENERAL) { ok = false; continue; }
                const int td = 1 - s / 2, s_lo = 2 * td, s_hi = s_lo + 1, opp = s ^ 1;
                const int tp = (s % 2 == 0) ? 7 : 0;  // index of the halo line along the neighbour's perpendicular sides
                auto edge_ids = [&](int32_t ni, int side, int32_t* out) -> bool {
                    if (blocks[ni].type[side] == SIDE_GENERAL) return false;
                    out[0] = htab[(size_t)ni * 64 + (side * 8 + tp) * 2];
                    out[1] = htab[(size_t)ni * 64 + (side * 8 + tp) * 2 + 1];
                    return true;
                };
                const int32_t n0 = find(b.nb[s][0]);
                if (n0 < 0 || blocks[n0].type[opp] == SIDE_GENERAL) { ok = false; continue; }
                if (ty == SIDE_SAME) {
                    ok = edge_ids(n0, s_lo, e) && edge_ids(n0, s_hi, e + 2);
                } else if (ty == SIDE_COARSE) {
                    const int sub = b.sub[s];
                    if (sub == 0) ok = edge_ids(n0, s_lo, e);
                    else e[0] = e[1] = blocks[n0].base + pos_opp(s, 3);
                    if (ok) {
                        if (sub == 1) ok = edge_ids(n0, s_hi, e + 2);
                        else e[2] = e[3] = blocks[n0].base + pos_opp(s, 4);
                    }
                } else {  // FINE
                    const int32_t n1 = find(b.nb[s][1]);
                    if (n1 < 0 || blocks[n1].type[opp] == SIDE_GENERAL) { ok = false; continue; }
                    ok = blocks[n0].type[s_hi] == SIDE_SAME && blocks[n0].nb[s_hi][0] == blocks[n1].base &&
                         blocks[n1].type[s_lo] == SIDE_SAME && blocks[n1].nb[s_lo][0] == blocks[n0].base &&
                         edge_ids(n0, s_lo, e) && edge_ids(n1, s_hi, e + 2);
                }
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
