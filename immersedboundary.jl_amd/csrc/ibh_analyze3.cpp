// libibhip: 3-D counterpart of ibh_analyze.cpp -- recovers the 8x8x8 block structure of a Partition from
// part.domain + the face lists and classifies every block side, verified face by face against what the
// reference registered.  Cell numbering inside a block: local = i + 8*j + 64*k (mesher.jl:1064-1112).
#include <unordered_map>

#include "ibh_common.h"

namespace {

constexpr int NPB = 512;
const int STR[3] = {1, 8, 64};

// tangential dims of dim d, increasing
inline void tang(int d, int& a, int& b) {
    a = d == 0 ? 1 : 0;
    b = d == 2 ? 1 : 2;
}
inline int pos3(int d, int n, int t1, int t2) {
    int a, b;
    tang(d, a, b);
    return n * STR[d] + t1 * STR[a] + t2 * STR[b];
}
inline int coord(int pos, int d) { return (pos / STR[d]) % 8; }

}  // namespace

void ibh_analyze_blocks3(const HostPartView& v, std::vector<BlockDesc3>& blocks, std::vector<int32_t>& irr,
                         int64_t* info, const int32_t* image_in_domain, int32_t n_image, int32_t* n_phase1,
                         std::vector<int32_t>& htab, std::vector<int32_t>& ftab, Sweep3Host* sw) {
    const int32_t nc = v.nc;
    const float* hh[3] = {v.spacing, v.spacing + nc, v.spacing + 2 * (size_t)nc};
    auto gid = [&](int32_t c) { return (int64_t)v.domain[c] - v.index_base; };

    std::unordered_map<int64_t, int32_t> blockbase;
    std::vector<int32_t> bases;
    std::vector<char> in_full(nc, 0);
    for (int32_t c = 0; c + NPB <= nc;) {
        int64_t g = gid(c);
        bool ok = (g % NPB == 0) && gid(c + NPB - 1) == g + NPB - 1;
        if (ok)
            for (int k = 1; k < NPB && ok; ++k)
                ok = gid(c + k) == g + k && hh[0][c + k] == hh[0][c] && hh[1][c + k] == hh[1][c] &&
                     hh[2][c + k] == hh[2][c];
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
    for (int32_t c = 0; c < nc; ++c) cell_irr[c] = !in_full[c];
    int64_t counts[5] = {0, 0, 0, 0, 0};
    std::vector<int32_t> fine0;        // sub-face-0 cells of FINE sides, 64 per entry of fine0_key
    std::vector<int64_t> fine0_key;    // (block base << 3) | side
    std::unordered_map<int64_t, std::vector<int32_t>> fine_nb;  // same key -> bases of the 4 fine neighbour blocks

    for (int32_t base : bases) {
        bool good = true;
        for (int pos = 0; pos < NPB && good; ++pos) {
            const int32_t c = base + pos;
            for (int d = 0; d < 3 && good; ++d) {
                const int x = coord(pos, d);
                if (x < 7) {
                    int32_t f = single(v.roff[d], v.ridx[d], c);
                    good = f >= 0 && v.owners[d][f] == c && v.neighbors[d][f] == c + STR[d];
                }
                if (good && x > 0) {
                    int32_t f = single(v.loff[d], v.lidx[d], c);
                    good = f >= 0 && v.owners[d][f] == c - STR[d] && v.neighbors[d][f] == c;
                }
            }
        }
        if (!good) {
            for (int k = 0; k < NPB; ++k) cell_irr[base + k] = 1;
            continue;
        }
        BlockDesc3 b;
        b.base = base;
        b.fine = -1;
        int32_t fcell[6][64][4];  // fine neighbour cells of FINE sides, sub-face k = k1 + 2*k2
        for (int d = 0; d < 3; ++d) {
            b.h[d] = hh[d][base];
            b.rh[d] = 1.0f / b.h[d];
        }
        for (int s = 0; s < 6; ++s) {
            const int d = s / 2;
            const bool low = (s % 2) == 0;
            const float hc = hh[d][base];
            const std::vector<int32_t>& off = low ? v.loff[d] : v.roff[d];
            const std::vector<int32_t>& idx = low ? v.lidx[d] : v.ridx[d];
            int type = -1, sub = -1;
            int32_t nb = -1;
            bool ok = true;
            for (int t = 0; t < 64 && ok; ++t) {
                const int t1 = t % 8, t2 = t / 8;
                const int32_t c = base + pos3(d, low ? 0 : 7, t1, t2);
                const int nfc = off[c + 1] - off[c];
                if (nfc == 4) {  // candidate FINE side: 4 faces to the 2x2 fine cells behind this boundary cell
                    int a, bb2;
                    tang(d, a, bb2);
                    bool seen[4] = {false, false, false, false};
                    for (int kf = 0; kf < 4 && ok; ++kf) {
                        const int32_t f4 = idx[off[c] + kf];
                        const int32_t me4 = low ? v.neighbors[d][f4] : v.owners[d][f4];
                        const int32_t o4 = low ? v.owners[d][f4] : v.neighbors[d][f4];
                        if (me4 != c || o4 == c || !in_full[o4] || hh[d][o4] != hc * 0.5f) { ok = false; break; }
                        const int pos4 = (int)(gid(o4) % NPB);
                        if (coord(pos4, d) != (low ? 7 : 0)) { ok = false; break; }
                        const int k1 = coord(pos4, a) - 2 * (t1 & 3), k2 = coord(pos4, bb2) - 2 * (t2 & 3);
                        if (k1 < 0 || k1 > 1 || k2 < 0 || k2 > 1 || seen[k1 + 2 * k2]) { ok = false; break; }
                        seen[k1 + 2 * k2] = true;
                        fcell[s][t][k1 + 2 * k2] = o4;
                    }
                    if (!ok) break;
                    if (t == 0) type = SIDE_FINE; else if (type != SIDE_FINE) { ok = false; break; }
                    continue;
                }
                if (nfc != 1) { ok = false; break; }
                const int32_t f = idx[off[c]];
                const int32_t me = low ? v.neighbors[d][f] : v.owners[d][f];
                const int32_t o = low ? v.owners[d][f] : v.neighbors[d][f];
                if (me != c) { ok = false; break; }
                int ty;
                if (o == c) {
                    ty = SIDE_MIRROR;
                } else {
                    if (!in_full[o]) { ok = false; break; }
                    const int64_t g = gid(o);
                    const int32_t ob = blockbase[g / NPB];
                    const int pos = (int)(g % NPB);
                    if (coord(pos, d) != (low ? 7 : 0)) { ok = false; break; }
                    int a, bb2;
                    tang(d, a, bb2);
                    const int o1 = coord(pos, a), o2 = coord(pos, bb2);
                    if (hh[d][o] == hc) {
                        ty = SIDE_SAME;
                        if (o1 != t1 || o2 != t2) { ok = false; break; }
                        if (t == 0) nb = ob; else if (nb != ob) { ok = false; break; }
                    } else if (hh[d][o] == hc * 2.0f) {
                        ty = SIDE_COARSE;
                        const int q1 = o1 - t1 / 2, q2 = o2 - t2 / 2;
                        if ((q1 != 0 && q1 != 4) || (q2 != 0 && q2 != 4)) { ok = false; break; }
                        const int q = q1 / 4 + 2 * (q2 / 4);
                        if (t == 0) { nb = ob; sub = q; } else if (nb != ob || sub != q) { ok = false; break; }
                    } else { ok = false; break; }
                }
                if (t == 0) type = ty; else if (type != ty) { ok = false; break; }
            }
            if (!ok) type = SIDE_GENERAL;
            b.type[s] = type;
            b.nb[s] = nb;
            b.sub[s] = sub < 0 ? 0 : sub;
            b.q[s] = type == SIDE_COARSE ? (1.0f / 3.0f) : type == SIDE_FINE ? (2.0f / 3.0f) : 0.5f;
            b.rt[s] = type == SIDE_COARSE ? 2.0f : type == SIDE_FINE ? 0.5f : 1.0f;
            if (type == SIDE_FINE) {
                if (b.fine < 0) {
                    b.fine = (int32_t)(ftab.size() / (6 * 64 * 3));
                    ftab.resize(ftab.size() + 6 * 64 * 3, b.base);
                }
                b.nb[s] = -2;  // marker: sub-face 0 cells are taken from fcell when the halo table is written
                for (int t = 0; t < 64; ++t)
                    for (int k = 1; k < 4; ++k)
                        ftab[((size_t)b.fine * 6 + s) * 64 * 3 + t * 3 + (k - 1)] = fcell[s][t][k];
                for (int t = 0; t < 64; ++t) fine0.push_back(fcell[s][t][0]);
                fine0_key.push_back(((int64_t)base << 3) | s);
                std::vector<int32_t>& nbs = fine_nb[((int64_t)base << 3) | s];
                for (int t = 0; t < 64; t += 4)  // one boundary cell of every 4x4 quadrant (t1 = 0 or 4, t2 = 0 or 4)
                    if ((t % 8) % 4 == 0 && (t / 8) % 4 == 0) nbs.push_back(blockbase[gid(fcell[s][t][0]) / NPB]);
            }
            counts[type]++;
            if (type == SIDE_GENERAL)
                for (int t = 0; t < 64; ++t) cell_irr[base + pos3(d, low ? 0 : 7, t % 8, t / 8)] = 1;
        }
        blocks.push_back(b);
    }
    for (int32_t c = 0; c < nc; ++c)
        if (cell_irr[c]) irr.push_back(c);

    // overlap phases, as in 2-D
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
            for (int s = 0; s < 6 && ok; ++s) {
                const int ty = blocks[b].type[s];
                if (ty == SIDE_MIRROR) continue;
                if (ty == SIDE_GENERAL) { ok = false; break; }
                if (ty == SIDE_FINE) {
                    for (int32_t fb : fine_nb[((int64_t)blocks[b].base << 3) | s]) {
                        auto itf = base2a1.find(fb);
                        ok = ok && itf != base2a1.end() && itf->second;
                    }
                    continue;
                }
                auto it = base2a1.find(blocks[b].nb[s]);
                ok = it != base2a1.end() && it->second;
            }
            b1[b] = ok;
        }
        std::vector<BlockDesc3> ordered;
        ordered.reserve(blocks.size());
        for (size_t b = 0; b < blocks.size(); ++b) if (b1[b]) ordered.push_back(blocks[b]);
        n_phase1[1] = (int32_t)ordered.size();
        for (size_t b = 0; b < blocks.size(); ++b) if (a1[b] && !b1[b]) ordered.push_back(blocks[b]);
        n_phase1[0] = (int32_t)ordered.size();
        for (size_t b = 0; b < blocks.size(); ++b) if (!a1[b]) ordered.push_back(blocks[b]);
        blocks.swap(ordered);
    }
    std::unordered_map<int64_t, size_t> fine0_at;
    for (size_t e = 0; e < fine0_key.size(); ++e) fine0_at[fine0_key[e]] = e;
    htab.resize(blocks.size() * 384);
    for (size_t bi = 0; bi < blocks.size(); ++bi) {
        const BlockDesc3& b = blocks[bi];
        for (int s = 0; s < 6; ++s) {
            const int d = s / 2;
            const bool low = (s % 2) == 0;
            for (int t = 0; t < 64; ++t) {
                const int t1 = t % 8, t2 = t / 8;
                int32_t cell;
                if (b.type[s] == SIDE_SAME) cell = b.nb[s] + pos3(d, low ? 7 : 0, t1, t2);
                else if (b.type[s] == SIDE_COARSE)
                    cell = b.nb[s] + pos3(d, low ? 7 : 0, 4 * (b.sub[s] & 1) + t1 / 2, 4 * (b.sub[s] >> 1) + t2 / 2);
                else if (b.type[s] == SIDE_FINE)
                    cell = fine0[fine0_at[((int64_t)b.base << 3) | s] * 64 + t];
                else cell = b.base + pos3(d, low ? 0 : 7, t1, t2);
                htab[bi * 384 + s * 64 + t] = cell;
            }
        }
    }
    // ---- single-kernel sweep (blk3::sweep_adv): the workgroup of a block also computes slope + sensor of its halo
    // cells, so it needs the cell one step deeper behind every halo cell (same neighbour block: index arithmetic) and
    // the lateral neighbours of the halo cells: other halo cells of the same side, except along the rim of the side,
    // where they lie across a side of the NEIGHBOUR block -> rim table rtab[blk][side][64]: entry r*n + i = the cell
    // beyond rim r (tangential dim a low / high, dim b low / high) at position i along it, n = 8 (SAME / COARSE) or
    // 16 (FINE: the rims of the 16 x 16 patch of fine halo cells); >= 0: a cell (the halo cell itself at the domain
    // boundary), < 0: -(k + 1) = row k of r4tab, the four finer cells beyond.  The sweep is used when EVERY block
    // qualifies (sides SAME / COARSE / FINE / MIRROR towards verified full blocks, rim neighbours 1 or 4 cells).
    if (sw) {
        std::unordered_map<int32_t, int32_t> base2bi;
        for (size_t bi = 0; bi < blocks.size(); ++bi) base2bi[blocks[bi].base] = (int32_t)bi;
        bool all = irr.empty() && !blocks.empty();
        sw->rtab.assign(blocks.size() * 384, 0);
        for (size_t bi = 0; bi < blocks.size() && all; ++bi) {
            const BlockDesc3& b = blocks[bi];
            int32_t* row = sw->rtab.data() + bi * 384;
            for (int s = 0; s < 6 && all; ++s) {
                const int ty = b.type[s];
                const int d = s / 2;
                int a, bb2;
                tang(d, a, bb2);
                for (int r = 0; r < 64; ++r) row[s * 64 + r] = b.base;
                if (ty == SIDE_MIRROR) continue;
                if (ty != SIDE_SAME && ty != SIDE_COARSE && ty != SIDE_FINE) { all = false; break; }
                if (ty != SIDE_FINE && !base2bi.count(b.nb[s])) { all = false; break; }
                if (ty == SIDE_FINE)
                    for (int32_t fb : fine_nb[((int64_t)b.base << 3) | s])
                        if (!base2bi.count(fb)) all = false;
                if (!all) break;
                const int n = ty == SIDE_FINE ? 16 : 8;
                // halo cell at position (f1, f2) of the side's plane
                auto halo = [&](int f1, int f2) -> int32_t {
                    if (ty != SIDE_FINE) return htab[bi * 384 + s * 64 + f1 + 8 * f2];
                    const int t = (f1 >> 1) + 8 * (f2 >> 1), k = (f1 & 1) + 2 * (f2 & 1);
                    return k == 0 ? htab[bi * 384 + s * 64 + t]
                                  : ftab[((size_t)b.fine * 6 + s) * 64 * 3 + t * 3 + (k - 1)];
                };
                for (int r = 0; r < 4 && all; ++r) {
                    const int dim = r < 2 ? a : bb2;
                    const bool lo = (r & 1) == 0;
                    const std::vector<int32_t>& off = lo ? v.loff[dim] : v.roff[dim];
                    const std::vector<int32_t>& idx = lo ? v.lidx[dim] : v.ridx[dim];
                    for (int i = 0; i < n && all; ++i) {
                        const int e = lo ? 0 : n - 1;
                        const int32_t h = r < 2 ? halo(e, i) : halo(i, e);
                        if (ty == SIDE_FINE && !in_full[h]) { all = false; break; }
                        const int cnt = off[h + 1] - off[h];
                        if (cnt != 1 && cnt != 4) { all = false; break; }
                        int32_t o4[4];
                        for (int k = 0; k < cnt; ++k) {
                            const int32_t f = idx[off[h] + k];
                            const int32_t me = lo ? v.neighbors[dim][f] : v.owners[dim][f];
                            if (me != h) all = false;
                            o4[k] = lo ? v.owners[dim][f] : v.neighbors[dim][f];
                        }
                        if (cnt == 1) {
                            row[s * 64 + r * n + i] = o4[0];  // o == h: domain boundary, the difference is zero
                        } else {
                            row[s * 64 + r * n + i] = -(int32_t)(sw->r4tab.size() / 4 + 1);
                            sw->r4tab.insert(sw->r4tab.end(), o4, o4 + 4);
                        }
                    }
                }
            }
        }
        sw->all = all;
        if (!all) {
            sw->rtab.clear();
            sw->r4tab.clear();
        }
    }
    info[0] = (int64_t)blocks.size();
    info[1] = (int64_t)irr.size();
    for (int k = 0; k < 5; ++k) info[2 + k] = counts[k];
    info[7] = n_phase1[1];
}
