// libibhip: block analysis of the IMAGE blocks of a 3-D partition with skirt fragments -- what a rank of a multi-GPU run
// sweeps (dom(f, args...) scatters back the image cells only, ImmersedBoundary.jl:842-845; the skirt is the two cell
// layers the reference's skirt growth adds around them, :610-619).  Same classification as ibh_analyze3.cpp, but every
// class is read off the GEOMETRY of the registered faces (global ids give block and position of every neighbour cell), so
// the neighbour need not be a complete block of this partition:
//   * halo table: the cells the faces name (local ids: a fragment's cells are where the partition put them);
//   * deeper-cell table dtab3[blk][side][t][k]: the cell one step further behind halo cell k of slot t, from the halo
//     cell's own far face (inside a complete block it is index arithmetic; inside a fragment it is not);
//   * rim table: as in ibh_analyze3.cpp (it was table driven already).
// A side whose neighbour block IS complete here keeps nb >= 0 (arithmetic ids in the kernels); nb = -1 sends the kernels to
// the tables.  The single-kernel sweeps run over the image blocks when EVERY image block qualifies (img_all).
#include <unordered_map>

#include "ibh_common.h"

namespace {

constexpr int NPB = 512;
const int STR[3] = {1, 8, 64};
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

void ibh_analyze_image3(const HostPartView& v, const int32_t* image_in_domain, int32_t n_image, Image3Host& out) {
    out = Image3Host();
    const int32_t nc = v.nc;
    if (n_image <= 0 || !image_in_domain) return;
    const float* hh[3] = {v.spacing, v.spacing + nc, v.spacing + 2 * (size_t)nc};
    auto gid = [&](int32_t c) { return (int64_t)v.domain[c] - v.index_base; };
    std::vector<char> is_image(nc, 0);
    for (int32_t k = 0; k < n_image; ++k) is_image[image_in_domain[k] - v.index_base] = 1;

    // complete blocks of the partition (any), image blocks among them
    std::unordered_map<int64_t, int32_t> blockbase;
    std::vector<int32_t> bases;
    for (int32_t c = 0; c + NPB <= nc;) {
        const int64_t g = gid(c);
        bool ok = (g % NPB == 0) && gid(c + NPB - 1) == g + NPB - 1;
        if (ok)
            for (int k = 1; k < NPB && ok; ++k)
                ok = gid(c + k) == g + k && hh[0][c + k] == hh[0][c] && hh[1][c + k] == hh[1][c] && hh[2][c + k] == hh[2][c];
        if (ok) {
            blockbase[g / NPB] = c;
            bool img = true;
            for (int k = 0; k < NPB && img; ++k) img = is_image[c + k];
            if (img) bases.push_back(c);
            c += NPB;
        } else {
            ++c;
        }
    }
    if ((int64_t)bases.size() * NPB != (int64_t)n_image) return;  // image cells outside complete blocks: not eligible
    auto single = [&](const std::vector<int32_t>& off, const std::vector<int32_t>& idx, int32_t c) -> int32_t {
        return (off[c + 1] - off[c] == 1) ? idx[off[c]] : -1;
    };
    // the cell behind cell h across its far face in dimension d (away from the block: low side -> towards -)
    auto beyond = [&](int32_t h, int d, bool low, int32_t& o) -> bool {
        const std::vector<int32_t>& off = low ? v.loff[d] : v.roff[d];
        const std::vector<int32_t>& idx = low ? v.lidx[d] : v.ridx[d];
        const int32_t f = single(off, idx, h);
        if (f < 0) return false;
        const int32_t me = low ? v.neighbors[d][f] : v.owners[d][f];
        o = low ? v.owners[d][f] : v.neighbors[d][f];
        return me == h && o != h && hh[d][o] == hh[d][h];
    };

    bool all = !bases.empty();
    const size_t nb_ = bases.size();
    out.blocks.reserve(nb_);
    out.htab.assign(nb_ * 384, 0);
    out.dtab.assign(nb_ * 1536, 0);
    out.rtab.assign(nb_ * 384, 0);
    for (size_t bi = 0; bi < nb_ && all; ++bi) {
        const int32_t base = bases[bi];
        for (int pos = 0; pos < NPB && all; ++pos) {  // interior faces must be the implicit ones
            const int32_t c = base + pos;
            for (int d = 0; d < 3 && all; ++d) {
                const int x = coord(pos, d);
                if (x < 7) {
                    const int32_t f = single(v.roff[d], v.ridx[d], c);
                    all = f >= 0 && v.owners[d][f] == c && v.neighbors[d][f] == c + STR[d];
                }
                if (all && x > 0) {
                    const int32_t f = single(v.loff[d], v.lidx[d], c);
                    all = f >= 0 && v.owners[d][f] == c - STR[d] && v.neighbors[d][f] == c;
                }
            }
        }
        if (!all) break;
        BlockDesc3 b;
        b.base = base;
        b.fine = -1;
        for (int d = 0; d < 3; ++d) {
            b.h[d] = hh[d][base];
            b.rh[d] = 1.0f / b.h[d];
        }
        int32_t* hrow = out.htab.data() + bi * 384;
        int32_t* drow = out.dtab.data() + bi * 1536;
        for (int s = 0; s < 6 && all; ++s) {
            const int d = s / 2;
            const bool low = (s % 2) == 0;
            const float hc = hh[d][base];
            int a, bb2;
            tang(d, a, bb2);
            const std::vector<int32_t>& off = low ? v.loff[d] : v.roff[d];
            const std::vector<int32_t>& idx = low ? v.lidx[d] : v.ridx[d];
            int type = -1, sub = -1;
            int64_t nbg = -1;                 // global block of the neighbour (SAME / COARSE)
            int32_t fcell[64][4];
            for (int t = 0; t < 64 && all; ++t) {
                const int t1 = t % 8, t2 = t / 8;
                const int32_t c = base + pos3(d, low ? 0 : 7, t1, t2);
                const int nfc = off[c + 1] - off[c];
                int ty;
                if (nfc == 4) {
                    ty = SIDE_FINE;
                    bool seen[4] = {false, false, false, false};
                    for (int kf = 0; kf < 4 && all; ++kf) {
                        const int32_t f4 = idx[off[c] + kf];
                        const int32_t me4 = low ? v.neighbors[d][f4] : v.owners[d][f4];
                        const int32_t o4 = low ? v.owners[d][f4] : v.neighbors[d][f4];
                        if (me4 != c || o4 == c || hh[d][o4] != hc * 0.5f) { all = false; break; }
                        const int pos4 = (int)(gid(o4) % NPB);
                        if (coord(pos4, d) != (low ? 7 : 0)) { all = false; break; }
                        const int k1 = coord(pos4, a) - 2 * (t1 & 3), k2 = coord(pos4, bb2) - 2 * (t2 & 3);
                        if (k1 < 0 || k1 > 1 || k2 < 0 || k2 > 1 || seen[k1 + 2 * k2]) { all = false; break; }
                        seen[k1 + 2 * k2] = true;
                        fcell[t][k1 + 2 * k2] = o4;
                    }
                    if (!all) break;
                    for (int k = 0; k < 4; ++k) {
                        int32_t o;
                        if (!beyond(fcell[t][k], d, low, o)) { all = false; break; }
                        drow[(s * 64 + t) * 4 + k] = o;
                    }
                    hrow[s * 64 + t] = fcell[t][0];
                } else if (nfc == 1) {
                    const int32_t f = idx[off[c]];
                    const int32_t me = low ? v.neighbors[d][f] : v.owners[d][f];
                    const int32_t o = low ? v.owners[d][f] : v.neighbors[d][f];
                    if (me != c) { all = false; break; }
                    if (o == c) {
                        ty = SIDE_MIRROR;
                        hrow[s * 64 + t] = c;
                        for (int k = 0; k < 4; ++k) drow[(s * 64 + t) * 4 + k] = c;
                    } else {
                        const int64_t g = gid(o);
                        const int pos = (int)(g % NPB);
                        if (coord(pos, d) != (low ? 7 : 0)) { all = false; break; }
                        const int o1 = coord(pos, a), o2 = coord(pos, bb2);
                        if (hh[d][o] == hc) {
                            ty = SIDE_SAME;
                            if (o1 != t1 || o2 != t2) { all = false; break; }
                            if (t == 0) nbg = g / NPB; else if (nbg != g / NPB) { all = false; break; }
                        } else if (hh[d][o] == hc * 2.0f) {
                            ty = SIDE_COARSE;
                            const int q1 = o1 - t1 / 2, q2 = o2 - t2 / 2;
                            if ((q1 != 0 && q1 != 4) || (q2 != 0 && q2 != 4)) { all = false; break; }
                            const int q = q1 / 4 + 2 * (q2 / 4);
                            if (t == 0) { nbg = g / NPB; sub = q; } else if (nbg != g / NPB || sub != q) { all = false; break; }
                        } else { all = false; break; }
                        int32_t od;
                        if (!beyond(o, d, low, od)) { all = false; break; }
                        hrow[s * 64 + t] = o;
                        for (int k = 0; k < 4; ++k) drow[(s * 64 + t) * 4 + k] = od;
                    }
                } else { all = false; break; }
                if (t == 0) type = ty; else if (type != ty) { all = false; break; }
            }
            if (!all) break;
            b.type[s] = type;
            b.nb[s] = -1;
            if (type == SIDE_SAME || type == SIDE_COARSE) {
                auto it = blockbase.find(nbg);
                if (it != blockbase.end()) b.nb[s] = it->second;  // complete here: arithmetic ids
            }
            b.sub[s] = sub < 0 ? 0 : sub;
            b.q[s] = type == SIDE_COARSE ? (1.0f / 3.0f) : type == SIDE_FINE ? (2.0f / 3.0f) : 0.5f;
            b.rt[s] = type == SIDE_COARSE ? 2.0f : type == SIDE_FINE ? 0.5f : 1.0f;
            if (type == SIDE_FINE) {
                if (b.fine < 0) {
                    b.fine = (int32_t)(out.ftab.size() / (6 * 64 * 3));
                    out.ftab.resize(out.ftab.size() + 6 * 64 * 3, b.base);
                }
                b.nb[s] = -2;
                for (int t = 0; t < 64; ++t)
                    for (int k = 1; k < 4; ++k) out.ftab[((size_t)b.fine * 6 + s) * 64 * 3 + t * 3 + (k - 1)] = fcell[t][k];
            }
            // rim table (ibh_analyze3.cpp): the lateral neighbours of the halo cells along the rims of the side
            int32_t* rrow = out.rtab.data() + bi * 384;
            for (int r = 0; r < 64; ++r) rrow[s * 64 + r] = b.base;
            if (type == SIDE_MIRROR) continue;
            const int n = type == SIDE_FINE ? 16 : 8;
            auto halo = [&](int f1, int f2) -> int32_t {
                if (type != SIDE_FINE) return hrow[s * 64 + f1 + 8 * f2];
                return fcell[(f1 >> 1) + 8 * (f2 >> 1)][(f1 & 1) + 2 * (f2 & 1)];
            };
            for (int r = 0; r < 4 && all; ++r) {
                const int dim = r < 2 ? a : bb2;
                const bool lo = (r & 1) == 0;
                const std::vector<int32_t>& roff_ = lo ? v.loff[dim] : v.roff[dim];
                const std::vector<int32_t>& ridx_ = lo ? v.lidx[dim] : v.ridx[dim];
                for (int i = 0; i < n && all; ++i) {
                    const int e = lo ? 0 : n - 1;
                    const int32_t h = r < 2 ? halo(e, i) : halo(i, e);
                    const int cnt = roff_[h + 1] - roff_[h];
                    if (cnt != 1 && cnt != 4) { all = false; break; }
                    int32_t o4[4];
                    for (int k = 0; k < cnt; ++k) {
                        const int32_t f = ridx_[roff_[h] + k];
                        const int32_t me = lo ? v.neighbors[dim][f] : v.owners[dim][f];
                        if (me != h) all = false;
                        o4[k] = lo ? v.owners[dim][f] : v.neighbors[dim][f];
                    }
                    if (cnt == 1) {
                        rrow[s * 64 + r * n + i] = o4[0];
                    } else {
                        rrow[s * 64 + r * n + i] = -(int32_t)(out.r4tab.size() / 4 + 1);
                        out.r4tab.insert(out.r4tab.end(), o4, o4 + 4);
                    }
                }
            }
        }
        if (all) out.blocks.push_back(b);
    }
    out.all = all && out.blocks.size() == nb_;
    if (!out.all) out = Image3Host();
}
