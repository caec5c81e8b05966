// libibhip: runtime, memory and handle management (host side of the C ABI).
#include <string.h>

#include "ibh_common.h"

thread_local std::string ibh_err;
thread_local hipStream_t ibh_stream = nullptr;

int ibh_fail(int code, const char* what, const char* file, int line) {
    ibh_err = std::string(what) + " (" + file + ":" + std::to_string(line) + ")";
    return code ? code : -1;
}

template <class T>
int ibh_upload(T** dptr, const T* host, size_t n) {
    *dptr = nullptr;
    if (n == 0) return 0;
    IBH_HIP(hipMalloc((void**)dptr, n * sizeof(T)));
    IBH_HIP(hipMemcpy(*dptr, host, n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}
template int ibh_upload<int32_t>(int32_t**, const int32_t*, size_t);
template int ibh_upload<float>(float**, const float*, size_t);
template int ibh_upload<BlockDesc2>(BlockDesc2**, const BlockDesc2*, size_t);
template int ibh_upload<BlockDesc3>(BlockDesc3**, const BlockDesc3*, size_t);

static std::vector<int32_t> rebased(const int32_t* p, size_t n, int base) {
    std::vector<int32_t> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = p[i] - base;
    return v;
}

// Host-only part of ibh_partition_create for 2-D partitions: block analysis, launch lists, quads.  No HIP call in
// here, so the CPU tests can run it through ibh_analyze2_host().
struct Host2D {
    std::vector<BlockDesc2> blocks;
    std::vector<int32_t> irr, htab, etab, dtab, img, fz, ng, nf;
    std::vector<char> fus, needg;
    int32_t nph[2] = {0, 0};
    int32_t n_img_int = 0, img_all_fz = 0, fuse_all = 0, rows_ok = 0, n_fz_int = 0, n_ng_int = 0, n_nf_int = 0;
    int64_t info[12] = {0};
    QuadSet2 quads[2];  // 0: all blocks (fuse_all partitions), 1: image blocks (img_all_fz partitions)
};
struct ibh_host2d {
    Host2D H;
};

static void analyze2_host(const HostPartView& v, int32_t n_image, const int32_t* image_in_domain, int index_base,
                          Host2D& H) {
    const int32_t nc = v.nc;
    const bool have_img = n_image > 0 && image_in_domain;
    std::vector<int32_t> none;
    ibh_analyze_blocks2(v, H.blocks, H.irr, H.info, have_img ? image_in_domain : none.data(), have_img ? n_image : 0,
                        H.nph, H.htab, H.etab, H.fus, H.needg, H.dtab);
    const int32_t nb = (int32_t)H.blocks.size();
    H.fuse_all = nb > 0 && H.irr.empty() && H.info[8] == (int64_t)nb;
    std::vector<char> is_imgblk(nb, 0);
    if (have_img) {  // image blocks: eligible, all of them?
        std::vector<char> is_img(nc, 0);
        for (int32_t k = 0; k < n_image; ++k) is_img[image_in_domain[k] - index_base] = 1;
        bool all = true;
        for (int32_t b = 0; b < nb; ++b)
            if (is_img[H.blocks[b].base]) {
                H.img.push_back(b);
                is_imgblk[b] = 1;
                H.n_img_int += b < H.nph[1];
                all = all && H.fus[b];
            }
        H.img_all_fz = all && (int64_t)H.img.size() * 64 == (int64_t)n_image;
        H.info[10] = H.img_all_fz;
        H.info[11] = (int64_t)H.img.size();
    }
    if (!H.fuse_all && H.info[8] > 0)
        for (int32_t b = 0; b < nb; ++b) {
            if (H.fus[b]) { H.fz.push_back(b); H.n_fz_int += b < H.nph[1]; }
            else { H.nf.push_back(b); H.n_nf_int += b < H.nph[1]; }
            if (H.needg[b]) { H.ng.push_back(b); H.n_ng_int += b < H.nph[0]; }
        }
    H.info[9] = H.fuse_all ? 0 : (int64_t)H.ng.size();
    if (H.fuse_all) ibh_build_quads2(H.blocks, H.htab, H.etab, H.fus, H.nph[1], H.quads[0], nc);
    // Row sweep (ibh_rows2d.h): its halo cell ids are arithmetic on the neighbour block bases.  Taken only where that
    // arithmetic -- restated here -- reproduces the halo table (which was read off the registered faces) for every slot
    // of every block, every neighbour block is complete, and no block takes its deeper cells from a table.
    if (H.fuse_all && H.dtab.empty()) {
        bool ok = true;
        for (int32_t b = 0; b < nb && ok; ++b) {
            const BlockDesc2& B = H.blocks[b];
            for (int s = 0; s < 4 && ok; ++s) {
                const int d = s >> 1, sd = d == 0 ? 1 : 8, st = d == 0 ? 8 : 1;
                const bool low = (s & 1) == 0;
                const int opp = low ? 7 * sd : 0, own = low ? 0 : 7 * sd;
                const int ty = B.type[s];
                if (ty == SIDE_GENERAL) { ok = false; break; }
                if (ty != SIDE_MIRROR && (B.nb[s][0] < 0 || (ty == SIDE_FINE && B.nb[s][1] < 0))) { ok = false; break; }
                for (int t = 0; t < 8 && ok; ++t)
                    for (int k = 0; k < 2; ++k) {
                        const int tt = ty == SIDE_COARSE ? (t >> 1) + 4 * B.sub[s] : ty == SIDE_FINE ? 2 * (t & 3) + k : t;
                        const int32_t nbb = (ty == SIDE_FINE && (t >> 2)) ? B.nb[s][1] : B.nb[s][0];
                        const int32_t id = ty == SIDE_MIRROR ? B.base + own + t * st : nbb + opp + tt * st;
                        if (id != H.htab[(size_t)b * 64 + (s * 8 + t) * 2 + k]) { ok = false; break; }
                    }
            }
        }
        H.rows_ok = ok;
    }
    if (H.img_all_fz && !H.fuse_all) ibh_build_quads2(H.blocks, H.htab, H.etab, is_imgblk, H.nph[1], H.quads[1], nc);
}

static int make_view(HostPartView& v, int nd, int32_t nc, const float* spacing, const int32_t* nf,
                     const int32_t* const* owners, const int32_t* const* neighbors, const int32_t* const* left_off,
                     const int32_t* const* left_idx, const int32_t* const* right_off, const int32_t* const* right_idx,
                     const int32_t* domain, int block_size, int index_base) {
    v.nd = nd;
    v.nc = nc;
    v.spacing = spacing;
    v.nf = nf;
    v.domain = domain;
    v.index_base = index_base;
    v.bs = block_size;
    v.owners.resize(nd); v.neighbors.resize(nd);
    v.loff.resize(nd); v.lidx.resize(nd); v.roff.resize(nd); v.ridx.resize(nd);
    for (int d = 0; d < nd; ++d) {
        IBH_REQUIRE(nf[d] >= 0, "ibh_partition_create: negative face count");
        v.owners[d] = rebased(owners[d], nf[d], index_base);
        v.neighbors[d] = rebased(neighbors[d], nf[d], index_base);
        // offsets: accept either base (first entry tells)
        int32_t ob = left_off[d][0];
        v.loff[d] = rebased(left_off[d], (size_t)nc + 1, ob);
        v.lidx[d] = rebased(left_idx[d], v.loff[d][nc], index_base);
        ob = right_off[d][0];
        v.roff[d] = rebased(right_off[d], (size_t)nc + 1, ob);
        v.ridx[d] = rebased(right_idx[d], v.roff[d][nc], index_base);
        for (int32_t f = 0; f < nf[d]; ++f)
            IBH_REQUIRE(v.owners[d][f] >= 0 && v.owners[d][f] < nc && v.neighbors[d][f] >= 0 && v.neighbors[d][f] < nc,
                        "ibh_partition_create: owner/neighbour index out of range");
        for (int32_t c = 0; c < nc; ++c)
            IBH_REQUIRE(v.loff[d][c + 1] >= v.loff[d][c] && v.roff[d][c + 1] >= v.roff[d][c],
                        "ibh_partition_create: CSR offsets not monotone");
        for (int32_t x : v.lidx[d]) IBH_REQUIRE(x >= 0 && x < nf[d], "ibh_partition_create: left face id out of range");
        for (int32_t x : v.ridx[d]) IBH_REQUIRE(x >= 0 && x < nf[d], "ibh_partition_create: right face id out of range");
    }
    return 0;
}

extern "C" {

int ibh_version(void) { return 100; }

const char* ibh_last_error(void) { return ibh_err.c_str(); }

int ibh_init(int device) {
    IBH_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    IBH_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return ibh_fail(-2, (std::string("libibhip is built for gfx950 only, device is ") + prop.gcnArchName).c_str(),
                        __FILE__, __LINE__);
    return 0;
}

int ibh_set_stream(void* s) {
    ibh_stream = (hipStream_t)s;
    return 0;
}

int ibh_sync(void) {
    IBH_HIP(hipStreamSynchronize(ibh_stream));
    return 0;
}

int ibh_malloc(void** p, size_t bytes) {
    IBH_HIP(hipMalloc(p, bytes ? bytes : 4));
    return 0;
}
int ibh_free(void* p) {
    if (p) IBH_HIP(hipFree(p));
    return 0;
}
int ibh_h2d(void* dst, const void* src, size_t bytes) {
    IBH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ibh_stream));
    IBH_HIP(hipStreamSynchronize(ibh_stream));
    return 0;
}
int ibh_d2h(void* dst, const void* src, size_t bytes) {
    IBH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ibh_stream));
    IBH_HIP(hipStreamSynchronize(ibh_stream));
    return 0;
}
int ibh_memset(void* dst, int value, size_t bytes) {
    IBH_HIP(hipMemsetAsync(dst, value, bytes, ibh_stream));
    return 0;
}


int ibh_partition_create(ibh_part** out, int nd, int32_t nc, const float* spacing, const float* centers,
                         const int32_t* nf, const int32_t* const* owners, const int32_t* const* neighbors,
                         const int32_t* const* left_off, const int32_t* const* left_idx,
                         const int32_t* const* right_off, const int32_t* const* right_idx, int32_t n_image,
                         const int32_t* image_in_domain, const int32_t* domain, int block_size, int index_base) {
    IBH_REQUIRE(out && spacing && nf && owners && neighbors && left_off && left_idx && right_off && right_idx,
                "ibh_partition_create: null argument");
    IBH_REQUIRE(nd == 2 || nd == 3, "ibh_partition_create: nd must be 2 or 3");
    IBH_REQUIRE(nc >= 0 && (index_base == 0 || index_base == 1), "ibh_partition_create: bad nc/index_base");
    HostPartView v;
    int rc = make_view(v, nd, nc, spacing, nf, owners, neighbors, left_off, left_idx, right_off, right_idx, domain,
                       block_size, index_base);
    if (rc) return rc;
    ibh_part* p = new ibh_part();
    p->nd = nd;
    p->nc = nc;
    if ((rc = ibh_upload(&p->spacing, spacing, (size_t)nc * nd))) return rc;
    if (centers && (rc = ibh_upload(&p->centers, centers, (size_t)nc * nd))) return rc;
    for (int d = 0; d < nd; ++d) {
        DimData& D = p->dim[d];
        D.nf = nf[d];
        if ((rc = ibh_upload(&D.owners, v.owners[d].data(), v.owners[d].size()))) return rc;
        if ((rc = ibh_upload(&D.neighbors, v.neighbors[d].data(), v.neighbors[d].size()))) return rc;
        if ((rc = ibh_upload(&D.loff, v.loff[d].data(), v.loff[d].size()))) return rc;
        if ((rc = ibh_upload(&D.lidx, v.lidx[d].data(), v.lidx[d].size()))) return rc;
        if ((rc = ibh_upload(&D.roff, v.roff[d].data(), v.roff[d].size()))) return rc;
        if ((rc = ibh_upload(&D.ridx, v.ridx[d].data(), v.ridx[d].size()))) return rc;
    }
    {
        std::vector<int32_t> side((size_t)2 * nd * nc);
        int64_t direct = 0;
        for (int d = 0; d < nd; ++d)
            for (int s = 0; s < 2; ++s) {
                const std::vector<int32_t>& off = s ? v.roff[d] : v.loff[d];
                const std::vector<int32_t>& idx = s ? v.ridx[d] : v.lidx[d];
                int32_t* T = side.data() + (size_t)(2 * d + s) * nc;
                for (int32_t c = 0; c < nc; ++c) {
                    const int32_t b = off[c], e = off[c + 1];
                    int32_t t = -1;
                    if (e == b) t = -2;
                    else if (e == b + 1) {
                        const int32_t f = idx[b];
                        const int32_t o = v.owners[d][f], n = v.neighbors[d][f];
                        if (s ? (o == c) : (n == c)) t = s ? n : o;      // right faces: c owns; left faces: c is the neighbour
                    }
                    T[c] = t;
                    direct += t >= 0;
                }
            }
        if ((rc = ibh_upload(&p->side, side.data(), side.size()))) return rc;
        p->info[17] = direct;
    }
    p->n_image = n_image;
    if (n_image > 0 && image_in_domain) {
        std::vector<int32_t> iid = rebased(image_in_domain, n_image, index_base);
        for (int32_t x : iid) IBH_REQUIRE(x >= 0 && x < nc, "ibh_partition_create: image_in_domain out of range");
        if ((rc = ibh_upload(&p->image_in_domain, iid.data(), iid.size()))) return rc;
    }
    p->bs = 0;
    if (domain && block_size > 0 && nd == 2 && block_size == 8) {
        Host2D H;
        analyze2_host(v, n_image, image_in_domain, index_base, H);
        for (int i = 0; i < 12; ++i) p->info[i] = H.info[i];
        if ((rc = ibh_upload(&p->htab, H.htab.data(), H.htab.size()))) return rc;
        if ((rc = ibh_upload(&p->etab, H.etab.data(), H.etab.size()))) return rc;
        if ((rc = ibh_upload(&p->dtab, H.dtab.data(), H.dtab.size()))) return rc;
        p->n_dt = (int32_t)(H.dtab.size() / 64);
        p->fuse_all = H.fuse_all;
        p->rows_ok = H.rows_ok;
        p->info[16] = H.rows_ok;
        p->n_img = (int32_t)H.img.size();
        p->n_img_int = H.n_img_int;
        p->img_all_fz = H.img_all_fz;
        if (p->img_all_fz && (rc = ibh_upload(&p->img_list, H.img.data(), H.img.size()))) return rc;
        if (!p->fuse_all && H.info[8] > 0) {
            p->n_fz = (int32_t)H.fz.size();
            p->n_ng = (int32_t)H.ng.size();
            p->n_nf = (int32_t)H.nf.size();
            p->n_fz_int = H.n_fz_int;
            p->n_ng_int = H.n_ng_int;
            p->n_nf_int = H.n_nf_int;
            if ((rc = ibh_upload(&p->fz_list, H.fz.data(), H.fz.size()))) return rc;
            if ((rc = ibh_upload(&p->ng_list, H.ng.data(), H.ng.size()))) return rc;
            if ((rc = ibh_upload(&p->nf_list, H.nf.data(), H.nf.size()))) return rc;
        }
        for (int k = 0; k < 2; ++k) {
            const QuadSet2& Q = H.quads[k];
            p->nq[k] = (int32_t)Q.qd.size();
            p->nq_int[k] = Q.nq_int;
            p->nqs[k] = (int32_t)Q.singles.size();
            p->nqs_int[k] = Q.ns_int;
            p->info[12 + 2 * k] = p->nq[k];
            p->info[13 + 2 * k] = p->nqs[k];
            if (k == 0 && !Q.pd.empty()) {
                // pair tiles behind the quads, in the same arrays (the kernel tells them apart by index)
                std::vector<QuadDesc2> qd(Q.qd);
                qd.insert(qd.end(), Q.pd.begin(), Q.pd.end());
                std::vector<int32_t> qt(Q.qtab);
                qt.insert(qt.end(), Q.ptab.begin(), Q.ptab.end());
                p->npair = (int32_t)Q.pd.size();
                p->nqs2 = (int32_t)Q.singles2.size();
                p->info[18] = p->npair;
                if ((rc = ibh_upload(&p->qd[k], qd.data(), qd.size()))) return rc;
                if ((rc = ibh_upload(&p->qtab[k], qt.data(), qt.size()))) return rc;
                if ((rc = ibh_upload(&p->qsingles2, Q.singles2.data(), Q.singles2.size()))) return rc;
                std::vector<int32_t> qa(Q.qaux);
                qa.insert(qa.end(), Q.paux.begin(), Q.paux.end());
                if ((rc = ibh_upload(&p->qaux[k], qa.data(), qa.size()))) return rc;
            } else {
                if ((rc = ibh_upload(&p->qd[k], Q.qd.data(), Q.qd.size()))) return rc;
                if ((rc = ibh_upload(&p->qtab[k], Q.qtab.data(), Q.qtab.size()))) return rc;
                if ((rc = ibh_upload(&p->qaux[k], Q.qaux.data(), Q.qaux.size()))) return rc;
            }
            {
                int64_t arith = 0;
                for (size_t i = 0; i < Q.qaux.size(); i += IBH_QAUX)
                    for (int l = 0; l < 8; ++l) arith += Q.qaux[i + 32 + l] >= 0;
                if (k == 0) p->info[19] = arith;     // half-sides of the quads whose halo ids are arithmetic
            }
            if ((rc = ibh_upload(&p->qsingles[k], Q.singles.data(), Q.singles.size()))) return rc;
        }
        p->nA1 = H.nph[0];
        p->nB1 = H.nph[1];
        p->bs = block_size;
        p->nblk = (int32_t)H.blocks.size();
        p->n_irr = (int32_t)H.irr.size();
        {   // complete blocks in order: cell c lies in block c / 64 at position c % 64 (face-list kernels take in-block
            // neighbours by index arithmetic then: k_timestep_advection<TILED>)
            bool tiled = !H.blocks.empty() && H.irr.empty() && (int64_t)H.blocks.size() * 64 == (int64_t)nc;
            for (size_t b = 0; tiled && b < H.blocks.size(); ++b) tiled = H.blocks[b].base == (int32_t)(64 * b);
            p->info[20] = tiled ? 1 : 0;
        }
        if ((rc = ibh_upload(&p->blocks2, H.blocks.data(), H.blocks.size()))) return rc;
        if ((rc = ibh_upload(&p->irr_cells, H.irr.data(), H.irr.size()))) return rc;
    } else if (domain && block_size == 8 && nd == 3) {
        std::vector<BlockDesc3> blocks;
        std::vector<int32_t> irr, htab, ftab;
        int32_t nph[2] = {0, 0};
        std::vector<int32_t> none;
        const bool have_img = n_image > 0 && image_in_domain;
        Sweep3Host sw;
        ibh_analyze_blocks3(v, blocks, irr, p->info, have_img ? image_in_domain : none.data(), have_img ? n_image : 0,
                            nph, htab, ftab, &sw);
        p->sweep3 = sw.all ? 1 : 0;
        p->info[8] = sw.all ? (int64_t)blocks.size() : 0;   // blocks of the single-kernel sweep
        p->info[9] = sw.all ? 0 : (int64_t)blocks.size();   // blocks whose gradients go through the workspace
        p->info[12] = (int64_t)(sw.r4tab.size() / 4);         // rim neighbours made of four finer cells
        if (sw.r4tab.empty()) sw.r4tab.assign(4, 0);
        if ((rc = ibh_upload(&p->rtab3, sw.rtab.data(), sw.rtab.size()))) return rc;
        if ((rc = ibh_upload(&p->r4tab3, sw.r4tab.data(), sw.r4tab.size()))) return rc;
        if ((rc = ibh_upload(&p->ftab3, ftab.data(), ftab.size()))) return rc;
        p->nA1 = nph[0];
        p->nB1 = nph[1];
        p->bs = block_size;
        p->nblk = (int32_t)blocks.size();
        p->n_irr = (int32_t)irr.size();
        {
            bool tiled = !blocks.empty() && irr.empty() && (int64_t)blocks.size() * 512 == (int64_t)nc;
            for (size_t b = 0; tiled && b < blocks.size(); ++b) tiled = blocks[b].base == (int32_t)(512 * b);
            p->info[20] = tiled ? 1 : 0;
        }
        if ((rc = ibh_upload(&p->blocks3, blocks.data(), blocks.size()))) return rc;
        if ((rc = ibh_upload(&p->htab3, htab.data(), htab.size()))) return rc;
        if (have_img && !sw.all) {   // a partition with skirt fragments: the image blocks for the image-only sweeps
            Image3Host im;
            ibh_analyze_image3(v, image_in_domain, n_image, im);
            if (im.all) {
                if (im.r4tab.empty()) im.r4tab.assign(4, 0);
                if (im.ftab.empty()) im.ftab.assign(4, 0);
                if ((rc = ibh_upload(&p->iblocks3, im.blocks.data(), im.blocks.size()))) return rc;
                if ((rc = ibh_upload(&p->ihtab3, im.htab.data(), im.htab.size()))) return rc;
                if ((rc = ibh_upload(&p->iftab3, im.ftab.data(), im.ftab.size()))) return rc;
                if ((rc = ibh_upload(&p->irtab3, im.rtab.data(), im.rtab.size()))) return rc;
                if ((rc = ibh_upload(&p->ir4tab3, im.r4tab.data(), im.r4tab.size()))) return rc;
                if ((rc = ibh_upload(&p->idtab3, im.dtab.data(), im.dtab.size()))) return rc;
                p->n_img3 = (int32_t)im.blocks.size();
                p->img_all3 = 1;
            }
        }
        p->info[10] = p->img_all3 || sw.all;       // image blocks all eligible for the single-kernel sweeps
        p->info[11] = p->img_all3 ? p->n_img3 : (sw.all ? (int64_t)blocks.size() : 0);
        if ((rc = ibh_upload(&p->irr_cells, irr.data(), irr.size()))) return rc;
    } else {
        p->info[0] = 0;
        p->info[1] = nc;
    }
    // flattened stencil records of the face-list cells
    if (p->n_irr > 0) {
        std::vector<int32_t> irr(p->n_irr);
        IBH_HIP(hipMemcpy(irr.data(), p->irr_cells, sizeof(int32_t) * p->n_irr, hipMemcpyDeviceToHost));
        const size_t n = (size_t)p->n_irr;
        std::vector<int32_t> rec((size_t)nd * 2 * 5 * n, 0);
        bool ok = true;
        for (size_t t = 0; t < n && ok; ++t) {
            const int32_t cc = irr[t];
            for (int d = 0; d < nd && ok; ++d)
                for (int side = 0; side < 2 && ok; ++side) {
                    const std::vector<int32_t>& off = side ? v.roff[d] : v.loff[d];
                    const std::vector<int32_t>& idx = side ? v.ridx[d] : v.lidx[d];
                    const int cnt = off[cc + 1] - off[cc];
                    if (cnt > 4) { ok = false; break; }
                    const size_t q = (size_t)(2 * d + side);
                    rec[(q * 5) * n + t] = cnt;
                    for (int k = 0; k < cnt; ++k) {
                        const int32_t f = idx[off[cc] + k];
                        const int32_t o = v.owners[d][f], nn = v.neighbors[d][f];
                        // left face: this cell is the neighbour, the other cell is the owner (and vice versa)
                        if ((side ? o : nn) != cc) { ok = false; break; }
                        rec[(q * 5 + 1 + k) * n + t] = side ? nn : o;
                    }
                }
        }
        if (ok && (rc = ibh_upload(&p->irr_rec, rec.data(), rec.size()))) return rc;
    }
    *out = p;
    return 0;
}

// ---- host-only introspection of the 2-D block analysis (no HIP call: runs without a GPU) ----
int ibh_analyze2_host(ibh_host2d** out, int32_t nc, const float* spacing, const int32_t* nf,
                      const int32_t* const* owners, const int32_t* const* neighbors, const int32_t* const* left_off,
                      const int32_t* const* left_idx, const int32_t* const* right_off, const int32_t* const* right_idx,
                      int32_t n_image, const int32_t* image_in_domain, const int32_t* domain, int index_base) {
    IBH_REQUIRE(out && spacing && nf && owners && neighbors && left_off && left_idx && right_off && right_idx && domain,
                "ibh_analyze2_host: null argument");
    HostPartView v;
    int rc = make_view(v, 2, nc, spacing, nf, owners, neighbors, left_off, left_idx, right_off, right_idx, domain, 8,
                       index_base);
    if (rc) return rc;
    ibh_host2d* h = new ibh_host2d();
    analyze2_host(v, n_image, image_in_domain, index_base, h->H);
    *out = h;
    return 0;
}

int ibh_host2d_get(const ibh_host2d* h, int what, int set, void* dst, int64_t cap_bytes, int64_t* nbytes) {
    IBH_REQUIRE(h && nbytes && (set == 0 || set == 1), "ibh_host2d_get: bad argument");
    const Host2D& H = h->H;
    const QuadSet2& Q = H.quads[set];
    const void* src = nullptr;
    int64_t n = 0;
    int64_t counts[8] = {(int64_t)H.blocks.size(), (int64_t)Q.qd.size(), Q.nq_int, (int64_t)Q.singles.size(), Q.ns_int,
                         H.fuse_all, H.img_all_fz, H.nph[1]};
    switch (what) {
        case IBH_H2D_BLOCKS: src = H.blocks.data(); n = (int64_t)(H.blocks.size() * sizeof(BlockDesc2)); break;
        case IBH_H2D_HTAB: src = H.htab.data(); n = (int64_t)(H.htab.size() * 4); break;
        case IBH_H2D_ETAB: src = H.etab.data(); n = (int64_t)(H.etab.size() * 4); break;
        case IBH_H2D_FUSABLE: src = H.fus.data(); n = (int64_t)H.fus.size(); break;
        case IBH_H2D_QUAD_DESC: src = Q.qd.data(); n = (int64_t)(Q.qd.size() * sizeof(QuadDesc2)); break;
        case IBH_H2D_QUAD_TAB: src = Q.qtab.data(); n = (int64_t)(Q.qtab.size() * 4); break;
        case IBH_H2D_SINGLES: src = Q.singles.data(); n = (int64_t)(Q.singles.size() * 4); break;
        case IBH_H2D_PAIR_DESC: src = Q.pd.data(); n = (int64_t)(Q.pd.size() * sizeof(QuadDesc2)); break;
        case IBH_H2D_PAIR_TAB: src = Q.ptab.data(); n = (int64_t)(Q.ptab.size() * 4); break;
        case IBH_H2D_SINGLES2: src = Q.singles2.data(); n = (int64_t)(Q.singles2.size() * 4); break;
        case IBH_H2D_QUAD_AUX: src = Q.qaux.data(); n = (int64_t)(Q.qaux.size() * 4); break;
        case IBH_H2D_PAIR_AUX: src = Q.paux.data(); n = (int64_t)(Q.paux.size() * 4); break;
        case IBH_H2D_COUNTS: src = counts; n = (int64_t)sizeof(counts); break;
        case IBH_H2D_INFO: src = H.info; n = (int64_t)sizeof(H.info); break;
        default: return ibh_fail(-1, "ibh_host2d_get: unknown item", __FILE__, __LINE__);
    }
    *nbytes = n;
    if (dst) {
        IBH_REQUIRE(cap_bytes >= n, "ibh_host2d_get: destination too small");
        if (n) memcpy(dst, src, (size_t)n);
    }
    return 0;
}

int ibh_host2d_destroy(ibh_host2d* h) {
    delete h;
    return 0;
}

int ibh_partition_destroy(ibh_part* p) {
    if (!p) return 0;
    hipFree(p->spacing);
    hipFree(p->centers);
    for (int d = 0; d < p->nd; ++d) {
        DimData& D = p->dim[d];
        hipFree(D.owners); hipFree(D.neighbors);
        hipFree(D.loff); hipFree(D.lidx); hipFree(D.roff); hipFree(D.ridx);
    }
    hipFree(p->image_in_domain);
    hipFree(p->side);
    hipFree(p->blocks2);
    hipFree(p->htab);
    hipFree(p->etab);
    hipFree(p->dtab);
    hipFree(p->img_list);
    hipFree(p->march_tmp);
    for (int k = 0; k < 2; ++k) {
        hipFree(p->qd[k]);
        hipFree(p->qtab[k]);
        hipFree(p->qaux[k]);
        hipFree(p->qsingles[k]);
        if (k == 0) hipFree(p->qsingles2);
    }
    hipFree(p->fz_list);
    hipFree(p->ng_list);
    hipFree(p->nf_list);
    hipFree(p->blocks3);
    hipFree(p->htab3);
    hipFree(p->iblocks3); hipFree(p->ihtab3); hipFree(p->iftab3); hipFree(p->irtab3); hipFree(p->ir4tab3);
    hipFree(p->idtab3);
    hipFree(p->ftab3);
    hipFree(p->rtab3);
    hipFree(p->r4tab3);
    hipFree(p->irr_cells);
    hipFree(p->irr_rec);
    hipFree(p->G);
    delete p;
    return 0;
}

int ibh_partition_info(const ibh_part* p, int64_t* info, int n) {
    IBH_REQUIRE(p && info, "ibh_partition_info: null argument");
    for (int i = 0; i < n && i < 24; ++i) info[i] = p->info[i];
    return 0;
}

static int acc_fill(ibh_acc* a, int32_t n_output, int32_t n_input, const int32_t* off, const int32_t* idx,
                    const float* w, int index_base) {
    IBH_REQUIRE(off && (n_output >= 0), "ibh_acc_create: null offsets");
    int32_t ob = off[0];
    std::vector<int32_t> o = rebased(off, (size_t)n_output + 1, ob);
    for (int32_t r = 0; r < n_output; ++r) IBH_REQUIRE(o[r + 1] >= o[r], "ibh_acc_create: offsets not monotone");
    size_t nnz = o[n_output];
    IBH_REQUIRE(nnz == 0 || idx, "ibh_acc_create: null indices");
    std::vector<int32_t> ix = rebased(idx, nnz, index_base);
    for (int32_t x : ix) IBH_REQUIRE(x >= 0 && x < n_input, "ibh_acc_create: donor index out of range");
    a->n_out = n_output;
    a->n_in = n_input;
    int rc;
    if ((rc = ibh_upload(&a->off, o.data(), o.size()))) return rc;
    if ((rc = ibh_upload(&a->idx, ix.data(), ix.size()))) return rc;
    a->w = nullptr;
    if (w && (rc = ibh_upload(&a->w, w, nnz))) return rc;
    return 0;
}

int ibh_acc_create(ibh_acc** out, int32_t n_output, int32_t n_input, const int32_t* off, const int32_t* idx,
                   const float* w, int index_base) {
    IBH_REQUIRE(out, "ibh_acc_create: null out");
    ibh_acc* a = new ibh_acc();
    int rc = acc_fill(a, n_output, n_input, off, idx, w, index_base);
    if (rc) { delete a; return rc; }
    *out = a;
    return 0;
}

int ibh_acc_destroy(ibh_acc* a) {
    if (!a) return 0;
    hipFree(a->off); hipFree(a->idx); hipFree(a->w); hipFree(a->packed);
    delete a;
    return 0;
}

int ibh_bc_create(ibh_bc** out, int32_t ng, const int32_t* ghost_indices, const float* ghost_distances,
                  const float* image_distances, int32_t nid, const int32_t* image_domain,
                  const int32_t* interp_off, const int32_t* interp_idx, const float* interp_w, int index_base) {
    IBH_REQUIRE(out && ghost_indices && ghost_distances && image_distances && image_domain,
                "ibh_bc_create: null argument");
    ibh_bc* b = new ibh_bc();
    b->ng = ng;
    b->nid = nid;
    std::vector<int32_t> g = rebased(ghost_indices, ng, index_base);
    std::vector<int32_t> idm = rebased(image_domain, nid, index_base);
    std::vector<float> eta(ng);
    for (int32_t i = 0; i < ng; ++i) eta[i] = ghost_distances[i] / image_distances[i];  // :1220
    int rc;
    if ((rc = ibh_upload(&b->ghost, g.data(), g.size()))) return rc;
    if ((rc = ibh_upload(&b->image_domain, idm.data(), idm.size()))) return rc;
    if ((rc = ibh_upload(&b->eta, eta.data(), eta.size()))) return rc;
    if ((rc = acc_fill(&b->interp, ng, nid, interp_off, interp_idx, interp_w, index_base))) return rc;
    // host copies (ibh_bcset_create)
    b->h_ghost = g;
    b->h_eta = eta;
    if (ng > 0 && interp_off && interp_idx) {
        const int32_t ob = interp_off[0];
        b->h_off.resize((size_t)ng + 1);
        for (int32_t i = 0; i <= ng; ++i) b->h_off[i] = interp_off[i] - ob;
        const size_t nnz = (size_t)b->h_off[ng];
        b->h_donor.resize(nnz);
        b->h_w.resize(nnz);
        for (size_t k = 0; k < nnz; ++k) {
            const int32_t j = interp_idx[k] - index_base;
            IBH_REQUIRE(j >= 0 && j < nid, "ibh_bc_create: stencil index out of range");
            b->h_donor[k] = idm[j];
            b->h_w[k] = interp_w ? interp_w[k] : 1.0f;
        }
    }
    *out = b;
    return 0;
}

int ibh_bc_destroy(ibh_bc* b) {
    if (!b) return 0;
    hipFree(b->ghost); hipFree(b->image_domain); hipFree(b->eta);
    hipFree(b->interp.off); hipFree(b->interp.idx); hipFree(b->interp.w);
    delete b;
    return 0;
}

}  // extern "C"
