// Internal declarations shared by the libibhip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/ibhip.h"

#define IBH_MAXD 3

// Block-side classes of the block-structured fast path (see ibh_analyze.cpp).
enum : int32_t { SIDE_SAME = 0, SIDE_MIRROR = 1, SIDE_COARSE = 2, SIDE_FINE = 3, SIDE_GENERAL = 4 };

// One full block (all bs^nd cells present, consecutive local ids starting at `base`).
// side s = 2*d + (0: low / left, 1: high / right).
// nb[s][k]: local id of the first cell (`base`) of the k-th neighbour block across side s
//   SAME:   nb[s][0]
//   COARSE: nb[s][0], and sub[s] = which half (2-D) / quadrant (3-D) of the coarse side we abut
//   FINE:   nb[s][0..2^(nd-1)-1], ordered x-fastest over the tangential directions
struct BlockDesc2 {  // 2-D
    int32_t base;
    int32_t type[4];
    int32_t nb[4][2];
    int32_t sub[4];
    float h[2];
    float rh[2];   // 1/h
    float q[4];    // per side: 1/(1 + h_nb/h) = 1/2 (same, mirror), 1/3 (coarse), 2/3 (fine)
    float rt[4];   // per side: h_nb/h = 1, 2, 1/2
    int32_t dt;    // single-kernel sweep: row of the deeper-cell table, or -1 (deeper cell = halo id -/+ 1 or 8)
};
// Per block: 64 halo slots, slot = (side*8 + t)*2 + k -> local id of the k-th neighbour cell across `side`
// of boundary cell t.  Always a valid cell: single-face sides repeat sub-face 0 in slot k=1 (so that
// averaging the two sub-faces is exact), MIRROR sides name the boundary cell itself, GENERAL sides too
// (their lanes are handled by the face-list body and never stored).

// A "quad": 2x2 same-level complete blocks with consecutive bases (base, +64, +128, +192 = lower-left, lower-right,
// upper-left, upper-right: the sibling order of the quadtree, mesher.jl:845-861) whose 8 outer half-sides are SAME /
// COARSE / FINE.  One wavefront sweeps the 16x16 tile with 4 cells per lane (ibh_quad2d.h).
// cls: 4 bits per outer half-side, index l = 2*g + half with g = 0 left (side 0), 1 bottom (side 2), 2 top (side 3),
// 3 right (side 1) -- g is the 16-lane row that owns the side's halo slots -- and half = 0 for boundary cells 0..7.
struct QuadDesc2 {
    int32_t base;
    uint32_t cls;
    float rh[2];
};
// Per quad one row of IBH_QROW ints: [0,128) halo cell ids, entry 2*lane + k (lane = 16*g + t: side g, boundary cell t,
// sub-face k); [128,160) end ids, entry 4*l + e (e = 0,1 low end k = 0,1; 2,3 high end) -- the per-block halo / end
// tables of ibh_analyze.cpp re-indexed for the quad's lanes.
#define IBH_QROW 160
// Compact companion row (IBH_QAUX ints per quad / pair): [0,32) the end ids of the row above, [32,40) per half-side l the
// halo cell id of its boundary cell 0 where the ids of the half-side are ARITHMETIC -- SAME: id(t) = orig + stride t,
// COARSE: id(t) = orig + stride (t >> 1), stride = 8 on left / right sides, 1 on bottom / top, both sub-face slots the same
// cell (verified against the row by the builder) -- or -1: FINE half-sides and anything else read the 128 ids of the row.
// A sweep that takes its ids from here touches 176 bytes of tables per quad instead of 656.
#define IBH_QAUX 40
struct QuadSet2 {
    std::vector<QuadDesc2> qd;     // interior-phase quads first (n_int of them)
    std::vector<int32_t> qtab;     // [nq][IBH_QROW]
    std::vector<int32_t> qaux, paux;  // [nq][IBH_QAUX], [npair][IBH_QAUX]
    std::vector<int32_t> singles;  // candidate blocks outside quads (block table indices), interior-phase ones first
    int32_t nq_int = 0, ns_int = 0;
    // pairs: two blocks of `singles` side by side in x with consecutive bases (a 16 x 8 tile swept by one wave of the quad
    // kernel in its HALF form); built only where every block is an interior-phase block (one partition, no skirt).
    // Descriptors / rows in the quad format (unused half-sides repeat a used one); singles2 = singles without them.
    std::vector<QuadDesc2> pd;
    std::vector<int32_t> ptab;
    std::vector<int32_t> singles2;
};

// 3-D: one full 8x8x8 block (512 consecutive local ids).  side s = 2*d + (0 low / 1 high); boundary cell of a
// side indexed t = t1 + 8*t2 over the two tangential dims in increasing order.  The 3-D fast path takes
// SAME / MIRROR / COARSE (2:1) sides; sides facing finer blocks are GENERAL (face-list body), so every
// boundary cell has exactly one face and the halo table has one slot per (side, t): htab3[blk][s*64 + t].
// FINE sides (4 faces per boundary cell, sub-face k = k1 + 2*k2): sub-face 0 sits in the halo table, sub-faces
// 1..3 in ftab3[fine][s][t][k-1]; only the boundary threads of such a side touch them (wave-uniform branch).
struct BlockDesc3 {
    int32_t base;
    int32_t fine;     // index into the fine-side table (ftab3) or -1: blocks with a side facing finer blocks
    int32_t type[6];
    int32_t nb[6];
    int32_t sub[6];
    float h[3];
    float rh[3];
    float q[6];
    float rt[6];
};

struct DimData {
    int32_t nf = 0;
    int32_t *owners = nullptr, *neighbors = nullptr;  // [nf] 0-based
    int32_t *loff = nullptr, *lidx = nullptr;         // CSR of left faces  (cell is the neighbour)
    int32_t *roff = nullptr, *ridx = nullptr;         // CSR of right faces (cell is the owner)
};

struct ibh_part {
    int nd = 0;
    int32_t nc = 0;
    float* spacing = nullptr;  // (nc, nd) column-major
    float* centers = nullptr;  // (nc, nd) or null
    DimData dim[IBH_MAXD];
    // side table [2 * nd][nc]: entry (2 d + s, c) = the cell across the ONE face cell c has on its left (s = 0) / right
    // (s = 1) in dimension d; -2 = no face on that side, -1 = anything else (several faces, a face that does not list c
    // where the CSR says): the face-list kernels take the direct form for entries >= 0 and the CSR walk otherwise
    int32_t* side = nullptr;
    int32_t n_image = 0;
    int32_t* image_in_domain = nullptr;
    // block-structured fast path
    int bs = 0;
    int32_t nblk = 0;            // full blocks handled by the fast kernels
    int32_t nA1 = 0, nB1 = 0;    // blocks [0,nA1): pass A independent of skirt data; [0,nB1): pass B too
    BlockDesc2* blocks2 = nullptr;
    int32_t* htab = nullptr;     // [nblk][64] halo cell table, same order as blocks2
    int32_t* etab = nullptr;     // [nblk][16] end table of the single-kernel sweep (see ibh_analyze.cpp step 6)
    int32_t* dtab = nullptr;     // [n_dt][64] deeper-cell rows for blocks next to skirt fragments (BlockDesc2::dt)
    int32_t n_dt = 0;
    // image-only sweeps (IBH_IMAGE_ONLY): every image block eligible -> one launch per phase, nothing else
    int32_t* img_list = nullptr; // image blocks, ascending, the first n_img_int of them < nB1
    int32_t n_img = 0, n_img_int = 0, img_all_fz = 0;
    int32_t fuse_all = 0;        // 1: every block is eligible for the single-kernel sweep and there are no face-list cells
    int32_t rows_ok = 0;         // 1: fuse_all and the arithmetic halo ids of the row sweep (ibh_rows2d.h) reproduce the halo table
    // quad sweeps (ibh_quad2d.h): set 0 = all blocks (used when fuse_all), set 1 = image blocks (used when img_all_fz)
    QuadDesc2* qd[2] = {nullptr, nullptr};
    int32_t* qtab[2] = {nullptr, nullptr};
    int32_t* qaux[2] = {nullptr, nullptr};
    int32_t* qsingles[2] = {nullptr, nullptr};
    int32_t nq[2] = {0, 0}, nq_int[2] = {0, 0}, nqs[2] = {0, 0}, nqs_int[2] = {0, 0};
    int32_t npair = 0, nqs2 = 0;          // set 0 only: pair tiles (stored behind the quads in qd / qtab), blocks left over
    int32_t* qsingles2 = nullptr;
    // mixed launches (fuse_all == 0): ascending block indices, interior-phase entries first
    int32_t* fz_list = nullptr;  // eligible blocks                       [n_fz], the first n_fz_int of them < nB1
    int32_t* ng_list = nullptr;  // blocks whose gradients somebody reads [n_ng], the first n_ng_int of them < nA1
    int32_t* nf_list = nullptr;  // blocks left to the two-kernel form    [n_nf], the first n_nf_int of them < nB1
    int32_t n_fz = 0, n_fz_int = 0, n_ng = 0, n_ng_int = 0, n_nf = 0, n_nf_int = 0;
    BlockDesc3* blocks3 = nullptr;  // 3-D block table (nd == 3)
    int32_t* htab3 = nullptr;    // [nblk][384]
    int32_t* ftab3 = nullptr;    // [nfine][6][64][3] sub-faces 1..3 of FINE sides
    // 3-D single-kernel sweep: used when every block qualifies (sweep3 != 0)
    int32_t *rtab3 = nullptr, *r4tab3 = nullptr;
    int32_t sweep3 = 0;
    // 3-D image-only single-kernel sweeps (IBH_IMAGE_ONLY on a partition with skirt fragments): the image blocks with
    // their own tables incl. the deeper-cell table (img_all3 != 0: every image block qualifies)
    BlockDesc3* iblocks3 = nullptr;
    int32_t *ihtab3 = nullptr, *iftab3 = nullptr, *irtab3 = nullptr, *ir4tab3 = nullptr, *idtab3 = nullptr;
    int32_t n_img3 = 0, img_all3 = 0;
    int32_t n_irr = 0;           // cells handled by the general kernels when the fast path is on
    int32_t* irr_cells = nullptr;
    // Flattened stencils of the face-list cells (built when every such cell has <= 4 faces per direction):
    // rec[(q*5)*n_irr + t] = number of faces of cell irr_cells[t] in direction q = 2*d + (0 left / 1 right),
    // rec[(q*5 + 1 + k)*n_irr + t] = the cell across its k-th face (accumulator order).  One coalesced read
    // replaces the offsets -> face ids -> owner/neighbour chain of the CSR walk.
    int32_t* irr_rec = nullptr;
    int64_t info[24] = {0};
    // workspace for per-cell gradients + sensor (pass A output)
    float* G = nullptr;
    float* march_tmp = nullptr;  // workgroup maxima of ibh_timestep_advection's reduction
    size_t march_tmp_n = 0;
    size_t G_bytes = 0;
};

struct ibh_acc {
    int32_t n_out = 0, n_in = 0;
    int32_t *off = nullptr, *idx = nullptr;
    float* w = nullptr;
    float* packed = nullptr;  // [n_in][8] scratch of ibh_accumulate_diff_add: the fields of a donor side by side
};

struct ibh_bc {
    int32_t ng = 0, nid = 0;
    int32_t *ghost = nullptr, *image_domain = nullptr;
    float* eta = nullptr;
    ibh_acc interp;
    // host copies for ibh_bcset_create (which boundaries may share a launch): ghost cells, eta, and the interpolation
    // stencils with the donor CELLS resolved (image_domain[idx])
    std::vector<int32_t> h_ghost, h_off, h_donor;
    std::vector<float> h_eta, h_w;
};

// An ordered list of ghost-cell boundary conditions with closures the library knows (mode 0: a constant value, mode 1:
// copy(u): the closures of test/advection.jl:30-46), concatenated on the device (ibh_march.hip)
#define IBH_MAX_BC 8
struct ibh_bcset {
    int nbc = 0, nlev = 0;
    int32_t ng = 0;                 // ghost cells of all boundaries, ordered by level
    int32_t seg[IBH_MAX_BC + 1] = {0};  // ghost ranges of the levels
    bool direct[IBH_MAX_BC] = {false};  // no ghost cell of the level is a donor of the level: blended straight into the field
    unsigned int* sync = nullptr;       // [4] one-launch form: barrier counter, finished workgroups, status, spare
    int32_t *ghost = nullptr, *off = nullptr, *donor = nullptr, *bidx = nullptr;
    float *eta = nullptr, *w = nullptr, *gval = nullptr, *value = nullptr;
    int32_t* mode = nullptr;
};

// internal: face-list forms of the fused turbulence closures (ibh_ops.hip), dispatched from ibh_fused.hip
extern "C" int ibh_shear_rate_of_velocity_cells(const ibh_part* p, const float* vel, int64_t ldv, float* S, float* Gout,
                                                int64_t ldg);
extern "C" int ibh_wray_agarwal_of_cells(const ibh_part* p, const float* R, const float* S, float sigmaR, float C1,
                                         float kappa, float* nut, float* nuR, float* Sout);

extern "C" int ibh_scalar_transport_blocks(const ibh_part* p, const float* R, const float* nuR, float nu, const float* vel,
                                           int64_t ldv, const float* S, float* out, int* done);

// thread-local state
extern thread_local std::string ibh_err;
extern thread_local hipStream_t ibh_stream;

int ibh_fail(int code, const char* what, const char* file, int line);

#define IBH_HIP(call)                                                        \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) return ibh_fail((int)e__, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
#define IBH_REQUIRE(cond, msg)                                               \
    do {                                                                     \
        if (!(cond)) return ibh_fail(-1, msg, __FILE__, __LINE__);           \
    } while (0)
#define IBH_LAUNCH_CHECK() IBH_HIP(hipGetLastError())

template <class T>
int ibh_upload(T** dptr, const T* host, size_t n);

// host-side block analysis (ibh_analyze.cpp)
struct HostPartView {
    int nd;
    int32_t nc;
    const float* spacing;
    const int32_t* nf;
    std::vector<std::vector<int32_t>> owners, neighbors, loff, lidx, roff, ridx;  // 0-based copies
    const int32_t* domain;  // global ids, base `index_base`
    int index_base;
    int bs;
};
void ibh_analyze_blocks2(const HostPartView& v, std::vector<BlockDesc2>& blocks,
                         std::vector<int32_t>& irr_cells, int64_t* info, const int32_t* image_in_domain,
                         int32_t n_image, int32_t* n_phase1, std::vector<int32_t>& htab,
                         std::vector<int32_t>& etab, std::vector<char>& fusable, std::vector<char>& needg,
                         std::vector<int32_t>& dtab);

// quads among the candidate blocks (cand[b] != 0); blocks < nB1 are interior-phase blocks
void ibh_build_quads2(const std::vector<BlockDesc2>& blocks, const std::vector<int32_t>& htab,
                      const std::vector<int32_t>& etab, const std::vector<char>& cand, int32_t nB1, QuadSet2& out,
                      int32_t nc);

// tables of the 3-D single-kernel sweep (blk3::sweep_adv), see ibh_analyze3.cpp
struct Sweep3Host {
    bool all = false;            // every block qualifies: the sweep is one launch, nothing through the workspace
    std::vector<int32_t> rtab;   // [nblk][6][64] rim table
    std::vector<int32_t> r4tab;  // [n][4] rim neighbours that are four finer cells
};
// image blocks of a partition with skirt fragments and their tables (ibh_analyze3_image.cpp)
struct Image3Host {
    bool all = false;            // every image block qualifies for the single-kernel sweeps
    std::vector<BlockDesc3> blocks;
    std::vector<int32_t> htab, ftab, rtab, r4tab;  // as htab3 / ftab3 / rtab3 / r4tab3, rows per image block
    std::vector<int32_t> dtab;   // [nblk][6][64][4] the cell one step deeper behind halo cell k of slot t
};
void ibh_analyze_image3(const HostPartView& v, const int32_t* image_in_domain, int32_t n_image, Image3Host& out);
void ibh_analyze_blocks3(const HostPartView& v, std::vector<BlockDesc3>& blocks, std::vector<int32_t>& irr_cells,
                         int64_t* info, const int32_t* image_in_domain, int32_t n_image, int32_t* n_phase1,
                         std::vector<int32_t>& htab, std::vector<int32_t>& ftab, Sweep3Host* sw);

static inline int ibh_grid(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 65535 * 16) g = 65535 * 16;
    return (int)g;
}

// XCD-aware placement of workgroup `wg` of `nwg` (cdna_hip_programming.md T1): workgroups are dealt round-robin over the 8
// XCDs, each with its own L2; this gives every XCD one CONTIGUOUS run of the work list (cells and blocks are in depth-first
// order: a run is a compact patch of the mesh), so that the lines a workgroup gathers from its neighbours' cells are in the
// L2 that fetched them as own cells.  Bijective for any nwg.
#ifdef __HIPCC__
__device__ __forceinline__ int32_t ibh_xcd_chunk(int32_t wg, int32_t nwg) {
    const int32_t q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
// the workgroup's position along x for kernels that run over cells (or output rows) and gather from neighbouring cells
#ifdef IBH_NO_XCD_CELLS   // (A/B builds)
#define IBH_WG_X() ((int64_t)blockIdx.x)
#else
#define IBH_WG_X() ((int64_t)ibh_xcd_chunk((int32_t)blockIdx.x, (int32_t)gridDim.x))
#endif
#endif
