/*
 * libibhip -- C ABI of the MI355X (gfx950) residual hot path for
 * ImmersedBoundary.jl-style block-octree partitions.
 *
 * This header is the drop-in boundary.  The reference has no
 * FFI of its own: its plug-in hook is `conv_to_backend` / `to_backend`
 * (/root/reference/src/ImmersedBoundary.jl:846-855, src/arraybends.jl:14-77)
 * plus Julia multiple dispatch on the operators
 * (/root/reference/src/ImmersedBoundary.jl:873-1157).  Each entry point below
 * names the reference function it replaces; julia/IBHip.jl shows the `ccall`
 * binding a maintainer would add.
 *
 * Conventions
 *   - fields are Float32, column-major `(rows, nv)` with leading dimension
 *     `ld` (in elements): variable v of row i is `a[i + v*ld]`  (README.md:172);
 *   - `dim` is 1-based like the reference; `dim = 0` means "all dims" where
 *     the reference allows it (JST_sensor);
 *   - index arrays handed to the *_create functions are HOST pointers in the
 *     caller's base (`index_base` = 1 for Julia, 0 for C/Python); the library
 *     keeps its own device copies;
 *   - field pointers (`u`, `out`, ...) are DEVICE pointers (from ibh_malloc or
 *     any HIP allocation of the caller, e.g. a torch tensor's data_ptr);
 *   - every function returns 0 on success, a non-zero code otherwise and
 *     never throws; `ibh_last_error()` gives the message (thread-local);
 *   - kernels are launched on the stream given to `ibh_set_stream` (per host
 *     thread; default: the null stream) and are asynchronous.
 */
#ifndef IBHIP_H
#define IBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ibh_part ibh_part; /* device-resident Partition  (ImmersedBoundary.jl:383-392) */
typedef struct ibh_acc ibh_acc;   /* device-resident Accumulator (accumulator.jl:12-16)        */
typedef struct ibh_bc ibh_bc;     /* device-resident Boundary    (ImmersedBoundary.jl:406-414) */

/* `Fluid` (cfd.jl:14-53).  The fused Euler residual uses R and gamma only; the transport
 * properties feed ibh_cfd_viscous_fluxes / dynamic_viscosity / heat_conductivity:
 * Sutherland's law with the reference's exponent 2/3 (cfd.jl:71-77) and k(T) = sum k[i]*T^i (:84-90). */
typedef struct ibh_fluid {
    float R;
    float gamma;
    float mu_ref;
    float Tref;
    float S;
    int32_t nk;   /* number of heat-conductivity coefficients, <= 4 */
    float k[4];
} ibh_fluid;

/* ---- runtime ---------------------------------------------------------- */
int ibh_init(int device);             /* hipSetDevice + sanity check that the device is gfx950 */
const char* ibh_last_error(void);
int ibh_set_stream(void* hip_stream); /* stream for subsequent calls of this host thread */
int ibh_sync(void);                   /* hipStreamSynchronize on that stream */
int ibh_version(void);

/* ---- memory (for hosts without their own HIP allocator, e.g. julia/IBHip.jl) */
int ibh_malloc(void** dptr, size_t bytes);
int ibh_free(void* dptr);
int ibh_h2d(void* dst, const void* src, size_t bytes);
int ibh_d2h(void* dst, const void* src, size_t bytes);
int ibh_memset(void* dst, int value, size_t bytes);

/* ---- Partition: replaces `to_backend(part, conv)` (ImmersedBoundary.jl:848, :788) ----
 * Uploads once.  `spacing`/`centers` are `(nc, nd)` column-major (part.spacing,
 * part.centers).  For each dim d (0..nd-1): `nf[d]` faces with
 * `owners[d]`/`neighbors[d]` (part.face_owners_neighbors[dim]); left/right
 * incidence in CSR form (`*_off[d]` has nc+1 entries, `*_idx[d]` face ids in
 * the accumulator's stencil order) = part.face_accumulators[(dim,false/true)]
 * whose weights are 1/len (ImmersedBoundary.jl:501-506, :675-684).
 * `domain` = part.domain (global cell ids, sorted) and `block_size` let the
 * library recover the block structure for the block-structured fast path;
 * pass domain = NULL or block_size = 0 to disable it.
 */
int ibh_partition_create(ibh_part** out, int nd, int32_t nc,
                         const float* spacing, const float* centers,
                         const int32_t* nf,
                         const int32_t* const* owners, const int32_t* const* neighbors,
                         const int32_t* const* left_off, const int32_t* const* left_idx,
                         const int32_t* const* right_off, const int32_t* const* right_idx,
                         int32_t n_image, const int32_t* image_in_domain,
                         const int32_t* domain, int block_size, int index_base);
int ibh_partition_destroy(ibh_part* part);
/* Introspection of the block analysis: info[0]=full blocks, [1]=irregular cells,
 * [2..6] = number of block sides classified SAME, MIRROR, COARSE, FINE, GENERAL,
 * [7] = blocks whose whole sweep is independent of skirt cells (IBH_PHASE_INTERIOR),
 * [8] = blocks eligible for the single-kernel sweep, [9] = blocks whose gradients go through the workspace
 *       in a mixed launch (0 when every block is eligible), [10] = 1 when every image block is eligible (IBH_IMAGE_ONLY
 *       sweeps are then one launch per phase), [11] = image blocks, [12] / [13] = 2x2 block groups of the quad sweep and
 *       blocks outside them when every block is eligible, [14] / [15] = the same among the image blocks. */
int ibh_partition_info(const ibh_part* part, int64_t* info, int n);

/* ---- Elementwise kernels: what `Base.Broadcast` on device arrays lowers to in the reference-side binding
 * (julia/IBHip.jl) -- the arithmetic a user closure writes between the operators, e.g.
 * `@. (uL + uR) * Cf / 2 + abs(Cf) * (uL - uR) / 2`, `ud .-= ...` (test/advection.jl:67-83), `max.(a, b)` and
 * `maximum(...)` (:52-59).  Fields are column-major (n, nv), contiguous.  An operand of ibh_ew_binary is a field of
 * the result's shape (nv* == nv), a column vector broadcast over the columns (nv* == 1) or a scalar (pointer NULL,
 * value s*).  `out` may alias an operand.  ibh_ew_reduce leaves its scalar on the device. */
enum { IBH_EW_ADD = 0, IBH_EW_SUB = 1, IBH_EW_MUL = 2, IBH_EW_DIV = 3, IBH_EW_MAX = 4, IBH_EW_MIN = 5, IBH_EW_SUM = 6 };
enum { IBH_EW_ABS = 16, IBH_EW_NEG = 17, IBH_EW_SQRT = 18, IBH_EW_COPY = 19 };
int ibh_ew_binary(int op, int64_t n, int nv, const float* a, int nva, float sa, const float* b, int nvb, float sb,
                  float* out);
int ibh_ew_unary(int op, int64_t total, const float* a, float* out);
int ibh_ew_fill(int64_t total, float value, float* out);
int ibh_ew_reduce(int op, int64_t total, const float* a, float* out_device);
/* A whole broadcast expression in ONE launch -- what Julia's broadcast fusion makes of
 * `@. (uL + uR) * Cf / 2 + abs(Cf) * (uL - uR) / 2` (test/advection.jl:76-80): a postfix program over up to 8 arrays
 * (fields (n, nv) or column vectors, nv* = 1) and 8 scalars; instruction = opcode | operand << 8 with opcodes
 * IBH_EW_ADD..MIN, IBH_EW_ABS..SQRT, IBH_EW_PUSH_ARRAY (operand = array index), IBH_EW_PUSH_SCALAR; at most 48
 * instructions, stack depth 8.  Every node is evaluated in Float32 exactly as the one-node kernels do (no contraction),
 * so the result equals the node-by-node evaluation bit for bit.  `out` may alias an array. */
enum { IBH_EW_PUSH_ARRAY = 32, IBH_EW_PUSH_SCALAR = 33 };
int ibh_ew_eval(int64_t n, int nv, int nprog, const int32_t* prog, int narr, const float* const* arrays,
                const int32_t* arr_nv, int nscal, const float* scalars, float* out);

/* Measurement switches of the kernels (A/B runs inside one process, no effect on results beyond rounding):
 *   "quad_variant"  variant of the quad / 3-D sweeps (4: wave time stamps; 512, 518: round-2 3-D kernels; ...)
 *   "quad_parts" 1 / 2 only the quads / only the single blocks; "quad_singles_first" grid order; "quad_singles_iters"
 *   "rows" 1: the row sweep instead of the quad sweep; "rows_singles" -1 by size / 0 / 1 second launch / 2 inside the launch
 *   "pairs" 0: no pair tiles; "arith_ids" 0: halo ids from the table rows; "transport_blocks" 0: face-list transport kernel */
int ibh_set_tuning(const char* key, int value);
/* Wave timeline of the quad sweep (quad_variant 4): 8 x uint64 per wave {start, end (100 MHz ticks), HW_ID, is_quad, 4 phase stamps of a quad wave}. */
int ibh_debug_buffer(void* device_buffer);

/* Measurement probe (not on the product path): the launch of the 2-D quad sweep with the work stripped down.
 * mode 0 = dispatch only, 1 = + own-cell loads and the store (16 B per cell), 2 = + tables and halo gathers. */
int ibh_probe_sweep(ibh_part* part, const float* u, const float* C, int64_t ldc, float* ud, int mode);
/* Dispatch cost of an empty grid (measurement): nwg workgroups x threads, lds_bytes of LDS per workgroup. */
int ibh_probe_dispatch(int nwg, int threads, int lds_bytes);

/* Host-only view of the 2-D block analysis ibh_partition_create runs (block table, halo / end tables, the 2x2 block
 * groups of the quad sweep): same inputs, no device needed.  Test infrastructure for the library's host logic; not
 * part of the reference's surface.  ibh_host2d_get copies item `what` (of quad set `set`: 0 = all blocks, 1 = image
 * blocks) into dst (may be NULL to query *nbytes). */
typedef struct ibh_host2d ibh_host2d;
enum {
    IBH_H2D_BLOCKS = 0,    /* block descriptors, 120 bytes each (struct BlockDesc2 of csrc/ibh_common.h) */
    IBH_H2D_HTAB = 1,      /* int32 [nblk][64] halo cell ids */
    IBH_H2D_ETAB = 2,      /* int32 [nblk][16] side-end cell ids */
    IBH_H2D_FUSABLE = 3,   /* char  [nblk] eligible for the single-kernel sweep */
    IBH_H2D_QUAD_DESC = 4, /* {int32 base; uint32 cls; float rh[2]} per quad */
    IBH_H2D_QUAD_TAB = 5,  /* int32 [nq][160] */
    IBH_H2D_SINGLES = 6,   /* int32 block indices outside quads */
    IBH_H2D_COUNTS = 7,    /* int64 [8]: blocks, quads, interior quads, singles, interior singles, fuse_all, img_all_fz, nB1 */
    IBH_H2D_INFO = 8,      /* int64 [12] as ibh_partition_info */
    IBH_H2D_PAIR_DESC = 9, /* pair tiles (two blocks side by side, base and base + 64) in the quad format; set 0 only */
    IBH_H2D_PAIR_TAB = 10, /* int32 [npair][160] */
    IBH_H2D_SINGLES2 = 11, /* int32 block indices outside quads and pairs */
    IBH_H2D_QUAD_AUX = 12, /* int32 [nq][40] companion rows: 32 end ids + 8 origins of arithmetic halo ids (or -1) */
    IBH_H2D_PAIR_AUX = 13  /* int32 [npair][40] */
};
int ibh_analyze2_host(ibh_host2d** out, int32_t nc, const float* spacing, const int32_t* nf,
                      const int32_t* const* owners, const int32_t* const* neighbors,
                      const int32_t* const* left_off, const int32_t* const* left_idx,
                      const int32_t* const* right_off, const int32_t* const* right_idx,
                      int32_t n_image, const int32_t* image_in_domain, const int32_t* domain, int index_base);
int ibh_host2d_get(const ibh_host2d* h, int what, int set, void* dst, int64_t cap_bytes, int64_t* nbytes);
int ibh_host2d_destroy(ibh_host2d* h);

/* ---- grid operators on a Partition (ImmersedBoundary.jl:873-1157) ----------
 * `u`: (nc, nv) cell field; `uf`: (nf_dim, nv) face field.  Outputs are new
 * arrays supplied by the caller (the reference returns fresh arrays).        */
int ibh_at_owners(const ibh_part*, int dim, const float* u, int nv, int64_t ldu, float* out, int64_t ldo);    /* :879 */
int ibh_at_neighbors(const ibh_part*, int dim, const float* u, int nv, int64_t ldu, float* out, int64_t ldo); /* :889 */
int ibh_at_faces(const ibh_part*, int dim, const float* u, int nv, int64_t ldu, float* out, int64_t ldo);     /* :899 */
int ibh_green_gauss(const ibh_part*, int dim, const float* uf, int nv, int64_t ldf, float* out, int64_t ldo,
                    int unsigned_sum);                                                                         /* :918, :934 */
int ibh_cell_gradient(const ibh_part*, int dim, const float* u, int nv, int64_t ldu, float* out, int64_t ldo); /* :965 */
/* every dimension in one face-list launch: out[(d * nv + v) * ldo + c] (what ibh_cell_gradient_nd calls on partitions
 * without the block structure) */
int ibh_cell_gradient_all(const ibh_part*, const float* u, int nv, int64_t ldu, float* out, int64_t ldo);
/* cell_gradient(part, u): the tuple form, src/ImmersedBoundary.jl:980-988 -- all dimensions in one sweep per field:
 * out (nc, nd*nv), gradient of field v along dimension d in column d*nv + v; sensor (nc, nv) or NULL: JST_sensor(part, u)
 * (:1077-1097, dim = 0) of every field for free.  Block-structured partitions: pass A of the two-kernel sweeps (tuned
 * arithmetic, inside 5e-6 norm-wise of ibh_cell_gradient / ibh_jst_sensor); others: those kernels, one dimension at a time.
 * With nv = 1, ldo = lds = nc and sensor = out + nd * nc (one (nc, nd + 1) buffer) the sweep writes in place: no copy. */
int ibh_cell_gradient_nd(ibh_part*, const float* u, int nv, int64_t ldu, float* out, int64_t ldo, float* sensor,
                         int64_t lds);
/* The same FIELD by field, every sweep in place (no copies): out (nc, nv * (nd + 1)), leading dimension nc -- gradient of field
 * v along dimension d in column v * (nd + 1) + d, JST sensor of field v in column v * (nd + 1) + nd; the gradients of all
 * fields along d are the strided view of columns d, d + (nd + 1), ... (leading dimension (nd + 1) * nc). */
int ibh_cell_gradient_fields(ibh_part*, const float* u, int nv, int64_t ldu, float* out);
int ibh_face_distance(const ibh_part*, int dim, float* out);     /* :995  */
int ibh_owner_distance(const ibh_part*, int dim, float* out);    /* :1010 */
int ibh_neighbor_distance(const ibh_part*, int dim, float* out); /* :1024 */
int ibh_face_gradient(const ibh_part*, int dim, const float* u, int nv, int64_t ldu, float* out, int64_t ldo); /* :1039 */
int ibh_jst_sensor(const ibh_part*, int dim, const float* p, int nv, int64_t ldp, float* out, int64_t ldo);    /* :1077 */
/* MUSCL(part,u,du,dim;D,high_order) :1113.  `D` may be NULL (no sensor blending). */
int ibh_muscl(const ibh_part*, int dim, const float* u, const float* du, int nv, int64_t ld,
              const float* D, int high_order, float* uL, float* uR, int64_t ldf);

/* ---- Accumulator (accumulator.jl:39-130): CSR form of the bucketed stencils ----
 * Row r sums `w[k]*v[idx[k]]` for k in [off[r], off[r+1]) in that order; `w` may
 * be NULL (unweighted).  Used for face incidence, BC image interpolation and
 * multigrid coarseners/prolongators (ImmersedBoundary.jl:1391-1392).           */
int ibh_acc_create(ibh_acc** out, int32_t n_output, int32_t n_input,
                   const int32_t* off, const int32_t* idx, const float* w, int index_base);
int ibh_acc_destroy(ibh_acc*);
int ibh_accumulate(const ibh_acc*, const float* v, int nv, int64_t ldv, float* out, int64_t ldo);
/* out .+= acc(v .- v2) in one launch (the prolongation step of FAS!, solver.jl:76: Q .+= prolong(Qc .- Qcold)); same
 * arithmetic as the three separate operations.  For 2..8 fields on a many-rows-per-donor operator (a prolongation) the donors'
 * differences are first packed side by side into a scratch buffer the accumulator allocates on its first such call (so: not
 * inside a stream capture the first time, and not from two host threads at once on the same accumulator). */
int ibh_accumulate_diff_add(const ibh_acc*, const float* v, const float* v2, int nv, int64_t ldv, float* out, int64_t ldo);

/* ---- ghost-cell BC: impose_bc! (ImmersedBoundary.jl:1197-1247) ----------------
 * Boundary data: ghost ids (global rows of `a`), image-point interpolator
 * already re-indexed to `image_domain` (Boundary ctor :436-447), and
 * eta = ghost_distances ./ image_distances (:1220).                             */
int ibh_bc_create(ibh_bc** out, int32_t n_ghost, const int32_t* ghost_indices,
                  const float* ghost_distances, const float* image_distances,
                  int32_t n_image_domain, const int32_t* image_domain,
                  const int32_t* interp_off, const int32_t* interp_idx, const float* interp_w,
                  int index_base);
int ibh_bc_destroy(ibh_bc*);
/* ia = image_interpolator(a[image_domain, :])  (:1228-1230); ia is (n_ghost, nv) */
int ibh_bc_interp(const ibh_bc*, const float* a, int nv, int64_t lda, float* ia, int64_t ldi);
/* a[ghost, :] = eta*ia + (1-eta)*ba (:1242-1245).  `ba` (n_ghost, nv), or NULL
 * to use the per-variable constants `ba_const[nv]` (closures returning a scalar,
 * test/advection.jl:34).                                                        */
int ibh_bc_blend(const ibh_bc*, float* a, int nv, int64_t lda, const float* ia, int64_t ldi,
                 const float* ba, int64_t ldb, const float* ba_const);
/* Fused interp + blend for the two closures the reference's tests use:
 * mode 0: Dirichlet constant  (ba = ba_const),  mode 1: copy / zero-gradient (ba = ia). */
int ibh_bc_apply(const ibh_bc*, float* a, int nv, int64_t lda, int mode, const float* ba_const);

/* ---- partition runtime: (dom::Domain)(f, args...) (ImmersedBoundary.jl:836-859) ----
 * gather: local[i,:] = global[domain[i],:]; scatter: global[image[j],:] = local[image_in_domain[j],:].
 * `rows` is a DEVICE index array (0-based) of length n.                          */
int ibh_gather_rows(const int32_t* rows, int32_t n, const float* src, int nv, int64_t lds, float* dst, int64_t ldd);
int ibh_scatter_rows(const int32_t* rows, int32_t n, const float* src, int nv, int64_t lds, float* dst, int64_t ldd);
/* dst[dst_rows[i],:] = src[src_rows[i],:] (halo unpack / image scatter) */
int ibh_copy_rows(const int32_t* dst_rows, const int32_t* src_rows, int32_t n,
                  const float* src, int nv, int64_t lds, float* dst, int64_t ldd);

/* ---- fused residual sweeps (the headline hot path) --------------------------------
 * ibh_residual_advection: exactly the closure of test/advection.jl:67-83,
 *     D = JST_sensor(part,u); per dim: Cf = at_faces(C[:,dim]); gu = cell_gradient(u,dim);
 *     uL,uR = MUSCL(u,gu,dim; D, high_order=true);
 *     ud -= green_gauss((uL+uR)*Cf/2 + |Cf|*(uL-uR)/2, dim)
 *   with ud starting from 0.  `C` is (nc, nd) with leading dimension ldc.
 * ibh_residual_euler_hll: R2 of SURVEY.md 8d composed from reference operators,
 *     D = JST_sensor(part,P[:,1]); per dim: gP = cell_gradient(P,dim);
 *     PL,PR = MUSCL(P,gP,dim; D, high_order=true); F = inviscid_fluxes(fluid,PL,PR,dim)
 *     (cfd.jl:459-508); R -= green_gauss(F,dim), R starting from 0;  P = [p T u v (w)].
 * flags: bit0 = force the general face-list kernels (no block fast path);
 *        bit1 = image cells only (skirt rows of the output are left untouched);
 *        bit2 / bit3 = run only pass A / only pass B of the two-kernel sweep.
 */
#define IBH_FORCE_GENERAL 1
#define IBH_IMAGE_ONLY 2
#define IBH_PASS_A_ONLY 4 /* launch only the gradient+sensor kernel (profiling / overlap staging) */
#define IBH_PASS_B_ONLY 8 /* launch only the flux kernel; the workspace must hold a current pass A */
#define IBH_PHASE_INTERIOR 32 /* only blocks that do not depend on skirt cells (run while the halo exchange is in flight) */
#define IBH_PHASE_BOUNDARY 64 /* the complement: remaining blocks + face-list cells (run after the exchange) */
#define IBH_NO_FUSE 128       /* keep the two-kernel form (gradient workspace) even where one kernel could do the sweep */
#define IBH_FORCE_MIXED 512    /* take the mixed launch (single kernel + two-kernel form) whatever the partition size */
#define IBH_SWEEP_ONLY 256     /* measurement: of a mixed launch run only the single-kernel part */
#define IBH_NO_QUAD 1024      /* A/B: per-block single kernel where the quad sweep (2x2 block groups per wavefront) would run */
#define IBH_EXACT 16      /* block fast path with the literal IEEE arithmetic (bit-comparable with the face-list path) */
int ibh_residual_advection(ibh_part*, const float* u, const float* C, int64_t ldc, float* ud, int flags);
/* n such sweeps launched back to back by one call (the step loop of a compiled host) */
int ibh_residual_advection_n(ibh_part*, const float* u, const float* C, int64_t ldc, float* ud, int flags, int n);
int ibh_residual_euler_hll(ibh_part*, const float* P, int64_t ldp, float* R, int64_t ldr,
                           const ibh_fluid* fluid, int flags);

/* ---- CFD pointwise physics that residual closures call (cfd.jl), rows of P are [p T u v (w)],
 * rows of Q are [rho E rho*u rho*v (rho*w)]; all arrays (n, nd+2) column-major, `dim` 1-based. ---- */
int ibh_cfd_speed_of_sound(const ibh_fluid*, int64_t n, const float* T, float* a);          /* cfd.jl:62-64  */
int ibh_cfd_dynamic_viscosity(const ibh_fluid*, int64_t n, const float* T, float* mu);      /* cfd.jl:71-77  */
int ibh_cfd_heat_conductivity(const ibh_fluid*, int64_t n, const float* T, float* k);       /* cfd.jl:84-90  */
int ibh_cfd_primitive2state(const ibh_fluid*, int nd, int64_t n, const float* P, int64_t ldp,
                            float* Q, int64_t ldq);                                          /* cfd.jl:106-123 */
int ibh_cfd_state2primitive(const ibh_fluid*, int nd, int64_t n, const float* Q, int64_t ldq,
                            float* P, int64_t ldp);                                          /* cfd.jl:137-151 */
/* HLL flux (cfd.jl:459-508).  The reference evaluates the last combine in Float64 (its `0.0`
 * literals) and returns Float64; here the combine is Float64 too and the result is rounded to Float32. */
int ibh_cfd_inviscid_fluxes_hll(const ibh_fluid*, int nd, int dim, int64_t n, const float* PL, const float* PR,
                                int64_t ld, float* F, int64_t ldf);
/* central flux + Rusanov dissipation scaled by sensors nuL/nuR (n) (cfd.jl:516-554) */
int ibh_cfd_inviscid_fluxes_sensor(const ibh_fluid*, int nd, int dim, int64_t n, const float* PL, const float* PR,
                                   int64_t ld, const float* nuL, const float* nuR, float* F, int64_t ldf);
/* viscous flux along Cartesian `dim` (cfd.jl:664-736): `Pgrad` = HOST array of nd DEVICE pointers, the
 * gradient of P along each axis, each (n, nd+2) with leading dimension ldg; mu_t per row or NULL (+ constant). */
/* CFD.JST_sensor(Pim1, Pi, Pip1) (cfd.jl:563-573), elementwise over `n` values, and CFD.shock_sensor (cfd.jl:575-617):
 * velocity_gradients[i * nd + j] = device array of d u_i / d x_j (n values each). */
int ibh_cfd_jst_sensor3(int64_t n, const float* Pim1, const float* Pi, const float* Pip1, float* out);
int ibh_cfd_shock_sensor(int nd, int64_t n, const float* const* velocity_gradients, float* out);
int ibh_cfd_viscous_fluxes(const ibh_fluid*, int nd, int dim, int64_t n, const float* P, int64_t ldp,
                           const float* const* Pgrad, int64_t ldg, const float* mu_t, float mu_t_const,
                           float* F, int64_t ldf);
/* The viscous part of a Navier-Stokes residual closure in ONE launch (round 4):
 *   R[:, 2:end] .+= sum_d green_gauss(part, viscous_fluxes(fluid, at_faces(part, P, d), face_gradient(part, P, gradP, d), d;
 *                                                          mu_t = at_faces(part, mu_t, d)), d)
 * (cfd.jl:664-736 over ImmersedBoundary.jl:899-926, 1039-1069; R[:, 1] gets the zero mass flux: untouched), operation by
 * operation in the composition's order: bit-identical to the twenty-odd operator launches per dimension it replaces.
 * Pgrad[d] = cell_gradient(part, P, d + 1), (nc, nd + 2) column-major with leading dimension ldg (grad_vel_col = 2), or the
 * gradients of the velocity columns only, (nc, nd) (grad_vel_col = 0: the gradients of p and T are not read -- the normal
 * derivative of T at a face is a face_gradient); mu_t: nc values (cells). */
int ibh_viscous_residual(const ibh_part*, const ibh_fluid*, const float* P, int64_t ldp, const float* const* Pgrad,
                         int64_t ldg, int grad_vel_col, const float* mu_t, float* R, int64_t ldr);

/* ---- direct peer halo exchange over xGMI ("or direct peer ... IPC writes") -------
 * Building blocks; the orchestration (who writes where) is host-side (halo.py / the Julia shim):
 *   - receive buffers and flag words are allocated fine-grained (coherent across devices), exported as
 *     64-byte HIP IPC handles, and mapped by the peers;
 *   - per sweep the sender packs with ibh_gather_rows straight INTO the peer's mapped receive buffer, then
 *     ibh_flag_signal bumps a sequence number in the peer's flag word (system-scope release);
 *   - the receiver runs ibh_flag_wait (bounded spin, never hangs: on timeout bit 0 of *status is set and
 *     the host falls back to the RCCL path) and unpacks with ibh_scatter_rows.
 * Everything is an ordinary kernel on the caller's stream, so whole sweeps incl. the exchange can be
 * captured in a HIP graph. */
int ibh_ipc_alloc(void** dptr, size_t bytes, int fine_grained);
int ibh_ipc_free(void* dptr);
int ibh_ipc_export(void* dptr, void* handle64);        /* hipIpcGetMemHandle: writes 64 bytes            */
int ibh_ipc_import(const void* handle64, void** dptr); /* hipIpcOpenMemHandle (lazy peer access)         */
int ibh_ipc_close(void* dptr);
/* seq = ++(*counter); *slots[q] = seq for q < n.  `counter` local device memory, `slots` DEVICE array of n
 * device pointers (into the peers' flag arrays). */
int ibh_flag_signal(uint32_t* counter, uint32_t* const* slots, int n);
/* exp = ++(*counter); wait until *slots[q] >= exp for every q < n, at most max_spins polls each. */
int ibh_flag_wait(uint32_t* counter, const uint32_t* const* slots, int n, uint32_t max_spins, uint32_t* status);
/* The whole exchange in two launches.  state = 5 device words {signal seq, wait seq, status, 0, 0}.
 * push: for every peer q the rows send_all[seg[q]..seg[q+1]) of f go to dst[q] as (n_q, nv) column-major, then the
 *   last workgroup to finish stores ++signal seq into flags[q] of every peer (system-scope release).
 * pull: waits (bounded by max_spins polls, time-out sets status bit 0) until every flags[q] >= wait seq + 1, then
 *   scatters src (the peers' blocks back to back, block q = (n_q, nv) column-major) to the rows recv_all[..] of f.
 * seg / dst / flags are host arrays, copied into the launch.  At most 16 peers. */
int ibh_halo_push(const float* f, int nv, int64_t ld, const int32_t* send_all, int n_peers, const int32_t* seg,
                  float* const* dst, uint32_t* const* flags, uint32_t* state);
int ibh_halo_pull(float* f, int nv, int64_t ld, const int32_t* recv_all, const float* src, int n_peers,
                  const int32_t* seg, const uint32_t* const* flags, uint32_t* state, uint32_t max_spins);
/* push + pull in ONE launch (the same two steps; ranks running it concurrently cannot block each other: the push part
 * waits for nothing).  Sends rows of f, receives into other rows of f.  dst0/dst1, src0/src1: the two parities of
 * the double buffer; which one a launch uses follows the device-side sequence number (state[0] & 1), so graph
 * replays and eager launches can be mixed. */
int ibh_halo_exchange(float* f, int nv, int64_t ld, const int32_t* send_all, int n_send_peers, const int32_t* send_seg,
                      float* const* dst0, float* const* dst1, uint32_t* const* send_flags, const int32_t* recv_all,
                      const float* src0, const float* src1, int n_recv_peers, const int32_t* recv_seg,
                      const uint32_t* const* recv_flags, uint32_t* state, uint32_t max_spins);
/* One step of a rank in ONE launch: ibh_halo_exchange of the scalar field u + the image-only quad sweep of
 * ibh_residual_advection(u, C) -> ud (test/advection.jl:67-83 on the rank's image cells).  The exchange workgroups run
 * beside the interior quads; a boundary wave waits (bounded by max_spins, time-out sets status bit 1) until the skirt rows
 * of ITS launch are unpacked: the overlap of exchange and interior compute that north_star asks for, without a second
 * stream.  fstate: 2 device uint64, zeroed once per exchanger.  Fails (no launch) on partitions whose image blocks are
 * not all eligible for the quad sweep; the caller then runs the two entry points one after the other. */
int ibh_step_advection_xgmi(ibh_part*, float* u, const float* C, int64_t ldc, float* ud, const int32_t* send_all,
                            int n_send_peers, const int32_t* send_seg, float* const* dst0, float* const* dst1,
                            uint32_t* const* send_flags, const int32_t* recv_all, const float* src0, const float* src1,
                            int n_recv_peers, const int32_t* recv_seg, const uint32_t* const* recv_flags,
                            uint32_t* state, uint32_t max_spins, unsigned long long* fstate);

/* ---- small device-resident vector ops for the FAS loop (solver.jl:79-88) ---------- */
/* q += clamp(omega,0,1) * r ; omega scalar */
/* ---- an explicit solver step, device resident (test/advection.jl:30-89: march! = dt, closure, u .+= ud .* dt, apply_bcs!)
 * ibh_bcset: an ordered list of impose_bc! calls (ImmersedBoundary.jl:1197-1247) whose closures the library knows --
 *   mode 0: `do bdry, u; value end`, mode 1: `do bdry, u; copy(u) end` (advection.jl:30-46) -- applied with the semantics
 *   of the sequential calls: boundaries that do not read each other's ghost cells form a level; a level is two launches
 *   (interpolate + closure + blend into a side buffer, then scatter) whatever the number of boundaries.
 * ibh_timestep_advection: dt = scale * 0.5 / maximum(max.(unsigned_green_gauss(at_faces(C_d, d), d)...)) (:52-59, :65)
 *   written to device memory; ibh_step_advection: u_out = u + dt * R(u) in ONE launch where the quad sweep applies (else
 *   sweep + update), then the boundary conditions on u_out.  u and u_out must not alias. */
typedef struct ibh_bcset ibh_bcset;
int ibh_bcset_create(ibh_bcset** out, int n_bc, const ibh_bc* const* bcs, const int32_t* modes, const float* values);
int ibh_bcset_destroy(ibh_bcset*);
/* n_levels: low 16 bits = levels, high 16 bits = levels none of whose ghost cells is a donor of the level (one launch) */
int ibh_bcset_info(const ibh_bcset*, int32_t* n_ghost, int32_t* n_levels);
int ibh_bcset_apply(const ibh_bcset*, float* a);
int ibh_timestep_advection(ibh_part*, const float* C, int64_t ldc, float scale, float* dt_device);
int ibh_update_dev(int64_t n, const float* dt_device, const float* u, const float* r, float* out);
int ibh_step_advection(ibh_part*, const float* u, float* u_out, const float* C, int64_t ldc, const float* dt_device,
                       const ibh_bcset* bcs /* or NULL */);
/* The same step with the time step of the NEXT one evaluated on the way: dt_next = ibh_timestep_advection(p, C, ldc, scale)
 * -- it depends on C alone -- computed by extra workgroups of the BC set's own launches (partial maxima beside the first,
 * final reduction beside the second) instead of two launches in front of the next sweep.  Results identical to the separate
 * calls; dt_next may alias dt_dev. */
int ibh_step_advection_dt(ibh_part* p, const float* u, float* u_out, const float* C, int64_t ldc, const float* dt_dev,
                          const ibh_bcset* bcs, float scale, float* dt_next);
int ibh_axpy_clamped(int64_t n, float omega, const float* r, float* q);
/* y = a*x + y */
int ibh_axpy(int64_t n, float a, const float* x, float* y);
/* *out (device, double) = sum(x^2) */
int ibh_sumsq(int64_t n, const float* x, double* out);
/* q += clamp(omega,0,1) * r and *out = sum(r^2) in one pass (solver.jl:82 + the norm of :84 on the same array) */
int ibh_axpy_clamped_sumsq(int64_t n, float omega, const float* r, float* q, double* out);
/* One pass of FAS! over a residual array (solver.jl:80-84): rr = r [+ source]; [q += clamp(omega,0,1) * rr]; [*out_sumsq =
 * sum(rr^2)] -- `r .+= source`, the update and the norm on one read of r; source, q and out_sumsq may each be NULL. */
int ibh_fas_update(int64_t n, float omega, const float* r, const float* source, float* q, double* out_sumsq);

/* FlowBC call (cfd.jl:243-300): boundary state [p T u v (w)] from the image-point primitives P and the unit normals.
 * u_inf: nd components, or ONE component (the normal velocity) when normal_flow != 0.  image_distances / dudn: both
 * null or both given (wall-function slip scaling :287-292); transpiration: scalar, or per-row array when
 * transpiration_v != null.  Host array u_inf; everything else device. */
int ibh_cfd_flow_bc(const ibh_fluid* f, int nd, int64_t n, const float* P, int64_t ldp, const float* normals, int64_t ldn,
                    float p_inf, float T_inf, const float* u_inf, int normal_flow, const float* image_distances,
                    const float* dudn, float transpiration, const float* transpiration_v, float* out, int64_t ldo);

/* ---- turbulence closures (src/turbulence.jl), pointwise, Float32 ------------------------------------------------
 * params8 (host) = {kappa, C, A, beta, betastar, D, Aplus, omega_fixed_point}; velocity gradients g: host table of
 * nd*nd device pointers, g[i*nd + j] = d u_i / d x_j. */
/* wall_function(Rey) :27-70 -> y+, u+, mu+, k+, du+/dy+ */
int ibh_turb_wall_function_rey(int64_t n, const float* Rey, const float* params8, int n_iter, float* yplus, float* uplus,
                               float* muplus, float* kplus, float* dudy);
/* wall_function(y, u, nu) :72-100 -> u_tau, nu_t, k, omega, epsilon, du/dn */
int ibh_turb_wall_function(int64_t n, const float* y, const float* u, const float* nu, const float* params8, int n_iter,
                           float* utau, float* nut, float* k, float* omega, float* eps, float* dudn);
/* shear_rate :110-124 = sqrt(2 Sij Sij) */
int ibh_turb_shear_rate(int nd, int64_t n, const float* const* g, float* S);
/* Smagorinsky_nuSGS :135-138 = (Cs Delta)^2 S */
int ibh_turb_smagorinsky(int64_t n, const float* Delta, const float* S, float Cs, float* out);
/* standard_k-epsilon :176-196; params5 (host) = {Cmu, sigma_k, sigma_eps, C1eps, C2eps} */
int ibh_turb_k_epsilon(int64_t n, const float* k, const float* eps, const float* S, const float* params5, float* nuk,
                       float* nue, float* Sk, float* Se, float* nut);
/* Wray_Agarwal :222-241; gradR, gradS (n, nd) column-major */
int ibh_turb_wray_agarwal(int nd, int64_t n, const float* R, const float* S, const float* gradR, int64_t ldr,
                          const float* gradS, int64_t lds, float sigmaR, float C1, float kappa, float* nut, float* nuR,
                          float* Sout);
/* Ducros_sensor :252-282; WALE_nuSGS :291-337 (3-D only) */
/* transport of a scalar with variable diffusivity, all dimensions in one launch:
 *   out = S + sum_d green_gauss(at_faces(nu + nuR, d) .* face_gradient(R, d) .- at_faces(vel[:, d] .* R, d), d)
 * (the right-hand side a one-equation turbulence model such as Wray_Agarwal, turbulence.jl:222-241, is closed with):
 * the operator composition of ImmersedBoundary.jl:899-943, 1039-1043 evaluated face by face, bit-identical to calling the
 * operators one after the other. */
int ibh_scalar_transport(const ibh_part*, const float* R, const float* nuR, float nu, const float* vel, int64_t ldv,
                         const float* S, float* out);
/* On a 3-D partition made of complete 8^3 blocks (ibh_partition_info: full_blocks * 512 == nc, no face-list cells, no
 * GENERAL sides) or on a partition without block structure (full_blocks == 0: the coarse levels of multigrid()) -- an
 * error otherwise: compose the operators then -- the gradients consumed where they are made:
 *   S = shear_rate(cell_gradient(part, u), cell_gradient(part, v), cell_gradient(part, w))        turbulence.jl:110-124
 *   (nut, nuR, Sout) = Wray_Agarwal(R, S, cell_gradient(part, R), cell_gradient(part, S))          turbulence.jl:222-241
 * with the arithmetic of the tuple cell_gradient's block sweep (ibh_cell_gradient_nd) and of the pointwise kernels. */
int ibh_shear_rate_of_velocity(ibh_part*, const float* vel, int64_t ldv, float* S);
/* The same with the velocity gradients kept: G (nc, nd * nd), leading dimension ldg, d u_i / d x_j in column nd * j + i -- the
 * layout of the tuple cell_gradient (ibh_cell_gradient_nd), so that G + nd * j * ldg is `cell_gradient(part, vel)[j]` for
 * ibh_viscous_residual (grad_vel_col = 0): a Navier-Stokes closure with a turbulence model needs both, and the gradients
 * are made once.  G may be NULL (= ibh_shear_rate_of_velocity). */
int ibh_shear_rate_of_velocity_grad(ibh_part*, const float* vel, int64_t ldv, float* S, float* G, int64_t ldg);
int ibh_wray_agarwal_of(ibh_part*, const float* R, const float* S, float sigmaR, float C1, float kappa, float* nut,
                        float* nuR, float* Sout);
int ibh_turb_ducros(int nd, int64_t n, const float* const* g, float* out);
int ibh_turb_wale(int64_t n, const float* Delta, const float* const* g, float Cw, float* out);

/* ---- point-implicit smoother (reference: the orphan file src/point_implicit.jl) ------------------------------
 * Arrays are dense column-major (n points, nv variables) with leading dimension n, i.e. n*nv contiguous floats;
 * the block diagonal D is (n, nv, nv) column-major: D[p + n*(k + nv*i)] = d f_k / d x_i at point p (:56-91). */
/* z[i] = +1 / -1 from a counter-based hash of (seed, i)  (the Rademacher samples of :36-38) */
int ibh_pi_rademacher(int64_t n, uint64_t seed, float* z);
/* out = x + v*h  (:30, :112) */
int ibh_pi_perturb(int64_t n, const float* x, const float* v, float h, float* out);
/* out = (fxb - fx) / h : the Jacobian-vector product of a Linearization (:110-114) */
int ibh_pi_fd(int64_t n, const float* fxb, const float* fx, float h, float* out);
/* s[p,k] += z[p] * (fxb[p,k] - fx[p,k]) / h  (:40) */
int ibh_pi_hutch_accum(int64_t n, int nv, const float* fxb, const float* fx, const float* z, float h, float* s);
/* s /= d  (:43) */
int ibh_pi_div_scalar(int64_t n, float d, float* s);
/* in place: nv == 1: D = 1/(eps + D) (:124-126); else per-point Moore-Penrose inverse of the nv x nv block
 * (:127-135; one-sided Jacobi SVD, tolerance eps*nv*sigma_max like LinearAlgebra.pinv).  1 <= nv <= 8. */
int ibh_pi_invert_blocks(int64_t n, int nv, float* D);
/* out[p,k] = sum_i v[p,i] * invD[p,k,i]  (:153-161; nv == 1: out = v .* invD :141-146) */
int ibh_pi_apply_blocks(int64_t n, int nv, const float* invD, const float* v, float* out);
/* *out (device, double) = sum(a .* b);  *out (device, float) = maximum(abs, a) */
int ibh_dot(int64_t n, const float* a, const float* b, double* out);
int ibh_maxabs(int64_t n, const float* a, float* out);
/* alpha = dots[0] / (dots[1] + eps) read on the device; x += s*alpha; r -= As*alpha  (:229-236, :291-294) */
int ibh_pi_update(int64_t n, const double* dots, float eps, const float* s, const float* As, float* x, float* r);
/* s = r / (eps + *maxabs)  (:297-299) */
int ibh_pi_normalize(int64_t n, const float* r, const float* maxabs, float eps, float* s);

#ifdef __cplusplus
}
#endif
#endif /* IBHIP_H */
