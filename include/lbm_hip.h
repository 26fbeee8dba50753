/* lbm_hip.h -- C ABI of the MI355X-native D2Q9 lattice-Boltzmann engine (liblbm_hip.so).
 *
 * Drop-in boundary for ONE hot path of cristian-jfv/lattice-boltzmann-method: the
 * collide-then-stream time step of src/solver.cpp and the variants built on it.
 * The reference has no FFI of its own (its "API" is the C++ headers in src/ consumed by
 * the drivers in test/); every entry point below cites the reference interface it
 * replaces.  Plain pointers and sizes only -- no torch / HIP types in the signatures
 * (lbm_stream_t is a hipStream_t passed as void*; NULL = the default stream).
 *
 * DATA LAYOUT.  The reference holds f as contiguous row-major [R][C][Q] f64 ("AoS", q
 * innermost; src/domain.cpp:7-11).  The engine works on Structure-of-Arrays planes
 *     f[q][r][c]   (q slowest, c fastest, f64)
 * so that a wavefront reads 64 consecutive doubles of one population.  lbm_aos_to_soa /
 * lbm_soa_to_aos convert bit-exactly: AoS element ((r*C)+c)*Q+q  <->  SoA q*R*C + r*C + c.
 * Scalar fields: rho[R][C]; vector fields u[2][R][C] (component slowest).
 * A lattice with g ghost rows per side (multi-GPU slabs; g = the steps per launch D, or m x D with one
 * exchange per m launches, or 3 for the two-phase step) stores planes of (R+2g) rows: row index r in
 * [-g, R+g) lives at plane offset (r+g)*C; see lbm_geom.
 *
 * All functions return 0 (LBM_OK) or a negative status and never throw; the message of
 * the last failure on the calling thread is lbm_last_error_string().  Calls enqueue work
 * on the given stream and return; nothing allocates inside a step call (graph-capturable).
 */
#ifndef LBM_HIP_H
#define LBM_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LBM_OK 0
#define LBM_ERR_INVALID (-1) /* bad argument (shape, NULL pointer, unknown mode) */
#define LBM_ERR_HIP (-2)     /* a HIP runtime call or kernel launch failed */
#define LBM_ERR_STATE (-3)   /* call not valid in the solver's current state */

typedef void* lbm_stream_t;

const char* lbm_last_error_string(void);
int lbm_abi_version(void);
/* number of visible HIP devices (0 when none); never fails */
int lbm_device_count(void);
int lbm_set_device(int dev);

/* ---- device memory / streams / events (so a C or C++ host needs no HIP headers) ---- */
int lbm_malloc(void** dptr, size_t bytes);
int lbm_free(void* dptr);
int lbm_memset(void* dptr, int value, size_t bytes, lbm_stream_t s);
int lbm_memcpy_h2d(void* dst, const void* src, size_t bytes, lbm_stream_t s);
int lbm_memcpy_d2h(void* dst, const void* src, size_t bytes, lbm_stream_t s);
int lbm_memcpy_d2d(void* dst, const void* src, size_t bytes, lbm_stream_t s);
int lbm_stream_create(lbm_stream_t* s);
int lbm_stream_destroy(lbm_stream_t s);
int lbm_stream_sync(lbm_stream_t s);
int lbm_event_create(void** ev);
int lbm_event_destroy(void* ev);
int lbm_event_record(void* ev, lbm_stream_t s);
int lbm_stream_wait_event(lbm_stream_t s, void* ev); /* work enqueued on s afterwards waits for ev (hipStreamWaitEvent) */
int lbm_event_elapsed_ms(float* ms, void* start, void* stop); /* synchronises on stop */

/* ---- HIP graphs: capture a launch sequence of this library once, replay it per step -----------------
 * Step calls only enqueue kernels, so a launch-bound loop (many small launches per step: small
 * lattices, multi-block topologies) can be captured from a created (non-default) stream and
 * replayed with one submission per step.  Pointers and parameters are frozen at capture time. */
typedef struct lbm_graph lbm_graph;
int lbm_graph_begin_capture(lbm_stream_t s);
int lbm_graph_end_capture(lbm_stream_t s, lbm_graph** out);
int lbm_graph_launch(lbm_graph* g, int times, lbm_stream_t s);
int lbm_graph_destroy(lbm_graph* g);

/* ---- layout converters (device buffers).  Q = 9 (f), 1 (rho), 2 (u) ---------------- */
int lbm_aos_to_soa(double* soa, const double* aos, int R, int C, int Q, lbm_stream_t s);
int lbm_soa_to_aos(double* aos, const double* soa, int R, int C, int Q, lbm_stream_t s);
/* same with an explicit SoA plane stride in doubles (0 = dense R*C), see lbm_geom */
int lbm_aos_to_soa_ex(double* soa, const double* aos, int R, int C, int Q, long long plane_stride,
                      lbm_stream_t s);
int lbm_soa_to_aos_ex(double* aos, const double* soa, int R, int C, int Q, long long plane_stride,
                      lbm_stream_t s);
/* the same with a row pitch on the SoA side (lbm_geom.row_pitch; 0 = dense): node (r, c) of plane q at
 * q * plane_stride + r * row_pitch + c */
int lbm_aos_to_soa_pitched(double* soa, const double* aos, int R, int C, int Q,
                           long long plane_stride, int row_pitch, lbm_stream_t s);
int lbm_soa_to_aos_pitched(double* aos, const double* soa, int R, int C, int Q,
                           long long plane_stride, int row_pitch, lbm_stream_t s);
/* padding (doubles) the engine recommends between population planes of an R x C lattice */
long long lbm_default_plane_pad(int R, int C);

/* ---- unfused parity operators: one per solver:: function (src/solver.hpp:11-36) -----
 * SoA planes without ghost rows.  Unlike the reference outputs are written in
 * place, never re-bound. */
int lbm_calc_rho(double* rho, const double* f, int R, int C, lbm_stream_t s);                 /* solver.cpp:23-26 */
int lbm_calc_u(double* u, const double* f, const double* rho, int R, int C, lbm_stream_t s);  /* solver.cpp:34-37 */
int lbm_calc_incomp_u(double* u, const double* f, int R, int C, lbm_stream_t s);              /* solver.cpp:28-31 */
int lbm_equilibrium(double* feq, const double* u, const double* rho, int R, int C, lbm_stream_t s);        /* :51-62 */
int lbm_incomp_equilibrium(double* feq, const double* u, const double* rho, int R, int C, lbm_stream_t s); /* :39-49 */
int lbm_collision(double* f_coll, const double* f_curr, const double* f_equi, double omega,
                  int R, int C, lbm_stream_t s);                                              /* solver.cpp:65-74 */
int lbm_advect(double* g, const double* f, int R, int C, lbm_stream_t s);                     /* solver.cpp:76-131 */

/* ---- geometry and boundary descriptors ------------------------------------------------ */
typedef struct lbm_geom {
  int R;     /* rows owned by this block / slab (reference dim 0, pairs with c_x) */
  int C;     /* columns (reference dim 1, pairs with c_y) */
  int ghost; /* 0: planes are [R][C] and streaming wraps rows periodically inside the block
                g >= 1: planes are [R+2g][C]; rows -g..-1 and R..R+g-1 are ghost rows owned by the
                neighbouring slabs (1: one step per launch; n: n-step launches; 3: the
                colour-gradient step) */
  long long plane_stride; /* doubles between consecutive population planes; 0 = dense
                             ((R + 2*ghost) * row pitch).  Padding it off a power of two spreads the 18
                             concurrent streams of the fused step over the HBM channels. */
  int row_pitch;          /* doubles between consecutive rows of a plane; 0 = dense (C).  Node (r, c) of plane q
                             lives at q * plane_stride + (r + ghost) * row_pitch + c ("row padding, hidden behind the
                             ABI"): with C a power of two, rows a power of two apart land on the same
                             L2 sets and DRAM pages -- the solver contexts pad their own lattices (lbm_default_row_pitch).
                             Must be even and >= C.  Entry points that take a lattice honour it; buffers of
                             macroscopic fields (rho, u, psi ...) and AoS arrays are always dense. */
} lbm_geom;
/* rows [src_row, src_row + n_rows) of all 9 planes of `src` (geometry sg) into rows [dst_row, ...) of `dst` (geometry dg):
 * a node-addressed copy (row indices in owned-row numbering, ghost rows allowed), so that padded rows / planes and ghost
 * rows on either side do not matter; sg->C == dg->C.  For hosts that hold lattices of their own in a padded layout. */
int lbm_lattice_copy_rows(double* dst, const lbm_geom* dg, int dst_row, const double* src, const lbm_geom* sg,
                          int src_row, int n_rows, lbm_stream_t s);
/* the row pitch the solver contexts use for a lattice of C columns: C + 64 doubles where C * 8 bytes is a multiple
 * of 4 KiB and C >= 1024 (tuning "row_pad": that many doubles instead of 64, 0 = never), else C */
int lbm_default_row_pitch(int C);

/* What the reference drivers do to the populations a node cannot receive from inside the
 * domain.  Every mode restates one driver fix-up (all applied post-streaming, reading the
 * post-collision populations of the SAME node). */
enum lbm_edge_mode {
  LBM_EDGE_PERIODIC = 0,     /* nothing: solver::advect's own wrap (solver.cpp:84-128) */
  LBM_EDGE_HALO = 1,         /* rows only: ghost row is filled by the neighbouring slab */
  LBM_EDGE_BOUNCE_BACK = 2,  /* halfway bounce-back: horizontal_poiseuille_test.cpp:146-152 (columns),
                                mrtcg_rayleigh_taylor.cpp:525-531 (rows) */
  LBM_EDGE_SPECULAR = 3,     /* columns: cylinder_test.cpp:157-163 */
  LBM_EDGE_ABB_VELOCITY = 4, /* rows: anti-bounce-back with wall velocity, cylinder_test.cpp:135-154 */
  LBM_EDGE_WRAP_NOSHIFT = 5  /* columns, rows 1..R-2 only: the colour-gradient driver's same-row column
                                copy, mrtcg_rayleigh_taylor.cpp:517-523 */
};

typedef struct lbm_bc {
  int row_lo, row_hi; /* mode at global row 0 / row R-1 (enum lbm_edge_mode) */
  int col_lo, col_hi; /* mode at column 0 / column C-1 */
  /* pressure-periodic virtual rows, horizontal_poiseuille_test.cpp:25-45 (applied to the
     post-collision rows 0 and R-1 before streaming): 0 = off, 1 = on, using the model's own
     equilibrium (incompressible in that driver, compressible in decompose_domain.cpp:25-48) */
  int pressure_rows;
  double rho_inlet, rho_outlet;
  double uw_r, uw_c; /* wall velocity of LBM_EDGE_ABB_VELOCITY rows (u_w, cylinder_test.cpp:73) */
} lbm_bc;

/* ---- slab halo (multi-GPU): pack the rows a neighbour needs into ONE contiguous message -------
 * depth = ghost rows in use (1: one step per launch, n: n-step launches), or LBM_HALO_TWO_PHASE for
 * the colour-gradient step (3 ghost rows; the second row travels complete because of the driver's
 * same-row column copy, mrtcg_rayleigh_taylor.cpp:517-523).
 * Message = lbm_halo_rows(depth) rows of C doubles: 3 for depth 1, else 9 (depth - 1); 21 two-phase
 * (the block binding of test/decompose_domain.cpp:181-187 generalised).
 * pack  side 1: my last `depth` rows  -> message for the NEXT slab;  side 0: my first rows -> PREVIOUS.
 * unpack side 0: message from the PREVIOUS slab -> ghost rows above row 0;  side 1: from NEXT -> below. */
#define LBM_HALO_TWO_PHASE (-3)
/* all 9 populations of every one of d ghost rows (9 d rows): multi-step launches on slabs whose
 * columns are walls -- the fix-ups of a ghost-row wall node read that node's own populations */
#define LBM_HALO_FULL(d) (100 + (d))
int lbm_halo_rows(int depth);
int lbm_halo_pack(double* buf, const double* lattice, const lbm_geom* g, int depth, int side,
                  lbm_stream_t s);
int lbm_halo_unpack(double* lattice, const double* buf, const lbm_geom* g, int depth, int side,
                    lbm_stream_t s);

/* ---- which implementation of a collision runs (field `form` of the parameter structs below) ----------
 * Every model exists twice: in the reference's OPERATION ORDER (-ffp-contract=off, same expression order:
 * bitwise equal to the CPU oracle) and REASSOCIATED (same mathematics, fewer operations, reciprocals, FMA
 * decided per source expression: agreement to rounding, tolerances in tests/).  The choice is part of the
 * parameters of a solver / a call, so two solvers in one process may differ; LBM_FORM_DEFAULT (0, what a
 * zero-initialised struct holds) = the process-wide default of lbm_set_tuning ("bgk_fast", "kbc_fast",
 * "cg_fused": reassociated unless set to 0).  Lattices with pressure rows, a body force or the
 * incompressible equilibrium always run the reference order. */
#define LBM_FORM_DEFAULT 0
#define LBM_FORM_REFERENCE_ORDER 1
#define LBM_FORM_REASSOCIATED 2

/* ---- BGK (solver.cpp:23-74 fused with :76-131) ---------------------------------------- */
typedef struct lbm_bgk_params {
  double omega;
  int incompressible; /* 0: calc_u + equilibrium; 1: calc_incomp_u + incomp_equilibrium */
  int delta_form;     /* how the driver writes the relaxation (different rounding, same maths):
                         0: f* = (1-omega) f + omega feq        solver::collision, solver.cpp:73
                         1: f* = f + (-omega (f - feq))          cylinder_test.cpp:108,123-125 */
  /* constant body force as test/gravity_test.cpp applies it: force_mode = 1:
     u += (force_r, force_c) before the equilibrium (:146), and the source
     S_q = (1 - omega/2) w_q [(guo_a + guo_b c_q.u)(c_q.F) - guo_a u.F] (:154, (1/3, 1/9) there) is
     added to the post-collision populations: f* = f + (-omega (f - feq)) + S (:158-160);
     implies delta_form.  force_mode = 0: off. */
  int force_mode;
  double force_r, force_c, guo_a, guo_b;
  int form; /* LBM_FORM_*: REASSOCIATED on a delta_form lattice = the reassociated model standing in for the delta
               form (<= 1e-10 relative, not bitwise; the process-wide default needs "bgk_fast_delta" = 1 for that) */
} lbm_bgk_params;

/* P = collide(f): moments, equilibrium, collision of every node in place of one driver
 * iteration's calc_rho..collision calls; local, no streaming.  rho/u may be NULL. */
int lbm_bgk_collide(double* p, const double* f, const lbm_geom* g, const lbm_bc* bc,
                    const lbm_bgk_params* prm, double* rho, double* u, lbm_stream_t s);
/* The hot loop: p_new = collide(stream(p_old)) for rows [row_begin, row_end), boundary
 * fix-ups included.  p_old holds POST-collision populations (f_coll of the reference);
 * streaming happens at read time (pull).  rho/u (may be NULL) receive the moments of the
 * streamed state, i.e. what calc_rho/calc_u return at the top of the next driver iteration. */
int lbm_bgk_stream_collide(double* p_new, const double* p_old, const lbm_geom* g,
                           const lbm_bc* bc, const lbm_bgk_params* prm, int row_begin,
                           int row_end, double* rho, double* u, lbm_stream_t s);
/* Temporal blocking: p_new = TWO applications of the step above in one launch (an LDS tile keeps
 * the intermediate lattice on chip: 72 instead of 144 HBM bytes per lattice update; every node
 * undergoes the same arithmetic, results are bit-identical to two single steps).  Periodic or
 * ghost-row edges only (ghost = 0 or 2; with 2 the caller keeps TWO ghost rows per side current),
 * C % 64 == 0. */
int lbm_bgk_stream_collide_x2(double* p_new, const double* p_old, const lbm_geom* g,
                              const lbm_bc* bc, const lbm_bgk_params* prm, int row_begin,
                              int row_end, lbm_stream_t s);
/* Deeper temporal blocking: p_new = n_steps (2..6) applications of the step in one launch with
 * the register sliding-window kernel (a wavefront walks down a 64-column strip keeping the last
 * three rows of every intermediate step in registers; no LDS).  One lattice read + one write per
 * n_steps updates; results identical to n_steps single-step launches.  Edges: periodic or ghost rows
 * (ghost = 0 or >= n_steps); also bounce-back / specular columns and bounce-back / anti-bounce-back-
 * velocity rows (n_steps <= 5), applied inside the window at every level -- over slabs the ghost
 * rows must then be complete: exchange with LBM_HALO_FULL(n_steps). */
int lbm_bgk_stream_collide_xn(double* p_new, const double* p_old, const lbm_geom* g,
                              const lbm_bc* bc, const lbm_bgk_params* prm, int n_steps,
                              int row_begin, int row_end, lbm_stream_t s);
/* the same on TWO row ranges of equal height in one launch: [row_begin, row_end) and [row_begin2,
 * row_begin2 + (row_end - row_begin)), row_begin2 >= row_end -- the edge rows at both ends of a slab
 * ahead of the halo exchange (block binding, test/decompose_domain.cpp:181-187) */
int lbm_bgk_stream_collide_xn2(double* p_new, const double* p_old, const lbm_geom* g,
                               const lbm_bc* bc, const lbm_bgk_params* prm, int n_steps,
                               int row_begin, int row_end, int row_begin2, lbm_stream_t s);
/* f = stream(p) incl. boundary fix-ups == solver::advect + the driver's post-advect BCs. */
int lbm_stream(double* f, const double* p, const lbm_geom* g, const lbm_bc* bc, lbm_stream_t s);

/* ---- KBC entropic central-moment collision (src/ulbm.cpp; BASELINE config 3) -------------- */
typedef struct lbm_kbc_params {
  double s2; /* ulbm::d2q9::kbc ctor argument (src/ulbm.hpp:23) */
  int form;  /* LBM_FORM_*; multi-step launches exist for the reassociated collision only */
} lbm_kbc_params;
/* kbc::eval_equilibrium (ulbm.cpp:248-263) on SoA m0[R][C], m1[2][R][C].  zero_u2 != 0 is the
 * state in which the driver calls it (ux2 = uy2 = 0 from the ctor, ulbm_double_shear_flow.cpp:96) */
int lbm_kbc_equilibrium(double* feq, const double* m0, const double* m1, int R, int C,
                        int zero_u2, lbm_stream_t s);
/* kbc::collide (ulbm.cpp:91-126) with the moments the caller holds (unit parity) */
int lbm_kbc_collide_given_moments(double* coll, const double* f, const double* m0,
                                  const double* m1, const lbm_kbc_params* prm, int R, int C,
                                  lbm_stream_t s);
/* as the BGK pair above: moments recomputed from the populations
 * (ulbm_double_shear_flow.cpp:141-142), collide, (pull-)stream */
int lbm_kbc_collide(double* p, const double* f, const lbm_geom* g, const lbm_bc* bc,
                    const lbm_kbc_params* prm, double* rho, double* u, lbm_stream_t s);
int lbm_kbc_stream_collide(double* p_new, const double* p_old, const lbm_geom* g,
                           const lbm_bc* bc, const lbm_kbc_params* prm, int row_begin,
                           int row_end, double* rho, double* u, lbm_stream_t s);
/* n_steps = 2..4 time steps per launch (register sliding window, as lbm_bgk_stream_collide_xn) with the
 * reassociated KBC collision; periodic / halo edges, and on a single block bounce-back / specular /
 * velocity walls with n_steps <= 3 */
int lbm_kbc_stream_collide_xn(double* p_new, const double* p_old, const lbm_geom* g,
                              const lbm_bc* bc, const lbm_kbc_params* prm, int n_steps,
                              int row_begin, int row_end, lbm_stream_t s);
/* First iteration of a driver that HOLDS its moments (test/ulbm_poiseuille.cpp:85-86 starts from
 * adve_f = 0 with m0 = 1, m1 = 0): kbc::collide() on the given (m0 [R][C], m1 [2][R][C]) instead of
 * the populations' own moments, single block.  With bc->pressure_rows the KBC flavour of the
 * pressure-periodic rows is applied (ulbm_poiseuille.cpp:36-58: solver::incomp_equilibrium for the
 * imposed density, kbc.iequi_f.pow(-1) as f_equi); lbm_kbc_stream_collide applies it on the later
 * iterations. */
int lbm_kbc_collide_first(double* p, const double* f, const double* m0, const double* m1,
                          const lbm_geom* g, const lbm_bc* bc, const lbm_kbc_params* prm,
                          lbm_stream_t s);

/* ---- colour-gradient two-phase MRT (test/mrtcg_rayleigh_taylor.cpp; BASELINE config 4) ----
 * Two passes per step over post-collision populations of both colours:
 * lbm_cg_stream_moments (stream -> rho_r, rho_b, u) then lbm_cg_stream_collide (5x5 stencils
 * from an LDS tile, MRT + perturbation + recolouring + gravity source).  496 B/LUP + halo. */
typedef struct lbm_cg_colour {
  double rho_0; /* [red]/[blue] initial_density            (src/colour.cpp:12) */
  double alpha; /* alpha                                    (:13) */
  double nu;    /* kinematic_viscosity                      (:15) */
  double beta;  /* interface_thickness_control              (:17) */
} lbm_cg_colour;
typedef struct lbm_cg_params {
  lbm_cg_colour red, blue;
  double sigma;     /* [general] sigma              (mrtcg_rayleigh_taylor.cpp:360) */
  double gravity_r; /* Fg = (gravity_r, gravity_c): Rayleigh-Taylor driver ([general] gravity_magnitude, 0)
                       (:361,:403); static-droplet driver (0, -6.25e-6) (mrtcg_static_droplet.cpp:452) */
  double gravity_c;
  int add_source;   /* 1: the source term built from Fg is added to both colours (:460-464);
                       0: Fg only shifts the velocity, u += Fg/(2 rho) -- the static-droplet driver
                       comments the source out (mrtcg_static_droplet.cpp:513-514) */
  double delta;     /* interface half-width of the s_nu blend; the drivers hard-code 0.1 (:375) */
  int form;         /* LBM_FORM_* for lbm_cg_solver_step: REFERENCE_ORDER = the two-pass kernels (lbm_cg_stream_moments +
                       lbm_cg_stream_collide), REASSOCIATED = the one-launch step (lbm_cg_step_fused) */
} lbm_cg_params;
/* differential::x (dir 0, d/d row) and ::y (dir 1, d/d column): isotropic 5x5 finite
 * differences with replicate padding (src/differential.hpp:9-40, src/differential.cpp:3-33);
 * psi, out: [R][C].  Stand-alone parity operator; the fused step has its own LDS stencil. */
int lbm_diff5(double* out, const double* psi, int R, int C, int dir, lbm_stream_t s);
/* the driver's boundary set (:495-533): bounce-back rows, same-row column copy */
void lbm_cg_default_bc(lbm_bc* bc);
/* eval_equilibrium (:233-247): f[9][R][C] (plane stride as given) from rho_k[R][C], u[2][R][C] */
int lbm_cg_equilibrium(double* f, const double* rho_k, const double* u, const lbm_cg_colour* k,
                       int R, int C, long long plane_stride, lbm_stream_t s);
/* one loop body :431-464 on the GIVEN macroscopic fields and un-streamed populations (the
 * driver's first iteration: u = 0, rho_k = init); writes post-collision col_f of both colours */
int lbm_cg_collide(double* p_r, double* p_b, const double* f_r, const double* f_b,
                   const double* rho_r, const double* rho_b, const double* u, const lbm_geom* g,
                   const lbm_bc* bc, const lbm_cg_params* prm, double* psi /* may be NULL */,
                   double* s_nu /* may be NULL */, lbm_stream_t s);
/* Slabs: populations with lbm_geom.ghost = 3; the macroscopic fields then carry 2 ghost rows per
 * side (rho[(R+4)][C], u[2][(R+4)][C], row r at (r+2)*C): pass A recomputes them on rows -2..R+1
 * from the populations' ghost rows, so ONE population exchange per step (3 rows, both colours)
 * feeds both the streaming and the 5x5 stencils; the stencils clamp at GLOBAL edges only
 * (row_lo/row_hi != LBM_EDGE_HALO).  Single block: ghost = 0, fields [R][C]. */
/* pass A: :466-477 fused -- advect + BCs of both colours, rho_k, u incl. the Fg/(2 rho) shift */
int lbm_cg_stream_moments(double* rho_r, double* rho_b, double* u, const double* p_r,
                          const double* p_b, const lbm_geom* g, const lbm_bc* bc,
                          const lbm_cg_params* prm, lbm_stream_t s);
/* pass B: :431-464 on the streamed populations */
int lbm_cg_stream_collide(double* pn_r, double* pn_b, const double* p_r, const double* p_b,
                          const double* rho_r, const double* rho_b, const double* u,
                          const lbm_geom* g, const lbm_bc* bc, const lbm_cg_params* prm,
                          int row_begin, int row_end, double* psi, double* s_nu, lbm_stream_t s);
/* The whole step :431-477 in ONE launch (what lbm_cg_solver_step runs; tuning "cg_fused" = 0 selects
 * the two passes above).  Same mathematics, reassociated: the MRT operator is applied once to the
 * colour-summed non-equilibrium part (only the sum enters recolouring, :455), divisions by rho and
 * |grad psi| become reciprocals, rho_k / u / psi are rebuilt inside the tile from the streamed
 * populations (no macroscopic arrays are read; 288 B per node update instead of 496).  Results agree
 * with the two-pass form to rounding (see tests/test_gpu_cg.py for the stated tolerance), not bit for
 * bit.  Slabs: populations with 3 ghost rows, exactly as for the two passes.  The five field outputs
 * (rho_r, rho_b, u with the macro layout above; psi, s_nu dense [R][C]) are optional: all or none. */
int lbm_cg_step_fused(double* pn_r, double* pn_b, const double* p_r, const double* p_b,
                      const lbm_geom* g, const lbm_bc* bc, const lbm_cg_params* prm, int row_begin,
                      int row_end, double* rho_r, double* rho_b, double* u, double* psi, double* s_nu,
                      lbm_stream_t s);
/* the same step as TWO launches for a slab whose neighbours wait for its edge rows: LBM_CG_PART_FRAME = the boundary-gather
 * instantiation on the frame of the lattice widened to the first and last `edge_rows` rows (whole tiles of 16 rows),
 * LBM_CG_PART_INNER = everything else through the plain-offset inner kernel.  The parts write disjoint nodes and may run
 * on two streams at once (lbm_ring_cg_step: frame, pack and exchange on the ring's stream beside the inner launch);
 * together they are lbm_cg_step_fused on [0, R), bit for bit (test/mrtcg_rayleigh_taylor.cpp:431-477 per node). */
#define LBM_CG_PART_FRAME 1
#define LBM_CG_PART_INNER 2
int lbm_cg_step_fused_part(double* pn_r, double* pn_b, const double* p_r, const double* p_b,
                           const lbm_geom* g, const lbm_bc* bc, const lbm_cg_params* prm, int part,
                           int edge_rows, double* rho_r, double* rho_b, double* u, double* psi, double* s_nu,
                           lbm_stream_t s);
/* which kernel the calling thread's last lbm_cg_step_fused launched for the INNER rectangle of the lattice: 102 the
 * 16 x 64 tile kernel with two nodes per thread (the default since round 4; 101, 103..109: other shapes of an EXPERIMENTS
 * build, tuning "cg_big"), 0 the 16 x 32 tile kernel with one node per thread (tuning "cg_big" = 0, and every lattice too
 * small for an inner rectangle), 41..47 the walking block ("cg_big" = 0 and "cg_strip2"), other values the
 * strip kernels of an EXPERIMENTS build; -1 before the first call.  An opt-in form the geometry does not admit falls back to the tile kernel --
 * this says so (tests/test_gpu_cg.py asserts on it). */
int lbm_cg_last_inner_form(void);
/* driver loop context (single block); host arrays in the reference's shapes */
typedef struct lbm_cg_solver lbm_cg_solver;
int lbm_cg_solver_create(lbm_cg_solver** out, const lbm_geom* g, const lbm_bc* bc /* NULL = default */,
                         const lbm_cg_params* prm, lbm_stream_t s);
int lbm_cg_solver_destroy(lbm_cg_solver* sv);
int lbm_cg_solver_set_state(lbm_cg_solver* sv, const double* f_r, const double* f_b,
                            const double* rho_r, const double* rho_b, const double* u);
int lbm_cg_solver_step(lbm_cg_solver* sv, int n_steps);
/* how many two-step passes lbm_cg_solver_step has run so far.  EXPERIMENTS build only (0 otherwise): with tuning "cg_depth" = 2
 * single blocks with the driver's walls (rows >= 128, columns >= 352, multiple of 16) advance TWO steps per pass -- every
 * lattice row read once and written once per two steps (k_cg_two_step), the frame of the lattice through two single steps on
 * two small band lattices; same bits as single steps, measured SLOWER than them, hence not in the default build */
long long lbm_cg_solver_pair_launches(const lbm_cg_solver* sv);
int lbm_cg_solver_get_state(lbm_cg_solver* sv, double* f_r, double* f_b, double* rho_r,
                            double* rho_b, double* u, double* psi, double* s_nu);
int lbm_cg_solver_sync(lbm_cg_solver* sv);

/* ---- immersed boundary, multi-direct forcing (src/ibm.cpp; BASELINE config 5) ---------------
 * The reference loops over the markers on the HOST (~10 ATen launches per marker per forcing
 * iteration, ibm.cpp:166-187).  Here: markers, their 4x4 Peskin weight boxes and a per-node
 * CSR list of (marker, tap) pairs live on the device; one forcing iteration is two kernels
 * (interpolate + force per marker; gather-spread + velocity update per ROI node).  Spreading is
 * a GATHER in marker order, so the sums are bit-reproducible and equal the reference's
 * sequential accumulation -- no atomics.  The boundary is stationary (f_j = -2 rho_j u_j,
 * SURVEY Q10) and the weights keep the reference's transposed kernel (Q9). */
typedef struct lbm_ibm lbm_ibm;
/* x[i], y[i]: marker row / column coordinates (TOML arrays "x", "y", ibm.cpp:78-102);
 * m_max: ibm.hpp:25 default 5 (m_max - 1 forcing iterations); X, Y: lattice size */
int lbm_ibm_create(lbm_ibm** out, const double* x, const double* y, int n_markers, int m_max,
                   int X, int Y);
/* the same for a row slab that owns the whole ROI: X = rows of the slab, which start at global
 * row `row_offset`; x stays in GLOBAL coordinates (weights identical to the single-block ones),
 * lbm_ibm_roi and the u/rho/lattice arguments of the calls below are slab-local */
int lbm_ibm_create_slab(lbm_ibm** out, const double* x, const double* y, int n_markers, int m_max,
                        int X, int Y, int row_offset);
int lbm_ibm_destroy(lbm_ibm* ib);
/* EXTENSION without a reference counterpart (the reference's boundary is stationary, f_j = -2 rho_j u_j,
 * SURVEY Q10): a uniform marker velocity U_b = (Ur, Uc) -- the boundary drags the fluid along,
 * f_j = 2 rho_j (U_b - u_j), marker positions unchanged (a wall sliding in itself / a cylinder seen
 * from a moving frame).  (0, 0) restores the reference behaviour bit for bit. */
int lbm_ibm_set_velocity(lbm_ibm* ib, double Ur, double Uc);
/* region of interest rows [r0, r1), columns [c0, c1)  (ibm.cpp:104-156) */
int lbm_ibm_roi(const lbm_ibm* ib, int* r0, int* r1, int* c0, int* c1);
/* eulerian_force_density (ibm.cpp:158-190): u[2][X][Y], rho[X][Y] device SoA ->
 * F kept inside the context; F_out (may be NULL) receives it as [2][ROI_r][ROI_c] */
int lbm_ibm_force(lbm_ibm* ib, const double* u, const double* rho, double* F_out, lbm_stream_t s);
/* Guo source of the driver on the ROI (cylinder_test.cpp:116-127):
 * p[q][roi] += (1 - omega/2) w_q [ (a + b c_q.u)(c_q.F) - a u.F ]  with the UNcorrected u */
int lbm_ibm_add_source(lbm_ibm* ib, double* p, const lbm_geom* g, const double* u, double omega,
                       double a, double b, lbm_stream_t s);
/* lbm_ibm_force + lbm_ibm_add_source of one time step as TWO launches -- all forcing iterations in one
 * workgroup with its working set in LDS, then the source term on the touched nodes -- instead of
 * 2 (m_max-1) + 2 small dependent ones; bit-identical results.  What the solver context and the slab
 * engines call.  Tuning: "ibm_step_opt" 0 = the round-1 table layout, "ibm_step_split" 0 = source term
 * inside the workgroup, "ibm_step_chain" 1 = the launch chain. */
int lbm_ibm_step(lbm_ibm* ib, double* p, const lbm_geom* g, const double* u, const double* rho,
                 double omega, double a, double b, lbm_stream_t s);
/* F_s = sum over the ROI of F (drag, lift), cylinder_test.cpp:112; host out[2], synchronises */
int lbm_ibm_surface_force(lbm_ibm* ib, double* out2, lbm_stream_t s);

/* ---- solver context: one block, two lattices, the driver loop ----------------------------
 * Replaces the hand-written time loops of the reference drivers (e.g.
 * horizontal_poiseuille_test.cpp:100-153).  The context keeps POST-collision populations
 * resident and streams at read time, so n driver iterations cost one collide-only launch
 * plus n-1 fused launches; get_f streams lazily and returns the reference's f_adve. */
typedef struct lbm_solver lbm_solver;
enum lbm_model { LBM_MODEL_BGK = 0, LBM_MODEL_KBC = 1 };

int lbm_solver_create(lbm_solver** out, int model, const lbm_geom* g, const lbm_bc* bc,
                      const void* params /* lbm_bgk_params* or lbm_kbc_params* */, lbm_stream_t s);
int lbm_solver_destroy(lbm_solver* sv);
/* f_adve of the reference, host AoS [R][C][9] */
int lbm_solver_set_f_aos(lbm_solver* sv, const double* f_host);
int lbm_solver_get_f_aos(lbm_solver* sv, double* f_host);
/* same, device SoA [9][R][C] (no ghost rows) */
int lbm_solver_set_f_soa_dev(lbm_solver* sv, const double* f_dev);
/* KBC only, after set_f: the first iteration collides on these moments (host AoS rho [R][C],
 * u [R][C][2]) instead of the populations' own -- drivers that hold m0 / m1 (ulbm_poiseuille.cpp) */
int lbm_solver_set_moments_aos(lbm_solver* sv, const double* rho_host, const double* u_host);
int lbm_solver_get_f_soa_dev(lbm_solver* sv, double* f_dev);
/* n driver iterations; record_moments != 0: the last one also stores rho/u exactly as the
 * reference's rho/u tensors hold them when its loop has run n iterations */
int lbm_solver_step(lbm_solver* sv, int n, int record_moments);
/* host AoS rho[R][C], u[R][C][2] recorded by the last step(.., 1) */
int lbm_solver_get_moments_aos(lbm_solver* sv, double* rho_host, double* u_host);
int lbm_solver_sync(lbm_solver* sv);
/* how many multi-step blocks lbm_solver_step has launched so far (periodic / wall-bounded windows,
 * immersed-boundary blocks, pressure-row blocks): lets a caller or a test see that steps were fused
 * rather than silently run one per launch; -1 for NULL */
long long lbm_solver_block_launches(const lbm_solver* sv);
/* attach an immersed boundary to a BGK solver: every step then runs lbm_ibm_force on the
 * step's moments and adds the Guo source (a, b) on the ROI, as cylinder_test.cpp:110-127.
 * The solver does not take ownership. */
int lbm_solver_attach_ibm(lbm_solver* sv, lbm_ibm* ib, double guo_a, double guo_b);
/* device pointers of the resident lattices (current post-collision, scratch) for callers
 * that drive lbm_*_stream_collide themselves (benchmarks, multi-GPU slabs) */
int lbm_solver_lattices(lbm_solver* sv, double** cur, double** other, lbm_geom* geom /* may be NULL */);

/* ---- slab ring in C++: one process per GPU, packed halo messages between row slabs ------------------
 * Native counterpart of pylbm/slab.py (same kernels, same halo sets): edge rows + pack + ONE message
 * to and from each neighbour + unpack on the ring's own high-priority stream, interior rows on the
 * caller's stream (the block binding of test/decompose_domain.cpp:181-187, generalised to D ghost rows).
 * Two transports carry the messages:
 *   LBM_RING_RCCL  ncclSend / ncclRecv in one group (RCCL is dlopen()ed on first use): the default;
 *   LBM_RING_IPC   peer-mapped direct stores: every rank owns a receive window in device memory that its
 *                  neighbours map through hipIpcMemHandle_t; a message is stored straight into the
 *                  neighbour's window and announced by a sequence word (bounded device-side waits,
 *                  lbm_ring_status).  One node; also between processes that share ONE GPU, where RCCL
 *                  refuses to run.  Same bits either way.
 *   LBM_RING_DEFAULT = environment LBM_RING_TRANSPORT ("rccl" | "ipc"), else RCCL.
 * Rank 0 calls lbm_ring_unique_id(_ex) and distributes the 128 bytes by any means (file, MPI,
 * torch.distributed); every rank then calls lbm_ring_create(_ex) with its slab geometry (ghost = halo
 * depth; same columns and ghost rows on every rank). */
#define LBM_RING_DEFAULT (-1)
#define LBM_RING_RCCL 0
#define LBM_RING_IPC 1
typedef struct lbm_ring lbm_ring;
int lbm_ring_unique_id(unsigned char* id128);
int lbm_ring_unique_id_ex(unsigned char* id128, int transport);
int lbm_ring_create(lbm_ring** out, const unsigned char* id128, int rank, int nranks,
                    const lbm_geom* slab, int periodic);
int lbm_ring_create_ex(lbm_ring** out, const unsigned char* id128, int rank, int nranks,
                       const lbm_geom* slab, int periodic, int transport);
int lbm_ring_destroy(lbm_ring* rg);
int lbm_ring_transport(const lbm_ring* rg); /* LBM_RING_RCCL or LBM_RING_IPC */
/* LBM_OK, or LBM_ERR_STATE once a bounded wait of the peer-mapped transport has given up on a neighbour
 * (tuning "ring_ipc_timeout_ms", default 20000; the failure is sticky: every later wait of the queue returns at
 * once, copies are skipped, nothing more is announced) or once RCCL has reported an asynchronous error
 * (ncclCommGetAsyncError; such a communicator is aborted at lbm_ring_destroy): everything computed since is
 * void.  Host-side read, no sync; meant for once per batch of launches. */
int lbm_ring_status(const lbm_ring* rg);
/* 1 if the peer-mapped transport's receive window lives in ordinary cached device memory: the runtime refused the
 * uncached allocation and the caller had accepted that beforehand (lbm_set_tuning("ring_ipc_cached_ok", 1));
 * without that tuning lbm_ring_create_ex fails instead.  0 otherwise (and for RCCL). */
int lbm_ring_window_cached(const lbm_ring* rg);
/* refresh the ghost rows of `lattice` (ordered after the work enqueued on `after`); asynchronous */
int lbm_ring_exchange(lbm_ring* rg, double* lattice, lbm_stream_t after);
/* same with complete ghost rows (LBM_HALO_FULL): the initial fill before multi-step launches on a
 * slab with walls (lbm_ring_bgk_step then keeps exchanging complete rows by itself) */
int lbm_ring_exchange_full(lbm_ring* rg, double* lattice, lbm_stream_t after);
/* same for two lattices in one message per neighbour (both colours of the two-phase model) */
int lbm_ring_exchange2(lbm_ring* rg, double* lattice_a, double* lattice_b, lbm_stream_t after);
/* make `main` wait for the ring's stream */
int lbm_ring_join(lbm_ring* rg, lbm_stream_t main);
/* one overlapped launch-step of a BGK slab: n_steps = 1 (single-step kernel) or 2..ghost (sliding
 * window); bc: the physical edges of the GLOBAL domain (NULL = periodic), seams become HALO.
 * On a closed (periodic) ring whose slabs carry ghost = m x n_steps rows (m = 2, 3; ghost <= 15) only
 * every m-th call exchanges -- all m x n_steps ghost rows in one message; the calls in between are one
 * plain launch over the owned rows plus the ghost rows the later calls of the period still read.  The
 * owned rows are current after EVERY call; any lbm_ring_exchange* starts a new period.  lbm_ring_kbc_step
 * follows the same rule (ghost = m x n_steps, n_steps <= 4). */
int lbm_ring_bgk_step(lbm_ring* rg, double* dst, const double* src, const lbm_bc* bc,
                      const lbm_bgk_params* prm, int n_steps, int edge_rows, lbm_stream_t main);
/* the same for KBC (n_steps 1, or 2..4 with the reassociated collision) */
int lbm_ring_kbc_step(lbm_ring* rg, double* dst, const double* src, const lbm_bc* bc,
                      const lbm_kbc_params* prm, int n_steps, int edge_rows, lbm_stream_t main);

/* phase timing for diagnosing a scaling run: on = 1 records timed events around the edge rows, the
 * exchange (pack + send/recv + unpack) and the interior rows of every bgk / kbc launch-step;
 * lbm_ring_last_timing waits for the last one: out4 = {edge_rows_ms, exchange_ms, interior_ms, span_ms} */
int lbm_ring_profile(lbm_ring* rg, int on);
int lbm_ring_last_timing(lbm_ring* rg, double* out4);

/* one overlapped step of a two-phase (colour-gradient) slab: lbm_cg_step_fused on edge and interior
 * rows + ONE exchange of the 3 ghost rows of both colours (slab ghost must be 3; bc NULL = the
 * driver's walls, seams become HALO) */
int lbm_ring_cg_step(lbm_ring* rg, double* dst_r, double* dst_b, const double* src_r,
                     const double* src_b, const lbm_bc* bc, const lbm_cg_params* prm, int edge_rows,
                     lbm_stream_t main);

/* one overlapped single-step launch of a BGK slab that may own an immersed boundary (config 5 over
 * slabs): as lbm_ring_bgk_step with n_steps = 1, plus -- on the rank whose slab contains the ROI
 * (ib from lbm_ibm_create_slab; NULL elsewhere) -- rho / u written for the ROI rows only,
 * lbm_ibm_force and the Guo source on dst.  rho [R][C], u [2][R][C] slab-local. */
int lbm_ring_bgk_step_ibm(lbm_ring* rg, double* dst, const double* src, const lbm_bc* bc,
                          const lbm_bgk_params* prm, int edge_rows, lbm_ibm* ib, double guo_a,
                          double guo_b, double* rho, double* u, lbm_stream_t main);

/* ---- config 5 over slabs at multi-step speed: BGK slab with an immersed boundary anywhere, also across
 * a seam (cylinder_test.cpp:88-164 / ibm.cpp:158-190 over the block binding of decompose_domain.cpp:181-187).
 * Per block of D steps: a band of global rows [q0 - 2D, q1 + 2D) around the ROI advances D forced single
 * steps in a compact lattice of its own, replicated on every slab that owns valid band rows [q0 - D,
 * q1 + D); all other rows take the D-step window.  Across a seam inside the band the two co-owners swap
 * the band's D outermost rows instead of the ordinary halo (same size: 9 D rows of C doubles).  The
 * transport belongs to the caller (lbm_ring_bgk_block_ibm below uses RCCL): *_compute fills the two send
 * buffers, *_finish consumes the two receive buffers.  x, y: GLOBAL marker coordinates; slab: geometry with
 * ghost >= depth; the slab's rows start at global row slab_row0 of a domain of rows_global rows whose
 * physical edges are bc_global (seams become HALO).  Results equal the single-block solver bit for bit. */
typedef struct lbm_slab_ibm lbm_slab_ibm;
int lbm_slab_ibm_create(lbm_slab_ibm** out, const lbm_geom* slab, int slab_row0, int rows_global,
                        const lbm_bc* bc_global, const lbm_bgk_params* prm, int depth, const double* x,
                        const double* y, int n_markers, int m_max, double guo_a, double guo_b);
int lbm_slab_ibm_destroy(lbm_slab_ibm* sl);
/* slab heights for a chain of n_slabs slabs over rows_global rows such that all slabs finish a block together: the slab
 * that holds the forced band pays its chain of `depth` forced single steps on top of its rows, so it gets the band and few
 * other rows, and the band stays inside ONE slab (host arithmetic only; usable without a GPU).  Per-block cost model:
 * slabs without band rows far_us_per_row x rows, the owner owner_us + owner_us_per_row x rows; costs = {far_us_per_row,
 * owner_us, owner_us_per_row} or NULL = the table measured on MI355X (scaled with cols and depth).  x: GLOBAL marker rows.
 * rows_out[n_slabs] sums to rows_global; *predicted_us (may be NULL) = the slowest slab's block time under the model.
 * Every rank of a chain calls it with the same arguments and takes rows_out[rank]. */
int lbm_slab_ibm_plan_rows(int* rows_out, int n_slabs, int rows_global, int cols, int depth, const double* x,
                           int n_markers, const double* costs, double* predicted_us);
/* owner: this slab runs the band chain; straddle_*: the band's valid rows reach into that neighbour
 * (a co-owner); [b0, b1): global band rows.  Any output may be NULL. */
int lbm_slab_ibm_info(const lbm_slab_ibm* sl, int* owner, int* straddle_prev, int* straddle_next, int* b0, int* b1);
long long lbm_slab_ibm_msg_doubles(const lbm_slab_ibm* sl); /* per side per block */
/* once, on the initial post-collision state: ordinary seams exchange the complete D-row halo, co-owners
 * all their owned band rows (side 0 = previous slab, 1 = next; counts in doubles) */
int lbm_slab_ibm_prime_counts(const lbm_slab_ibm* sl, int side, long long* send, long long* recv);
int lbm_slab_ibm_prime_pack(lbm_slab_ibm* sl, const double* lattice, double* send_prev, double* send_next, lbm_stream_t s);
int lbm_slab_ibm_prime_finish(lbm_slab_ibm* sl, double* lattice, const double* recv_prev, const double* recv_next, lbm_stream_t s);
/* the driver's FIRST iteration (cylinder_test.cpp:103-127 on the initial state) instead of prime_finish:
 * prime_pack is then called on the PRE-collision lattice `pre`, whose ghost rows are filled here; `post`
 * = collision of every row + forcing and source on the band rows, ghost rows current, band primed */
int lbm_slab_ibm_start_finish(lbm_slab_ibm* sl, double* post, double* pre, const double* recv_prev,
                              const double* recv_next, lbm_stream_t s);
/* one block of `depth` steps: dst from src (ghost rows of src complete and current), messages packed */
int lbm_slab_ibm_block_compute(lbm_slab_ibm* sl, double* dst, const double* src, double* send_prev,
                               double* send_next, lbm_stream_t s);
int lbm_slab_ibm_block_finish(lbm_slab_ibm* sl, double* dst, const double* recv_prev, const double* recv_next, lbm_stream_t s);
int lbm_slab_ibm_surface_force(lbm_slab_ibm* sl, double* out2, lbm_stream_t s); /* owners: F_s of the last step */
/* n_rows complete rows (all 9 populations) between two lattices of equal column count; rows in owned-row
 * indices, ghost rows allowed where the geometry has them */
int lbm_rows_copy(double* dst, const lbm_geom* dg, int dst_row, const double* src, const lbm_geom* sg,
                  int src_row, int n_rows, lbm_stream_t s);
/* the same over the slab ring: priming exchange, then one block per call (non-owners: the overlapped
 * lbm_ring_bgk_step schedule; owners: band chain beside the far rows, exchange behind both) */
int lbm_ring_ibm_prime(lbm_ring* rg, lbm_slab_ibm* sl, double* lattice, lbm_stream_t main);
int lbm_ring_ibm_start(lbm_ring* rg, lbm_slab_ibm* sl, double* post, double* pre, lbm_stream_t main);
int lbm_ring_bgk_block_ibm(lbm_ring* rg, lbm_slab_ibm* sl, double* dst, const double* src, int edge_rows,
                           lbm_stream_t main);

/* ---- pressure-periodic rows over slabs (horizontal_poiseuille_test.cpp:25-45 over the block binding of
 * decompose_domain.cpp:50-73,181-187), blocks of `depth` steps.  The ring is PERIODIC: the virtual rows 0 / Rg-1 sit on
 * its two end slabs, which replicate the small seam lattice of lbm_solver_step's pressure blocks and swap, per block, the
 * D rows at distance [D, 2D) from the seam in place of that seam's halo (same size).  Transport-free: *_pack / *_compute
 * fill two send buffers, *_finish consume two receive buffers (side 0 = previous slab, 1 = next).  BGK; bc_global: the
 * domain's edges with pressure_rows = 1, periodic rows, periodic / wall columns.  Bitwise equal to the single block. */
typedef struct lbm_slab_pressure lbm_slab_pressure;
int lbm_slab_pressure_create(lbm_slab_pressure** out, const lbm_geom* slab, int slab_row0, int rows_global,
                             const lbm_bc* bc_global, const lbm_bgk_params* prm, int depth);
/* the same for KBC (test/ulbm_poiseuille.cpp:36-58, :85-139: KBC + pressure rows + wall columns): blocks of 2 steps (the
 * depth of the reference-order KBC window, as on one block); block_compute / block_finish / msg_doubles are shared.  The
 * driver's first iteration collides on HELD moments (kbc.m0, kbc.m1, :85-86): the start-up calls take them for the slab's
 * owned rows (m0 [R][C], m1 [2][R][C]); across the pressure seam the message carries the 2 D rows and their moments (12
 * planes).  After start_finish_kbc the ghost rows of `post` are not current: exchange complete halos (LBM_HALO_FULL(2)) over
 * the ordinary seams before the first block (lbm_ring_pressure_start_kbc does). */
int lbm_slab_pressure_create_kbc(lbm_slab_pressure** out, const lbm_geom* slab, int slab_row0, int rows_global,
                                 const lbm_bc* bc_global, const lbm_kbc_params* prm);
int lbm_slab_pressure_start_pack_kbc(lbm_slab_pressure* sl, const double* pre, const double* m0, const double* m1,
                                     double* send_prev, double* send_next, lbm_stream_t s);
int lbm_slab_pressure_start_finish_kbc(lbm_slab_pressure* sl, double* post, double* pre, const double* m0, const double* m1,
                                       const double* recv_prev, const double* recv_next, lbm_stream_t s);
int lbm_slab_pressure_destroy(lbm_slab_pressure* sl);
int lbm_slab_pressure_info(const lbm_slab_pressure* sl, int* R, int* C, int* ghost, int* depth); /* any output may be NULL */
long long lbm_slab_pressure_msg_doubles(const lbm_slab_pressure* sl, int side, int start /* 1: the start-up exchange */);
/* start-up from the driver's pre-collision state: exchange, then first iteration into `post` */
int lbm_slab_pressure_start_pack(lbm_slab_pressure* sl, const double* pre, double* send_prev, double* send_next, lbm_stream_t s);
int lbm_slab_pressure_start_finish(lbm_slab_pressure* sl, double* post, double* pre, const double* recv_prev,
                                   const double* recv_next, lbm_stream_t s);
int lbm_slab_pressure_block_compute(lbm_slab_pressure* sl, double* dst, const double* src, double* send_prev,
                                    double* send_next, lbm_stream_t s);
int lbm_slab_pressure_block_finish(lbm_slab_pressure* sl, double* dst, const double* recv_prev, const double* recv_next, lbm_stream_t s);
/* the same over the slab ring (created periodic): start-up, then one block per call (exchange behind the compute) */
int lbm_ring_pressure_start(lbm_ring* rg, lbm_slab_pressure* sl, double* post, double* pre, lbm_stream_t main);
int lbm_ring_bgk_block_pressure(lbm_ring* rg, lbm_slab_pressure* sl, double* dst, const double* src, lbm_stream_t main); /* BGK and KBC slabs alike */
/* KBC start-up over the ring: held moments of the owned rows; ends with the complete-halo exchange of `post` */
int lbm_ring_pressure_start_kbc(lbm_ring* rg, lbm_slab_pressure* sl, double* post, double* pre, const double* m0,
                                const double* m1, lbm_stream_t main);

/* ---- population links between lattices on one GPU (multi-block topologies) -----------------------
 * The reference glues blocks by slice assignments after advect (test/decompose_domain.cpp:181-187;
 * test/decompose_domain_loop.cpp:235-261, plus its slice-assignment walls :173-231): here a table of
 * element links built once, applied in ONE gather launch per step,
 *     dst[dst_lattice][q_dst][r0 + k dr][c0 + k dc] = src[src_lattice][q_src][sr0 + k sdr][sc0 + k sdc],
 * k = 0 .. count-1.  Slices are added in the order the driver executes them; where two write the
 * same destination element the later one wins (resolved on the host, the device pass is race-free).
 * dst / src of lbm_links_apply: one pointer per lattice (typically f_adve and f_coll of each block). */
typedef struct lbm_links lbm_links;
int lbm_links_create(lbm_links** out, int n_lattices /* <= 8 */, const lbm_geom* geoms);
int lbm_links_add(lbm_links* t, int dst_lat, int q_dst, int r0, int c0, int dr, int dc, int src_lat,
                  int q_src, int sr0, int sc0, int sdr, int sdc, int count);
/* affine links: dst = scale * src (+ addends[add0 + k add_stride]; add0 < 0: no addend).  scale = -1
 * with an addend is the anti-bounce-back "-f_coll + term" of rectangle_sedimentation_test.cpp:152-170;
 * the addend array is handed to lbm_links_apply_affine (it changes every step there). */
int lbm_links_add_affine(lbm_links* t, int dst_lat, int q_dst, int r0, int c0, int dr, int dc,
                         int src_lat, int q_src, int sr0, int sc0, int sdr, int sdc, int count,
                         double scale, long long add0, long long add_stride);
int lbm_links_apply_affine(lbm_links* t, double* const* dst, const double* const* src,
                           const double* addends, lbm_stream_t s);
int lbm_links_finalize(lbm_links* t);
int lbm_links_count(const lbm_links* t); /* distinct destination elements */
int lbm_links_apply(lbm_links* t, double* const* dst, const double* const* src, lbm_stream_t s);
int lbm_links_destroy(lbm_links* t);
/* per-row wall terms of rectangle_sedimentation_test.cpp, out[r][9], from the wall velocity
 * uw = wa u[:, r, col_a] + wb u[:, r, col_b] + shift of the moment field u [2][X][Y]:
 * mode 0: factor ((2 + 9 (uw.c_q)^2) - 3 uw.uw) w_q                       (anti-bounce-back, :135,:149)
 * mode 1: factor ((((1 + 3 uw.c_q) + 4.5 (uw.c_q)^2) - 1.5 uw.uw) w_q) field[r]   (concentration inlet, :202) */
int lbm_wall_terms(double* out, const double* u, int X, int Y, int col_a, double wa, int col_b, double wb,
                   double shift, int mode, const double* field, double factor, lbm_stream_t s);
/* one pressure-periodic virtual row at the operator level, possibly between two blocks
 * (test/decompose_domain.cpp:50-73; same lattice for dst and src: horizontal_poiseuille_test.cpp:25-45):
 * coll_dst[dst_row] = (feq(rho_bc, u_src[src_row]) + coll_src[src_row]) - equi_src[src_row];
 * u_src is the moment field [2][R][C] of the source block */
int lbm_pressure_row(double* coll_dst, const lbm_geom* g_dst, int dst_row, const double* coll_src,
                     const double* equi_src, const double* u_src, const lbm_geom* g_src, int src_row,
                     double rho_bc, int incompressible, lbm_stream_t s);
/* out = a * in + b, elementwise (the driver's "u + w_s", :124) */
int lbm_axpb(double* out, const double* in, double a, double b, long long n, lbm_stream_t s);
/* uniform momentum source on rows [row_begin, row_end) of a post-collision lattice
 * (decompose_domain_loop.cpp:152-160): p_q += ((1 - omega/2)((a + b u.c_q)(F.c_q) - a u.F)) w_q with the
 * step's u [2][R][C]; (a, b) = (3, 9) in that driver */
int lbm_bgk_add_force_rows(double* p, const lbm_geom* g, const double* u, double omega, double Fr,
                           double Fc, double a, double b, int row_begin, int row_end, lbm_stream_t s);

/* ---- snapshots and checkpoints (the reference only torch::save()s snapshot
 * stacks at the end of a run, e.g. horizontal_poiseuille_test.cpp:157-160) ------------------------ */
typedef struct lbm_snapshot lbm_snapshot;
int lbm_snapshot_create(lbm_snapshot** out, lbm_solver* sv);
int lbm_snapshot_destroy(lbm_snapshot* sn);
/* capture the moments of the last lbm_solver_step(.., record_moments = 1): device staging on the
 * solver's stream, device->pinned-host copy on a private stream; returns at once, the solver may
 * keep stepping */
int lbm_snapshot_record(lbm_snapshot* sn);
/* wait for that copy; write rho [R,C] / u [R,C,2] as NumPy .npy (either path may be NULL) */
int lbm_snapshot_write_npy(lbm_snapshot* sn, const char* rho_path, const char* u_path);
/* wait for that copy; host pointers (valid until the next record) and the step they belong to */
int lbm_snapshot_host(lbm_snapshot* sn, const double** rho, const double** u, long long* step);
/* the same for the two-phase solver: rho_r, rho_b [R,C] and u [R,C,2] of the CURRENT state (what the
 * driver holds after the iterations run so far); record() returns at once */
typedef struct lbm_cg_snapshot lbm_cg_snapshot;
int lbm_cg_snapshot_create(lbm_cg_snapshot** out, lbm_cg_solver* sv);
int lbm_cg_snapshot_destroy(lbm_cg_snapshot* sn);
int lbm_cg_snapshot_record(lbm_cg_snapshot* sn);
int lbm_cg_snapshot_host(lbm_cg_snapshot* sn, const double** rho_r, const double** rho_b,
                         const double** u, long long* step);
int lbm_cg_snapshot_write_npy(lbm_cg_snapshot* sn, const char* rho_r_path, const char* rho_b_path,
                              const char* u_path);
/* raw checkpoint of the resident lattice + state form + step counter + parameters; restart is
 * bitwise.  load needs a solver of the same model and size. */
int lbm_solver_checkpoint_save(lbm_solver* sv, const char* path);
int lbm_solver_checkpoint_load(lbm_solver* sv, const char* path);

/* Tuning table.  Launch-shape keys (every setting produces identical results) and the three
 * implementation switches, which select between a model's two collision implementations:
 *   "bgk_fast", "kbc_fast", "cg_fused" (default 1): the reassociated collision / the one-launch
 *   two-phase step; 0 = the reference's operation order, bit-identical to the CPU oracle.
 * Further keys: "kbc_depth" (steps lbm_solver_step fuses per launch for KBC, default 4; 3 with walls), "cg_tile"
 * (0: 8x32, 1: 16x32, 2: 8x64, 3 / 4 [default]: 16x32 budgeted for 3 / 4 waves per SIMD, 5: 32x32, 6: 16x64), "cg_xcd" (0: hardware order; 1: XCD-contiguous eighths; 2 [default] / 4 / 8: groups of that many
 * column-neighbour tiles per XCD), "cg_strip" (0 [default]: LDS tile kernel; 1/2/4:
 * column-strip sliding-window kernel with that many waves per workgroup, "cg_rows" rows per chunk),
 * "cg_split" (1 [default]: inner tiles through the boundary-free instantiation + a frame launch;
 * 0: one launch, every tile through the general boundary gather; same bits).
 * The environment variable LBM_TUNE="key=value,key=value" pre-loads the table (compiled drivers).
 * Launch-shape keys: "variant" (0 generic, 1 one node/thread grid-stride,
 * 2 two nodes/thread 16-B accesses, 3 [default] 2-D grid one node/thread), "nt" (bit 0
 * non-temporal loads, bit 1 non-temporal stores; default 3), "block" (128..1024, default 256),
 * "rows" (rows per thread 1/2/4, default 1), "grid_cap" (variants 1-2); two-step LDS kernel:
 * "tb_rows" (tile height, default 8), "tb_block" (default 512), "tb_order"; sliding-window kernel:
 * "sw_rows" (rows per wavefront chunk; default: fitted per launch to the resident wave slots, 64 when the launch is many rounds deep), "sw_waves" (waves per workgroup, default 4; 2 for
 * the reassociated BGK model), "sw_xcd" (G > 0: G consecutive strip groups per XCD; measured no effect);
 * "solver_depth" (steps lbm_solver_step fuses per launch on periodic BGK blocks, default 5, 1 =
 * off), "solver_depth_walls" (the same on wall-bounded blocks, default 5), "bgk_fast_delta" (0 [default]: delta-form BGK parameters always run the reference operation order;
 * 1: they may use the reassociated model too, 1e-10 instead of bitwise on the cylinder preset), "ibm_depth" (steps lbm_solver_step advances per block on a BGK lattice with an immersed boundary: forced
 * band around the ROI in single steps, rows at least that far away through the multi-step window; default 5,
 * 1 = one step per launch everywhere; same bits), "pressure_depth" (steps per block on lattices with pressure-periodic rows: the 2 D rows on either side of the virtual rows in single steps on a small periodic
 * lattice beside the D-step window on all other rows; default 5 for BGK, 2 for KBC, 1 = one step per launch; same bits), "halo_grid" (workgroup cap of the halo pack / unpack copies, default 256), "ibm_box" (0: the forced single steps of an immersed-boundary block run over full-width band rows instead of a box of ROI +- 2 D rows and columns), "ibm_box_overlap" (0: box chain and window launch one after the other on the caller's stream), "ibm_box_sole" (0: a slab that owns the whole forced band alone still goes through the band lattice), "ibm_chain_kernel" (1: the forced-box chain of a block as ONE launch of "ibm_chain_wgs" [16] workgroups with grid barriers instead of 3 D launches; same bits, level), "bg_priority" (1: the background stream of those window launches gets the lowest priority; default 0: such a queue starves while any other queue of the process has work), "ring_period" (1: lbm_ring_bgk_step exchanges on every launch even when the slabs carry m x n_steps ghost rows; default 0 = one exchange per m launches), "ibm_gate" (1 [default]: lbm_solver_step holds its lattice launches behind a one-wave gate until the
 * forcing workgroup is resident; 0: off), "sw_split" (1 [default]: wall-bounded
 * multi-step launches run their wall-free interior through the plain instantiation and only the frame of
 * outermost strips / rows next to a wall row through the wall-carrying one, on a helper stream; 0: one
 * wall-carrying launch; same bits).  value < 0 restores the default.  Measurements: DESIGN.md "BGK kernel variants". */
int lbm_set_tuning(const char* key, int value);
/* 1 when the library was built with EXPERIMENTS=1: the launch forms that were measured and not kept (tuning keys "sw_pair",
 * "sw_pf2", "ibm_chain_kernel", "ring_edges_main", "cg_strip", "cg_strip2" = 1 / 2 / 4, "cg_merge") exist; 0: those keys are
 * ignored */
int lbm_build_has_experiments(void);
/* an empty one-thread kernel ("k_lbm_marker") to cut a profiler trace at: measurement harnesses bracket the launches
 * whose counters they sum with two of these */
int lbm_marker(int tag, lbm_stream_t s);
int lbm_get_tuning(const char* key);

#ifdef __cplusplus
}
#endif
#endif /* LBM_HIP_H */
