/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's
 * collide-then-stream hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (lattice-boltzmann-method_amd/) never links, imports or calls it.
 *
 * Layout is the REFERENCE layout (AoS, q innermost): f[R][C][9], rho[R][C],
 * u[R][C][2]; f64 everywhere.  Every function cites the reference lines it
 * restates.  Pinning status per function: see oracle/README.md.
 */
#ifndef LBM_ORACLE_H
#define LBM_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

int  orc_max_threads(void);
void orc_set_threads(int n);

/* ---- solver:: (src/solver.cpp:23-131) ---- */
void orc_calc_rho(double* rho, const double* f, int R, int C);
void orc_calc_u(double* u, const double* f, const double* rho, int R, int C);
void orc_calc_incomp_u(double* u, const double* f, int R, int C);
void orc_equilibrium(double* feq, const double* u, const double* rho, int R, int C);
void orc_incomp_equilibrium(double* feq, const double* u, const double* rho, int R, int C);
void orc_collision(double* fc, const double* f, const double* feq, double omega, int R, int C);
void orc_advect(double* g, const double* f, int R, int C);

/* BGK periodic box loop: calc_rho, calc_(incomp_)u, (incomp_)equilibrium, collision, advect. */
void orc_bgk_periodic_steps(double* f, double* rho_out, double* u_out, int R, int C,
                            double omega, int incompressible, int nsteps);

/* ---- test/horizontal_poiseuille_test.cpp:47-175 (H, W, T parametrised) ---- */
typedef struct {
  int H, W, T;
  double omega, u_max, rho_inlet, rho_outlet;
  int check_convergence; /* reproduce :113-126 early exit */
} orc_hpt_params;
/* returns the number of steps executed; f/u/rho = state when the loop ended. */
int orc_hpt_run(const orc_hpt_params* p, double* f, double* u, double* rho, double* l2_out);

/* ---- the remaining single-phase BGK drivers -------------------------------- */
/* test/specular_boundary_test.cpp:47-139: compressible BGK, pressure-periodic rows with the
 * compressible equilibrium (:23-45), specular columns (:121-127).  f/u/rho = state after T steps. */
void orc_sbt_run(int H, int W, int T, double omega, double rho_inlet, double rho_outlet,
                 double* f, double* u, double* rho);
/* test/free_stream_test.cpp:75-131 loop body, nsteps times: incompressible BGK, anti-bounce-back
 * rows with u_w = (uw, 0), specular columns.  f in/out. */
void orc_free_stream_steps(double* f, double* u, double* rho, int X, int Y, double omega,
                           double uw, int nsteps);
/* test/gravity_test.cpp:62-181: incompressible BGK with body force Fg = (Fr, Fc): u += Fg (:142),
 * relaxation in delta form plus the Guo-type source S with 1/3, 1/9 (:146-158), pressure-periodic
 * rows (rho_in = rho_out), halfway bounce-back columns, the reference's convergence rule.
 * Returns the number of steps executed. */
int orc_gravity_run(int H, int W, int T, double omega, double Fr, double Fc, double rho_inlet,
                    double rho_outlet, int check_convergence, double* f, double* u, double* rho);

/* ---- test/decompose_domain.cpp:75-188: two blocks A (upstream) and B ---- */
void orc_ddm_run(int H, int W, int T, double omega, double rho_inlet, double rho_outlet,
                 double* fA, double* fB, double* uA, double* uB, double* rhoA, double* rhoB);

/* ---- test/decompose_domain_loop.cpp: four blocks A (L x L/4), B (L/4 x L/2), C, D closed into a loop
 * channel by column-seam bindings with row offsets (:235-261), no-slip walls (:173-231), momentum
 * source F = (Fr, 0) with the (3, 9) coefficients on rows [L/4+5, L/4+55) of A (:63,:152-160).
 * nsteps iterations from the driver's start state; rho[k] / u[k] = moments computed in the last
 * iteration (k = 0..3 for A..D). */
void orc_ddl_run(int L, int nsteps, double omega, double Fr, double* fA, double* fB, double* fC,
                 double* fD, double* rho[4], double* u[4]);

/* ---- test/rectangle_sedimentation_test.cpp:106-237: fluid f + sediment concentration g (settling
 * velocity w_s added to BOTH velocity components, :124), anti-bounce-back inlet / outlet columns,
 * zero-gradient copies on g_coll, specular top, no-slip bottom, the hard-coded rectangle (:71-73;
 * needs X > 151, Y > 250).  PARITY UNPINNED: the driver needs toml++ (params). init != 0: build the
 * start state (:79-104) first. */
void orc_sed_steps(int X, int Y, double omega, double u_in, double w_s, double Cw, int init,
                   int nsteps, double* f, double* g, double* rho, double* u, double* C);

/* ---- ulbm::d2q9::kbc (src/ulbm.cpp:91-320) ---- */
/* feq from (m0, ux, uy) with the caller-supplied ux2/uy2 (the ctor leaves them 0
 * for the driver's initialisation, ulbm_double_shear_flow.cpp:96). */
void orc_kbc_equilibrium(double* feq, const double* m0, const double* m1,
                         int use_zero_u2, int R, int C);
void orc_kbc_collide(double* coll, const double* f, const double* m0, const double* m1,
                     double s2, int R, int C, double* gamma_out);
/* collide, advect, moment update (ulbm_double_shear_flow.cpp:119-142). */
void orc_kbc_steps(double* f, double* m0, double* m1, int R, int C, double s2, int nsteps);
/* test/ulbm_poiseuille.cpp:104-141 loop body, nsteps times (KBC + pressure-periodic rows with the
 * incompressible equilibrium + halfway bounce-back columns); init != 0: start from the driver's
 * state (adve_f = 0, m0 = 1, m1 = 0). */
void orc_upo_steps(double* f, double* m0, double* m1, int H, int W, double s2, double rho_inlet,
                   double rho_outlet, int init, int nsteps);
/* shear-layer IC, ulbm_double_shear_flow.cpp:42-63 */
void orc_kbc_shear_init(double* m0, double* m1, int R, int C, double u_max,
                        double alpha, double delta);

/* ---- differential (src/differential.hpp:9-40, src/differential.cpp:3-33) ---- */
void orc_diff_x(double* out, const double* psi, int R, int C);
void orc_diff_y(double* out, const double* psi, int R, int C);

/* ---- colour-gradient MRT two-phase step (test/mrtcg_rayleigh_taylor.cpp) ---- */
typedef struct {
  double rho_0, alpha, nu, beta; /* [red]/[blue] TOML keys, src/colour.cpp:11-20 */
} orc_colour_params;
typedef struct {
  int R, C;
  orc_colour_params red, blue;
  double sigma;          /* [general] sigma (:360); 0.1 hard-coded in mrtcg_static_droplet.cpp:439 */
  double g_r, g_c;       /* Fg = (g_r, g_c): (gravity_magnitude, 0) (:361,:403) or (0, -6.25e-6) (droplet :452) */
  int add_source;        /* 0: the droplet driver comments the source term out (:513-514) */
  double delta;          /* hard-coded 0.1 at :375 */
} orc_cg_params;
/* init_rho_cosine + feq init (:182-210, :372-373, :407-410): fills f_r, f_b (adv_f),
 * rho_r, rho_b, u (= 0). */
void orc_cg_init(const orc_cg_params* p, double* f_r, double* f_b,
                 double* rho_r, double* rho_b, double* u);
/* test/mrtcg_static_droplet.cpp:182-204, :420-421, :456-459: sigmoid droplet of radius 25 centred at
 * (R/2, R/2) ("center" = R/2 is used for both coordinates), u = 0 + Fg/(2 rho) BEFORE the
 * equilibrium initialisation. */
void orc_cg_init_droplet(const orc_cg_params* p, double* f_r, double* f_b,
                         double* rho_r, double* rho_b, double* u);
/* nsteps iterations of the loop body :431-477. State in/out: adv_f of both colours,
 * rho_r, rho_b, u[R][C][2].  Optional outputs (may be NULL): psi, s_nu of the last step,
 * col_r/col_b = post-collision populations of the last step. */
void orc_cg_steps(const orc_cg_params* p, double* f_r, double* f_b, double* rho_r,
                  double* rho_b, double* u, int nsteps, double* psi_out, double* snu_out,
                  double* col_r_out, double* col_b_out);

/* ---- immersed boundary (src/ibm.cpp) + cylinder driver (test/cylinder_test.cpp) ---- */
typedef struct {
  int n_markers;
  const double* x; /* marker row coordinates (TOML array "x") */
  const double* y; /* marker column coordinates */
  int m_max;       /* ibm.hpp:25 default 5 */
} orc_ibm_markers;
/* ROI per ibm.cpp:104-156: rows [r0, r1), cols [c0, c1). */
void orc_ibm_roi(const orc_ibm_markers* m, int* r0, int* r1, int* c0, int* c1);
/* eulerian_force_density (ibm.cpp:158-190): u[X][Y][2], rho[X][Y] -> F[ROI_r][ROI_c][2]. */
void orc_ibm_force(const orc_ibm_markers* m, const double* u, const double* rho,
                   int X, int Y, double* F);
/* cylinder_test.cpp:88-164 loop body, nsteps times. f in/out = f_adve; u/rho = last
 * moments; Fs = last surface force [2]. */
void orc_cylinder_steps(const orc_ibm_markers* m, double* f, double* u, double* rho,
                        int X, int Y, double omega, double u_in, int nsteps, double* Fs);

#ifdef __cplusplus
}
#endif
#endif
