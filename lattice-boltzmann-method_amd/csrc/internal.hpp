// Host-side plumbing shared by the C-ABI translation units: per-thread error record,
// HIP status checks, launch-geometry helpers and the tuning table.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/lbm_hip.h"

namespace lbm {

void set_error(const char* fmt, ...);
int tuning(const char* key, int dflt);
// capi_ibm.hip: one-wave kernel on `st` that ends once the last lbm_ibm_step's workgroup is resident
int ibm_gate(lbm_ibm* ib, hipStream_t st);
// capi_kbc.hip: 2 time steps per launch through the sliding window with the REFERENCE-ORDER KBC model
// (lattices with pressure rows keep that order: solver_pressure_block); rows [row_begin, row_end)
int kbc_stream_collide_x2_ref(double* p_new, const double* p_old, const lbm_geom* g, const lbm_bc* bc,
                              const lbm_kbc_params* prm, int row_begin, int row_end, hipStream_t st);
// capi_bgk.hip: lbm_bgk_stream_collide_xn pinned to the reference operation order
int bgk_stream_collide_xn_ref(double* p_new, const double* p_old, const lbm_geom* g, const lbm_bc* bc,
                              const lbm_bgk_params* prm, int n_steps, int row_begin, int row_end, hipStream_t st);
int bgk_collide_ref(double* p, const double* f, const lbm_geom* g, const lbm_bgk_params* prm, hipStream_t st);
// capi_ibm.hip: lbm_ibm_step on a WINDOW of the lattice the boundary was created for -- rows [row_off, row_off + g->R),
// columns [col_off, col_off + g->C) of it, held as a lattice of its own (p, u [2][R][C], rho [R][C] are the window's).
// Same tables, same arithmetic; only the origin the kernels add to ROI indices moves.
int ibm_step_window(lbm_ibm* ib, int row_off, int col_off, double* p, const lbm_geom* g, const double* u,
                    const double* rho, double omega, double a, double b, hipStream_t st);
// capi_ibm.hip: the D forced single steps of a box lattice pair (box step + forcing + source term, D times) as ONE
// launch of a few workgroups on compute units of their own; *cur flips when D is odd.  1 = not applicable.
int ibm_box_chain(lbm_ibm* ib, int row_off, int col_off, double* const box[2], int* cur, const lbm_geom* g,
                  const lbm_bgk_params* prm, bool fast_model, int D, double* xrho, double* xu, double a, double b,
                  hipStream_t st);
// capi_kbc.hip: the same question for KBC
bool kbc_uses_fast_model(const lbm_kbc_params* prm);
// capi_bgk.hip: the model lbm_bgk_stream_collide picks for these parameters (reassociated or reference order)
bool bgk_uses_fast_model(const lbm_bgk_params* prm, const lbm_bc* bc);
// capi_core.hip: a box of n_rows x n_cols nodes, all 9 populations, between two lattices (rows in owned-row
// indices, ghost rows allowed)
int box_copy(double* dst, const lbm_geom& dg, int dst_row, int dst_col, const double* src, const lbm_geom& sg,
             int src_row, int src_col, int n_rows, int n_cols, hipStream_t st);
// capi_core.hip: a non-blocking stream for a grid-filling launch that runs beside a chain of small dependent kernels
// on the caller's stream ("bg_priority" = 1: of the lowest priority, so that the dispatcher hands freed wave slots to
// the chain first -- off by default: such a queue starves whenever another queue of the process has work).
// (A stream that spares one compute unit for the chain -- hipExtStreamCreateWithCUMask -- was tried and is far slower:
// profiles/r02_ibm_box_bench.log.)
int make_background_stream(hipStream_t* out);
// NumPy .npy (v1.0, little-endian f64, C order) writer shared by the snapshot objects
int write_npy(const char* path, const double* data, const std::vector<long>& shape);

#define LBM_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      ::lbm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,  \
                       __LINE__);                                                        \
      return LBM_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

#define LBM_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      ::lbm::set_error(__VA_ARGS__);  \
      return LBM_ERR_INVALID;         \
    }                                 \
  } while (0)

#define LBM_CHECK_LAUNCH() LBM_CHECK_HIP(hipGetLastError())

// Launch with the thread's sticky error cleared first: the host process (e.g. PyTorch) may
// have left an unrelated, already-handled error behind that hipGetLastError would report.
#define LBM_KLAUNCH(...)               \
  do {                                 \
    (void)hipGetLastError();           \
    hipLaunchKernelGGL(__VA_ARGS__);   \
  } while (0)

inline hipStream_t as_stream(lbm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Grid for a memory-bound grid-stride kernel: enough blocks to fill 256 CUs x 8, no more
// (cdna_hip_programming.md Guideline 11).
inline int capped_grid(long work_items, int cap = 2048) {
  if (work_items < 1) work_items = 1;
  return (int)(work_items < cap ? work_items : cap);
}

}  // namespace lbm
