// Internal: per-rank state of a BGK row slab that may (co-)own the band of rows around an immersed
// boundary (capi_slab_ibm.hip; the RCCL transport on top of it lives in capi_ring.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/lbm_hip.h"

struct lbm_slab_ibm {
  lbm_geom g;            // slab geometry, ghost >= depth
  int row0, rows_global; // global row of slab row 0; rows of the whole domain
  lbm_bc bc_global, bc;  // the domain's edges; this slab's (seams = HALO)
  lbm_bgk_params prm;
  int D;                 // steps per block
  bool has_prev, has_next;
  // the band: global rows [b0, b1) = ROI +- 2 D; rows [b0 + D, b1 - D) are valid after a block
  bool owner;            // the valid band rows intersect this slab's owned rows
  bool straddle_prev, straddle_next;  // ... and reach into the previous / next slab (co-owner there)
  int b0, b1;
  lbm_geom bg;           // band lattice: b1 - b0 rows, periodic (the wrap only ever reaches rows that are dropped)
  lbm_bc bbc;
  double* blat[2];
  int bcur;
  double *brho, *bu;
  lbm_ibm* ib;           // created in band-local rows
  double ga, gb;
  double* stash;         // [9][D][C]: the outer band rows the co-owner computed (state after its last block)
  hipStream_t aux;       // band chain, beside the far rows on the caller's stream
  hipEvent_t ev_fork, ev_join;
  // the forced BOX inside the band: rows of the band x columns ROI +- 2 D (widened to multiples of 8), a small
  // periodic lattice pair for the D forced single steps; everything else in the band takes the D-step window
  bool boxed;
  int bc0, bc1;          // box columns [bc0, bc1) of the lattice
  lbm_geom xg;           // box lattice: band rows x (bc1 - bc0) columns
  double* box[2];
  double *xrho, *xu;
  hipStream_t bgst;      // the window launches (band + far rows) beside the box chain
  bool blat_stale;       // a sole owner's boxed blocks work on the slab lattice directly: the band lattice is behind
};
