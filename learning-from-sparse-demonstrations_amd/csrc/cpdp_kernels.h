// Batched Continuous-PDP kernels for gfx950 (MI355X).
//
// One *lane group* of G lanes (G | 64, one 64-lane wavefront per workgroup) owns
// one trajectory.  Matrices of the per-trajectory recursions are held
// "column per lane": lane j keeps column j of the n x (n+m) shooting
// sensitivity, of V_xx, of the Riccati pair Z = [P W], of the auxiliary state
// X = dx/dtheta.  Quantities every column needs (the current state, costate,
// gains, packed Jacobian/Hessian entries) are group-uniform and are exchanged
// through LDS.  Per-trajectory records in HBM are trajectory-major, so each
// wavefront streams its own contiguous record.
//
// What each kernel replaces in the reference (CPDP/CPDP.py):
//   oc_solve_kernel       COCSys.cocSolver            CPDP.py:92-198  (IPOPT NLP solve ->
//                         batched DDP on the identical RK4 multiple-shooting discretisation;
//                         returns state/control/costate grids, costate == lam_g)
//   aux_riccati_kernel    COCSys.auxSysSolver part 1  CPDP.py:316-338 (Riccati sweep for P, W)
//   aux_forward_kernel    COCSys.auxSysSolver part 2  CPDP.py:340-381 (dx/dtheta forward sweep)
//                         + getloss_*corrections      lib/QuadAlgorithm.py:616-673
//   optimizer_kernel      Vanilla/Nesterov/Adam/Nadam/AMSGrad updates  lib/QuadAlgorithm.py:454-578
//
// The same source builds for the GPU with hipcc and, with -DLFSD_EMU, for the CPU
// SIMT emulator in tests/emu (test infrastructure; never used by the product path).
//
// Sources: cpdp_common.h (switches, primitives, dense helpers), cpdp_oc.h (OC solve), cpdp_aux.h (auxiliary
// system sweeps + loss), cpdp_opt.h (update rules).
#pragma once
#include "cpdp_common.h"
#include "cpdp_oc.h"
#include "cpdp_aux.h"
#include "cpdp_opt.h"
