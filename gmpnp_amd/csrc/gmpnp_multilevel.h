// Geometric multilevel term of the preconditioner on uniformly refined meshes (round 3; no reference counterpart: the reference's
// linear solver is a direct one, 3D:792 — this is what keeps the Krylov stand-in's iteration count from growing with 1/h).
//
// Red refinement gives NESTED P1 spaces: a fine vertex is a coarse vertex or the midpoint of a coarse edge, so the prolongation
// P_l (level l <- level l+1) is "copy, or mean of the two parents", applied field by field.  The preconditioner of the library,
//     M^-1 = Dinv (I + Ps Aci Ps^T)                       node-block Jacobi + slab-aggregate coarse space (gmpnp_kernels.h)
// gets the term
//     M^-1 += theta * P_1 S_1 P_1^T                       additive on the FINEST level (no second fine SpMV per application)
// where S_1 is one V(1,1) cycle over the coarser levels (gmpnp_api.hip::ml_level_apply): damped node-block Jacobi on the
// intermediate levels, the level's own two-level preconditioner (Jacobi + slabs) on the coarsest, which repeats its smoothing
// step `sweeps` times; every cycle costs two SpMVs per level, each 1/8 of the next finer level's.  The level operators are the
// Jacobians REDISCRETISED at the injected state: every level is an ordinary handle of its own mesh, assembled by the element
// kernel and the gathers of this library (tools/multilevel_experiment.py: rediscretised and Galerkin operators give the same
// counts).  Dirichlet dofs are masked on both sides of every transfer.  In the scaled system As N y = b (As = J Dinv) the term
// reads N += theta * D P S P^T, D = the diagonal node blocks of J: the half-iterations stage  z = vec + theta * D (P S P^T vec)
// instead of vec (materialised vector form), and the solution gets x += theta * P S P^T y at the end.  All kernels here are
// gathers: fixed summation order, bitwise repeatable.  Measured (profiles/r03/multilevel_refine{1,2}.json): BiCGStab iterations
// per solve 104 -> 31-40 at one refinement, 206 -> 25 at two, independent of h; Newton iterations 3.6x faster at two refinements.
#pragma once
#include "gmpnp_kernels.h"

namespace gmpnp {

// coarse[Ic] = fine[copy_of[Ic]]   (state injection: coarse vertices ARE fine vertices)
template <int NF>
__global__ __launch_bounds__(256) void k_ml_inject(const double* __restrict__ fine, const int32_t* __restrict__ copy_of, double* __restrict__ coarse, int ndof_c) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ndof_c) return;
  const int I = i / NF, f = i - I * NF;
  coarse[i] = fine[(size_t)copy_of[I] * NF + f];
}

// r_c = mask_c P^T mask_f src: one thread per coarse dof sums its children (coarse node -> fine nodes, weights 1 or 1/2) in list order
template <int NF>
__global__ __launch_bounds__(256) void k_ml_restrict(const double* __restrict__ src, const uint8_t* __restrict__ bc_f, const int32_t* __restrict__ child_ptr,
                                                     const int32_t* __restrict__ child, const uint8_t* __restrict__ bc_c, double* __restrict__ dst, int ndof_c) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ndof_c) return;
  const int I = i / NF, f = i - I * NF;
  double acc = 0.0;
  if (!bc_c[i])
    for (int k = child_ptr[I]; k < child_ptr[I + 1]; ++k) {
      const int code = child[k];                      // fine node << 1 | (1 = weight 1/2)
      const size_t j = (size_t)(code >> 1) * NF + f;
      const double v = bc_f[j] ? 0.0 : src[j];
      acc += (code & 1) ? 0.5 * v : v;
    }
  dst[i] = acc;
}

// dst = scale_dst * dst + omega * Dinv src: damped node-block Jacobi, the smoother of the INTERMEDIATE levels (one launch; the slab
// coarse space of a level's own preconditioner costs two or three more and is left to the coarsest level)
template <int NF>
__global__ __launch_bounds__(256) void k_ml_jacobi(const double* __restrict__ Dinv, const double* __restrict__ src, double* __restrict__ dst, double scale_dst,
                                                   double omega, int ndof) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ndof) return;
  const int I = i / NF;
  const double* d = Dinv + (size_t)i * NF;
  const double* x = src + (size_t)I * NF;
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < NF; ++j) acc += d[j] * x[j];
  dst[i] = (scale_dst != 0.0 ? scale_dst * dst[i] : 0.0) + omega * acc;
}

// (P w_c)_i for fine dof i: copy of a coarse vertex, or the mean of the two parents
template <int NF>
__device__ __forceinline__ double ml_prolonged(const double* __restrict__ wc, const int32_t* __restrict__ par, int I, int f) {
  const int pa = par[2 * I], pb = par[2 * I + 1];
  const double a = wc[(size_t)pa * NF + f];
  return pb < 0 ? a : 0.5 * (a + wc[(size_t)pb * NF + f]);
}
// w_f += mask_f P w_c   (coarse-grid correction of a V-cycle)
template <int NF>
__global__ __launch_bounds__(256) void k_ml_prolong_add(const double* __restrict__ wc, const int32_t* __restrict__ par, const uint8_t* __restrict__ bc_f,
                                                        double* __restrict__ wf, int ndof_f) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ndof_f) return;
  const int I = i / NF, f = i - I * NF;
  if (!bc_f[i]) wf[i] += ml_prolonged<NF>(wc, par, I, f);
}
// finest level, inside the Krylov loop:  z = vec + theta * D (mask P w_c)   (D = diagonal node blocks of the UNSCALED Jacobian,
// SELL position 0: c.vals[slice_off[s] + Iloc*NF + row + col*64]).  A workgroup holds whole nodes (kMlStageNodes x NF threads): every
// thread prolongs ITS dof once, the node's NF values meet in LDS, then every thread multiplies its row of D (one gather per dof
// instead of NF: 440 -> 300 us on the three-times-refined mesh).
constexpr int kMlStageNodes = 28;
template <int NF>
__global__ __launch_bounds__(kMlStageNodes* NF) void k_ml_stage(const Ctx c, const double* __restrict__ wc, const int32_t* __restrict__ par,
                                                                 const double* __restrict__ vec, double* __restrict__ z, double theta) {
  __shared__ double t[kMlStageNodes * NF];
  const int i = blockIdx.x * (kMlStageNodes * NF) + threadIdx.x;
  const bool on = i < c.ndof;
  const int I = on ? i / NF : 0, r = on ? i - I * NF : 0;
  t[threadIdx.x] = (on && !c.bcflag[i]) ? ml_prolonged<NF>(wc, par, I, r) : 0.0;
  __syncthreads();
  if (!on) return;
  const int s = c.node_slice[I], Iloc = I - c.slice_node0[s];
  const double* d = c.vals + c.slice_off[s] + Iloc * NF + r;
  const double* tn = t + (threadIdx.x - r);
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < NF; ++j) acc += d[(size_t)j * kWave] * tn[j];
  z[i] = vec[i] + theta * acc;
}
// finest level, end of a solve:  x += theta * mask P w_c
template <int NF>
__global__ __launch_bounds__(256) void k_ml_add_solution(const Ctx c, const double* __restrict__ wc, const int32_t* __restrict__ par, double* __restrict__ x, double theta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= c.ndof) return;
  const int I = i / NF, f = i - I * NF;
  if (!c.bcflag[i]) x[i] += theta * ml_prolonged<NF>(wc, par, I, f);
}

}  // namespace gmpnp
