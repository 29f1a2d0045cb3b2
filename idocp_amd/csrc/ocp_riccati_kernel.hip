// S3 / S4 -- backward and forward Riccati sweeps of the contact path.
//
// Replaces RiccatiRecursionSolver::backwardRiccatiRecursion / computeInitialStateDirection /
// forwardRiccatiRecursion for a horizon without discrete events (src/ocp/riccati_recursion_solver.cpp:48-162)
// and the per-stage factorizers they call:
//   BackwardRiccatiRecursionFactorizer::factorizeKKTMatrix / factorizeRiccatiFactorization
//                              (include/idocp/ocp/backward_riccati_recursion_factorizer.hxx:44-161)
//   SplitRiccatiFactorizer::backwardRiccatiRecursion / forwardRiccatiRecursion
//                              (include/idocp/ocp/split_riccati_factorizer.hxx:36-52, 103-128)
// The dynamics are x+ = A x + B u + Fx with A = [Fqq Fqv; Fvq Fvv], B = [0; Fvu]; only the leading
// 6x6 blocks of Fqq / Fqv differ from I / dt I (floating base) and are the only ones stored.
//
// S3: one 256-thread workgroup per OCP instance walks the horizon backwards with
// P_{i+1} and the stage's LQR blocks resident in LDS (~52 kB -> three instances per CU).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "dev_dense.hpp"
#include "dev_lie.hpp"
#include "ocp_device.hpp"
#include "ocp_launch.hpp"

namespace idocp_dev {

template <typename D>
struct RiccatiSmem {
  static constexpr int NV = D::NV, NX = D::NX, NU = D::NU;
  static constexpr int PQQ = 0, PQV = PQQ + NV * NV, PVV = PQV + NV * NV, SQ = PVV + NV * NV, SV = SQ + NV,
                       STAGE = SV + NV + 4,                                  // copy of the kkt record without Qxx
                       STAGE_LEN = OcpLayout<D>::KKT - OcpLayout<D>::K_QXU,
                       ATPQQ = STAGE + STAGE_LEN, ATPQV = ATPQQ + NV * NV, ATPVQ = ATPQV + NV * NV, ATPVV = ATPVQ + NV * NV,
                       BTPQ = ATPVV + NV * NV, BTPV = BTPQ + NU * NV, KM = BTPV + NU * NV,
                       GK = BTPQ,                                            // Quu K reuses the B^T P scratch
                       KV = KM + NU * NX,
                       GW = KV + 16, SQN = GW + NU * NU, SVN = SQN + NV, INVD = SVN + NV + 4, TOTAL = INVD + NU + 4;
  static_assert(2 * NU * NV == NU * NX, "GK aliases B^T P");
  // extra blocks of the HYBRID instantiation (stages that carry a switching constraint: Schur-complement step of
  // SplitRiccatiFactorizer::backwardRiccatiRecursion, split_riccati_factorizer.hxx:43-101)
  // They alias blocks that are dead at that point, so the HYBRID instantiation needs no extra LDS:
  //   A^T P (dead once F, H, G and the k-independent part of the s recursion are done; only ATPQQ / ATPVV are reused
  //   later, as symmetrisation scratch):  DG, SS -> ATPQQ ;  DtM (lives until the write-back) -> ATPQV..ATPVQ ;
  //   SDG, m, Phix^T m -> ATPVV (consumed before the symmetrisation)
  //   B^T P / GK: M (dead once DtM is formed, before GK = Quu K is written)
  // Phix, Phiu, P are read from the swc record (L2) where needed.
  static constexpr int NF = D::NF;
  static constexpr int DG = ATPQQ, SS = DG + NF * NU, DTM = ATPQV, SDG = ATPVV, MV = SDG + NF * NU, SCORR = MV + NF, MMX = BTPQ;
  static_assert(2 * NF * NU <= NV * NV && NU * NX <= 2 * NV * NV && NF * NU + NF + NX <= NV * NV && NF * NX <= 2 * NU * NV, "hybrid aliases");
};

template <typename D, int NT, bool HYBRID>
__global__ __launch_bounds__(NT, 2) void ocp_riccati_backward_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  using S = RiccatiSmem<D>;
  constexpr int NV = D::NV, NX = D::NX, NU = D::NU, NN = NV * NV, NF = D::NF;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int s_ok;
  const OcpProblem* __restrict__ P = B.prob;
  const int M = P->M;
  const OcpNode* __restrict__ nodes = B.nodes;
  const int tid = threadIdx.x;
  constexpr int nt = NT;
  const long b = blockIdx.x;
  const long base = b * P->NS;                    // first record of this instance
  double* Pqq = &sm[S::PQQ];
  double* Pqv = &sm[S::PQV];
  double* Pvv = &sm[S::PVV];
  double* st = &sm[S::STAGE];     // st[k - K_QXU] holds kkt[k] for k >= K_QXU
  constexpr int KO = L::K_QXU, SL = S::STAGE_LEN;
  if (tid == 0) s_ok = 1;
  // terminal stage (riccati_recursion_solver.cpp:53-56)
  {
    const double* __restrict__ kk = B.kkt + (base + nodes[M - 1].slot) * L::KKT;
    double* __restrict__ rr = B.ric + (base + nodes[M - 1].slot) * L::RIC;
    for (int e = tid; e < NN; e += nt) {
      const int c = e / NV, r = e - c * NV;
      const double pqq = kk[L::K_QXX + r + NX * c], pvv = kk[L::K_QXX + (NV + r) + NX * (NV + c)];
      Pqq[e] = pqq; Pqv[e] = 0.0; Pvv[e] = pvv;
      rr[L::R_PQQ + e] = pqq; rr[L::R_PQV + e] = 0.0; rr[L::R_PVV + e] = pvv;
    }
    if (tid < NV) {
      const double sq = -kk[L::K_LX + tid], sv = -kk[L::K_LX + NV + tid];
      sm[S::SQ + tid] = sq; sm[S::SV + tid] = sv;
      rr[L::R_SQ + tid] = sq; rr[L::R_SV + tid] = sv;
    }
  }
  {
    const double* __restrict__ kk = B.kkt + (base + nodes[M - 2].slot) * L::KKT;
    for (int e = tid; e < SL; e += nt) st[e] = kk[KO + e];
  }
  __syncthreads();
  // walk the chain backwards (riccati_recursion_solver.cpp:48-107): impulse stages are ordinary steps with
  // Fqv = 0 (dtq = 0), Fvu = 0, Qxu = 0, Quu = I written by K5b, which makes K = 0, k = 0 and P = F
  for (int i = M - 2; i >= 0; --i) {
    const OcpNode* __restrict__ nd = nodes + i;
    const double dt = nd->dtq;
    const long rec = base + nd->slot;
    const int dimi = HYBRID ? nd->sw_dimi : 0;
    const bool stamp = tid == 0 && b == (gridDim.x > 7 ? 7 : 0) && i == M / 2 && B.prof != nullptr;
#define RSTAMP(k) do { if (stamp) B.prof[16 + k] = wall_clock64(); } while (0)
    RSTAMP(0);
    // software pipeline: the record of stage i was staged into LDS at the end of the previous
    // iteration; issue the global loads of stage i-1 now and park them in registers
    constexpr int PF = (SL + NT - 1) / NT;
    // Phases 1 and 2 below work on 3 x 3 register tiles, one JOB per thread and round (a job = one tile position of
    // one or two output blocks that share an operand):
    //   phase 1 (96 jobs):  0..35 (A^T P)qq|qv   36..71 (A^T P)vq|vv   72..95 (B^T P)q|v
    //   phase 2 (112 jobs): 0..35 F_qq|F_qv      36..71 F_vv           72..95 H_q|H_v     96..111 G
    constexpr int T6 = NV / 3, T4 = NU / 3, J1 = 2 * T6 * T6 + T4 * T6, J2A = T6 * T6, J2B = 2 * T6 * T6, J2C = J2B + T6 * T4,
                  J2D = J2C + T4 * T4, JPT = (J2D + NT - 1) / NT;
    static_assert(NV % 3 == 0 && NU % 3 == 0 && NV >= 6, "3 x 3 tiles aligned with the 6 x 6 base block");
    double pre[PF], qxx[JPT][2][3][3];
    {
      // Qxx of THIS stage is consumed once per element in the F phase: straight to the registers of the job that adds it
      const double* __restrict__ kc = B.kkt + rec * L::KKT;
#pragma unroll
      for (int jj = 0; jj < JPT; ++jj) {
        const int job = tid + NT * jj;
        if (job < J2B) {
          const int t = job < J2A ? job : job - J2A, r0 = 3 * (t % T6), c0 = 3 * (t / T6);
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              if (job < J2A) {
                qxx[jj][0][a][c] = kc[L::K_QXX + (r0 + a) + NX * (c0 + c)];
                qxx[jj][1][a][c] = kc[L::K_QXX + (r0 + a) + NX * (NV + c0 + c)];
              } else {
                qxx[jj][0][a][c] = kc[L::K_QXX + (NV + r0 + a) + NX * (NV + c0 + c)];
              }
            }
        }
      }
    }
    if (i > 0) {
      const double* __restrict__ kn = B.kkt + (base + nodes[i - 1].slot) * L::KKT + KO;
#pragma unroll
      for (int t = 0; t < PF; ++t) { const int e = tid + NT * t; pre[t] = (e < SL) ? kn[e] : 0.0; }
    }
    double* Qxu = st + (L::K_QXU - KO);
    double* Quu = st + (L::K_QUU - KO);
    const double* Fqq6 = st + (L::K_FQQ - KO);
    const double* Fqv6 = st + (L::K_FQV - KO);
    const double* Fvq = st + (L::K_FVQ - KO);
    const double* Fvv = st + (L::K_FVV - KO);
    const double* Fvu = st + (L::K_FVU - KO);
    const double* lx = st + (L::K_LX - KO);
    double* lu = st + (L::K_LU - KO);
    const double* Fx = st + (L::K_FX - KO);
    RSTAMP(1);
    // ---- A^T P blocks and B^T P (backward_riccati_recursion_factorizer.hxx:48-78) ----
    for (int job = tid; job < J1; job += nt) {
      // out1 = X^T Pvq (+ base1), out2 = X^T Pvv (+ base2) on one tile: X = Fvq (qq|qv), Fvv (vq|vv) or Fvu (B^T P)
      const bool bt = job >= 2 * T6 * T6, vhalf = !bt && job >= T6 * T6;
      const int t = bt ? job - 2 * T6 * T6 : (vhalf ? job - T6 * T6 : job);
      const int nr = bt ? T4 : T6, r0 = 3 * (t % nr), c0 = 3 * (t / nr);
      const double* X = (bt ? Fvu : (vhalf ? Fvv : Fvq)) + NV * r0;
      double a1[3][3], a2[3][3];
      if (bt) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) a1[a][c] = a2[a][c] = 0.0;
      } else if (r0 < 6) {
        const double* F6 = (vhalf ? Fqv6 : Fqq6) + 6 * r0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) a1[a][c] = a2[a][c] = 0.0;
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          double f[3], pq[3], pv[3];
#pragma unroll
          for (int a = 0; a < 3; ++a) { f[a] = F6[m + 6 * a]; pq[a] = Pqq[m + NV * (c0 + a)]; pv[a] = Pqv[m + NV * (c0 + a)]; }
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) { a1[a][c] += f[a] * pq[c]; a2[a][c] += f[a] * pv[c]; }
        }
      } else {
        const double sc = vhalf ? dt : 1.0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int e = (r0 + a) + NV * (c0 + c);
            a1[a][c] = vhalf ? sc * Pqq[e] : Pqq[e];
            a2[a][c] = vhalf ? sc * Pqv[e] : Pqv[e];
          }
      }
#pragma unroll 6
      for (int m = 0; m < NV; ++m) {
        double f[3], pvq[3], pvv[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { f[a] = X[m + NV * a]; pvq[a] = Pqv[(c0 + a) + NV * m]; pvv[a] = Pvv[m + NV * (c0 + a)]; }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) { a1[a][c] += f[a] * pvq[c]; a2[a][c] += f[a] * pvv[c]; }
      }
      double* o1 = bt ? &sm[S::BTPQ] : (vhalf ? &sm[S::ATPVQ] : &sm[S::ATPQQ]);
      double* o2 = bt ? &sm[S::BTPV] : (vhalf ? &sm[S::ATPVV] : &sm[S::ATPQV]);
      const int ldo = bt ? NU : NV;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) { o1[(r0 + a) + ldo * (c0 + c)] = a1[a][c]; o2[(r0 + a) + ldo * (c0 + c)] = a2[a][c]; }
    }
    __syncthreads();
    const double* AtPqq = &sm[S::ATPQQ];
    const double* AtPqv = &sm[S::ATPQV];
    const double* AtPvq = &sm[S::ATPVQ];
    const double* AtPvv = &sm[S::ATPVV];
    const double* BtPq = &sm[S::BTPQ];
    const double* BtPv = &sm[S::BTPV];
    RSTAMP(2);
    // ---- F, H, G and the vector term (:79-113); F overwrites P_{i+1}, which is dead from here ----
#pragma unroll
    for (int jj = 0; jj < JPT; ++jj) {
      const int job = tid + NT * jj;
      // Every tile job runs the same instruction stream (no divergence inside a wavefront):
      //   out1 = base1 + X1 Y1, out2 = base2 + X2 Y2 with per-job operand pointers
      //   F_qq | F_qv : X1 = X2 = (A^T P)qv, Y1 = Fvq, Y2 = Fvv ;  F_vv : X2 = (A^T P)vv, Y2 = Fvv (out1 unused)
      //   H_q | H_v   : X1 = (A^T P)qv, X2 = (A^T P)vv, Y1 = Y2 = Fvu ;  G : X1 = (B^T P)v, Y1 = Fvu (out2 unused)
      if (job < J2D) {
        int type, t, nr;
        if (job < J2A) { type = 0; t = job; nr = T6; }
        else if (job < J2B) { type = 1; t = job - J2A; nr = T6; }
        else if (job < J2C) { type = 2; t = job - J2B; nr = T6; }
        else { type = 3; t = job - J2C; nr = T4; }
        const int r0 = 3 * (t % nr), c0 = 3 * (t / nr);
        const double* X1 = (type == 3 ? BtPv : AtPqv) + r0;
        const double* X2 = (type == 0 ? AtPqv : AtPvv) + (type == 3 ? 0 : r0);
        const int ldx1 = type == 3 ? NU : NV;
        const double* Y1 = (type == 0 || type == 1 ? Fvq : Fvu) + NV * c0;
        const double* Y2 = (type == 0 || type == 1 ? Fvv : Fvu) + NV * c0;
        double a1[3][3], a2[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) a1[a][c] = a2[a][c] = 0.0;
        if (type <= 1) {
          const double* Ab = (type == 1 ? AtPvq : AtPqq) + r0;          // multiplies the 6 x 6 / identity part of A
          if (c0 < 6) {
#pragma unroll
            for (int m = 0; m < 6; ++m) {
              double x[3], fq[3], fv[3];
#pragma unroll
              for (int a = 0; a < 3; ++a) { x[a] = Ab[a + NV * m]; fq[a] = Fqq6[m + 6 * (c0 + a)]; fv[a] = Fqv6[m + 6 * (c0 + a)]; }
#pragma unroll
              for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) { a1[a][c] += x[a] * fq[c]; a2[a][c] += x[a] * fv[c]; }
            }
          } else {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
              for (int c = 0; c < 3; ++c) { const double x = Ab[a + NV * (c0 + c)]; a1[a][c] = x; a2[a][c] = dt * x; }
          }
        }
#pragma unroll 6
        for (int m = 0; m < NV; ++m) {
          double x1[3], x2[3], y1[3], y2[3];
#pragma unroll
          for (int a = 0; a < 3; ++a) { x1[a] = X1[a + ldx1 * m]; x2[a] = X2[a + NV * m]; y1[a] = Y1[m + NV * a]; y2[a] = Y2[m + NV * a]; }
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) { a1[a][c] += x1[a] * y1[c]; a2[a][c] += x2[a] * y2[c]; }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int e = (r0 + a) + NV * (c0 + c);
            if (type == 0) { Pqq[e] = qxx[jj][0][a][c] + a1[a][c]; Pqv[e] = qxx[jj][1][a][c] + a2[a][c]; }
            else if (type == 1) Pvv[e] = qxx[jj][0][a][c] + a2[a][c];
            else if (type == 2) { Qxu[(r0 + a) + NX * (c0 + c)] += a1[a][c]; Qxu[(NV + r0 + a) + NX * (c0 + c)] += a2[a][c]; }
            else Quu[(r0 + a) + NU * (c0 + c)] += a1[a][c];
          }
      }
    }
    // The two vector terms run behind the tile jobs of different wavefronts (NT >= 128): lu on the last NU threads,
    // the k-independent part of the s recursion (:141-160) on the first 2 NV.  Neither reads anything the tile jobs write
    // (P_{i+1} is only read through A^T P here).
    RSTAMP(11);
    if (tid >= NT - NU) {
      const int j = tid - (NT - NU);
      double acc = 0.0;
      for (int c = 0; c < NV; ++c) acc += BtPq[j + NU * c] * Fx[c] + BtPv[j + NU * c] * Fx[NV + c];
      for (int m = 0; m < NV; ++m) acc -= Fvu[m + NV * j] * sm[S::SV + m];
      lu[j] += acc;
    }
    if (tid < 2 * NV) {
      const bool isv = tid >= NV;
      const int r = isv ? tid - NV : tid;
      const double* F6 = isv ? Fqv6 : Fqq6;
      const double* Fm = isv ? Fvv : Fvq;
      const double* A1 = isv ? AtPvq : AtPqq;
      const double* A2 = isv ? AtPvv : AtPqv;
      double acc;
      if (r < 6) {
        acc = 0.0;
        for (int m = 0; m < 6; ++m) acc += F6[m + 6 * r] * sm[S::SQ + m];
      } else {
        acc = isv ? dt * sm[S::SQ + r] : sm[S::SQ + r];
      }
      for (int m = 0; m < NV; ++m) acc += Fm[m + NV * r] * sm[S::SV + m];
      for (int c = 0; c < NV; ++c) acc -= A1[r + NV * c] * Fx[c] + A2[r + NV * c] * Fx[NV + c];
      acc -= lx[isv ? NV + r : r];
      sm[(isv ? S::SVN : S::SQN) + r] = acc;
    }
    RSTAMP(12);
    __syncthreads();
    RSTAMP(3);
    // Qvq = Qqv^T (:94) -- only read through Qqv below, kept for completeness of the record
    // ---- LLT(Quu), K = -Quu^-1 Qxu^T, k = -Quu^-1 lu (split_riccati_factorizer.hxx:43-46) ----
    // (the reference factorises with Eigen::LLT; here Quu^-1 is formed by Gauss-Jordan on one
    // wavefront and applied with two small products -- same K, k up to rounding)
    RSTAMP(9);
    // Quu = L L^T and the solves K = -Quu^-1 Qxu^T, k = -Quu^-1 lu in the registers of one wavefront, one right-hand side per lane
    // (Eigen::LLT compute + solve, split_riccati_factorizer.hxx:43-46).  Round 1 multiplied with an explicit Gauss-Jordan
    // inverse: on the stage behind a switching constraint G = Quu + B^T P B has a condition number of 1e8 and P = F - K^T G K came
    // out 6e-9 off (5e-12 with the solves; long double referee, tests/test_hybrid_gpu.py).
    const bool constrained = HYBRID && dimi > 0;
    if (tid < 64) {
      double x[NU];
      if (!constrained) {
#pragma unroll
        for (int m = 0; m < NU; ++m) x[m] = (tid < NX) ? Qxu[tid + NX * m] : lu[m];
      } else {
#pragma unroll
        for (int m = 0; m < NU; ++m) x[m] = (m == tid) ? 1.0 : 0.0;      // Ginv = llt.solve(I) (:60): the Schur-complement step works with the explicit inverse, like the reference
      }
      choleskySolveRows<NU>(Quu, NU, tid, &s_ok, x);
      if (!constrained) {
        if (tid <= NX) {
#pragma unroll
          for (int m = 0; m < NU; ++m) { if (tid < NX) sm[S::KM + m + NU * tid] = -x[m]; else sm[S::KV + m] = -x[m]; }
        }
      } else if (tid < NU) {
#pragma unroll
        for (int m = 0; m < NU; ++m) sm[S::GW + m + NU * tid] = x[m];
      }
    }
    __syncthreads();
    RSTAMP(10);
    if (HYBRID && dimi > 0) {
      // ---- Schur complement w.r.t. the switching constraint Phix dx + Phiu du + P = 0 (split_riccati_factorizer.hxx:56-70) ----
      const double* __restrict__ W = B.swc + rec * L::SWC;
      const double* __restrict__ Phiu = W + L::W_PHIU;
      for (int e = tid; e < dimi * NU; e += nt) {                // DGinv = Phiu Ginv
        const int c = e / dimi, j = e - c * dimi;
        double acc = 0.0;
        for (int m = 0; m < NU; ++m) acc += Phiu[j + NF * m] * sm[S::GW + m + NU * c];
        sm[S::DG + j + NF * c] = acc;
      }
      __syncthreads();
      for (int e = tid; e < dimi * dimi; e += nt) {              // S = DGinv Phiu^T
        const int c = e / dimi, j = e - c * dimi;
        double acc = 0.0;
        for (int m = 0; m < NU; ++m) acc += sm[S::DG + j + NF * m] * Phiu[c + NF * m];
        sm[S::SS + j + NF * c] = acc;
      }
      __syncthreads();
      // S = L L^T and S^-1 [DGinv, Phix, P] by triangular solves like the reference's llt_s_.solve (:64-66, 71-74): Cholesky and the
      // 49 right-hand sides in the registers of one wavefront (lane = column).  Round 1 formed S^-1 by unpivoted Gauss-Jordan and
      // multiplied: on a stage a few milliseconds in front of a touch-down S is ill-conditioned and the explicit inverse added its
      // share to a direction 1e-6 off (now 1e-10, as far as the FP64 oracle itself is from the long double build, tests/test_hybrid_gpu.py).
      if (tid < 64) {
        static_assert(NU + NX + 1 <= 64, "one lane per right-hand side");
        const double* __restrict__ Wc = B.swc + rec * L::SWC;
        double x[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          double val = 0.0;
          if (j < dimi) val = tid < NU ? sm[S::DG + j + NF * tid] : (tid < NU + NX ? Wc[L::W_PHIX + j + NF * (tid - NU)] : (tid == NU + NX ? Wc[L::W_P + j] : 0.0));
          x[j] = val;
        }
        choleskySolveRows<NF>(&sm[S::SS], NF, tid, &s_ok, x, dimi);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          if (j < dimi) {
            if (tid < NU) sm[S::SDG + j + NF * tid] = x[j];
            else if (tid < NU + NX) sm[S::MMX + j + NF * (tid - NU)] = x[j];      // S^-1 Phix; - SinvDGinv Qxu^T follows below
            else if (tid == NU + NX) sm[S::MV + j] = x[j];
          }
        }
      }
      __syncthreads();
      for (int e = tid; e < NU * NU; e += nt) {                  // Ginv -= SinvDGinv^T DGinv
        const int c = e / NU, r = e - c * NU;
        double acc = 0.0;
        for (int l = 0; l < dimi; ++l) acc += sm[S::SDG + l + NF * r] * sm[S::DG + l + NF * c];
        sm[S::GW + e] -= acc;
      }
      __syncthreads();
    }
    if (constrained) {
      // K = -Ginv Qxu^T - SinvDGinv^T Phix, k = -Ginv lu - SinvDGinv^T P with the updated Ginv (:67-70)
      const double* __restrict__ Phix = B.swc + rec * L::SWC + L::W_PHIX;
      const double* __restrict__ Pv = B.swc + rec * L::SWC + L::W_P;
      for (int e = tid; e < NU * NX; e += nt) {
        const int c = e / NU, j = e - c * NU;
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < NU; ++m) acc += sm[S::GW + j + NU * m] * Qxu[c + NX * m];
        for (int l = 0; l < dimi; ++l) acc += sm[S::SDG + l + NF * j] * Phix[l + NF * c];
        sm[S::KM + e] = -acc;
      }
      if (tid >= NT - NU) {
        const int j = tid - (NT - NU);
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < NU; ++m) acc += sm[S::GW + j + NU * m] * lu[m];
        for (int l = 0; l < dimi; ++l) acc += sm[S::SDG + l + NF * j] * Pv[l];
        sm[S::KV + j] = -acc;
      }
    }
    if (HYBRID && dimi > 0) {
      // multiplier policy dxi = M dx + m (:71-74): M = S^-1 Phix - SinvDGinv Qxu^T, m = S^-1 P - SinvDGinv lu (S^-1 Phix, S^-1 P: see above)
      double* __restrict__ W = B.swc + rec * L::SWC;
      for (int e = tid; e < dimi * NX; e += nt) {
        const int c = e / dimi, l = e - c * dimi;
        double acc = sm[S::MMX + l + NF * c];
        for (int m = 0; m < NU; ++m) acc -= sm[S::SDG + l + NF * m] * Qxu[c + NX * m];
        sm[S::MMX + l + NF * c] = acc;
        W[L::W_M + l + NF * c] = acc;
      }
      if (tid < dimi) {
        const int l = tid;
        double acc = sm[S::MV + l];
        for (int m = 0; m < NU; ++m) acc -= sm[S::SDG + l + NF * m] * lu[m];
        sm[S::MV + l] = acc;
        W[L::W_m + l] = acc;
      }
    }
    __syncthreads();
    if (HYBRID && dimi > 0) {
      const double* __restrict__ W = B.swc + rec * L::SWC;
      for (int e = tid; e < NU * NX; e += nt) {                  // DtM = Phiu^T M (:88)
        const int c = e / NU, m = e - c * NU;
        double acc = 0.0;
        for (int l = 0; l < dimi; ++l) acc += W[L::W_PHIU + l + NF * m] * sm[S::MMX + l + NF * c];
        sm[S::DTM + m + NU * c] = acc;
      }
      if (tid < NX) {                                            // Phix^T m (:98-99), consumed at the write-back
        double acc = 0.0;
        for (int l = 0; l < dimi; ++l) acc += W[L::W_PHIX + l + NF * tid] * sm[S::MV + l];
        sm[S::SCORR + tid] = acc;
      }
      __syncthreads();                                           // M (aliasing GK) is dead from here
    }
    RSTAMP(4);
    // GK = Quu K (backward_riccati_recursion_factorizer.hxx:128)
    mm(colMajor(&sm[S::GK], NU), colMajor(Quu, NU), colMajor(&sm[S::KM], NU), NU, NX, NU, 1.0, false, tid, nt);
    // s recursion, part 2: - Qxu k
    if (tid < NV) {
      const int r = tid;
      double sq = sm[S::SQN + r], sv = sm[S::SVN + r];
      for (int j = 0; j < NU; ++j) { sq -= Qxu[r + NX * j] * sm[S::KV + j]; sv -= Qxu[(NV + r) + NX * j] * sm[S::KV + j]; }
      sm[S::SQN + r] = sq; sm[S::SVN + r] = sv;
    }
    __syncthreads();
    RSTAMP(5);
    // P = F - K^T G K (:122-131)
    for (int e = tid; e < NN; e += nt) {
      const int c = e / NV, r = e - c * NV;
      double a = 0.0, b2 = 0.0, d2 = 0.0;
      for (int j = 0; j < NU; ++j) {
        const double kq = sm[S::KM + j + NU * r], kv = sm[S::KM + j + NU * (NV + r)];
        a += kq * sm[S::GK + j + NU * c];
        b2 += kq * sm[S::GK + j + NU * (NV + c)];
        d2 += kv * sm[S::GK + j + NU * (NV + c)];
      }
      Pqq[e] -= a;
      Pqv[e] -= b2;
      Pvv[e] -= d2;
    }
    __syncthreads();
    RSTAMP(6);
    double sc_q = 0.0, sc_v = 0.0;                    // Phix^T m of this thread's row, read before ATPVV becomes scratch
    if (HYBRID && dimi > 0 && tid < NV) { sc_q = sm[S::SCORR + tid]; sc_v = sm[S::SCORR + NV + tid]; }
    __syncthreads();
    // preserve the symmetry (:133-135) -- symmetrised values staged in the A^T P scratch
    for (int e = tid; e < NN; e += nt) {
      const int c = e / NV, r = e - c * NV;
      sm[S::ATPQQ + e] = 0.5 * (Pqq[e] + Pqq[c + NV * r]);
      sm[S::ATPVV + e] = 0.5 * (Pvv[e] + Pvv[c + NV * r]);
    }
    __syncthreads();
    RSTAMP(7);
    double* __restrict__ rr = B.ric + rec * L::RIC;
    double* __restrict__ gg = B.gain + rec * L::GAIN;
    for (int e = tid; e < NN; e += nt) {
      double pqq = sm[S::ATPQQ + e], pqv = Pqv[e], pvv = sm[S::ATPVV + e];
      if (HYBRID && dimi > 0) {
        // P -= K^T D^T M + (K^T D^T M)^T, block by block (split_riccati_factorizer.hxx:88-97)
        const int c = e / NV, r = e - c * NV;
        double aqq = 0.0, aqv = 0.0, avv = 0.0;
        for (int j = 0; j < NU; ++j) {
          const double kqr = sm[S::KM + j + NU * r], kqc = sm[S::KM + j + NU * c], kvr = sm[S::KM + j + NU * (NV + r)], kvc = sm[S::KM + j + NU * (NV + c)];
          const double dqr = sm[S::DTM + j + NU * r], dqc = sm[S::DTM + j + NU * c], dvr = sm[S::DTM + j + NU * (NV + r)], dvc = sm[S::DTM + j + NU * (NV + c)];
          aqq += kqr * dqc + kqc * dqr;
          aqv += kqr * dvc + kvc * dqr;
          avv += kvr * dvc + kvc * dvr;
        }
        pqq -= aqq; pqv -= aqv; pvv -= avv;
      }
      Pqq[e] = pqq; Pqv[e] = pqv; Pvv[e] = pvv;
      rr[L::R_PQQ + e] = pqq; rr[L::R_PQV + e] = pqv; rr[L::R_PVV + e] = pvv;
    }
    if (tid < NV) {
      double sq = sm[S::SQN + tid], sv = sm[S::SVN + tid];
      if (HYBRID && dimi > 0) { sq -= sc_q; sv -= sc_v; }     // (:98-99)
      sm[S::SQ + tid] = sq; sm[S::SV + tid] = sv;
      rr[L::R_SQ + tid] = sq; rr[L::R_SV + tid] = sv;
    }
    for (int e = tid; e < NU * NX; e += nt) gg[L::G_K + e] = sm[S::KM + e];
    if (tid < NU) gg[L::G_k + tid] = sm[S::KV + tid];
    __syncthreads();
    if (i > 0) {
#pragma unroll
      for (int t = 0; t < PF; ++t) { const int e = tid + NT * t; if (e < SL) st[e] = pre[t]; }
    }
    __syncthreads();
    RSTAMP(8);
#undef RSTAMP
  }
  if (tid == 0 && !s_ok && B.status[b] == 0) B.status[b] = 1;
}

// S4: forward sweep, one wavefront per instance.
template <typename D>
__global__ __launch_bounds__(64) void ocp_riccati_forward_kernel(OcpBuffers B, const double* __restrict__ q0,
                                                                const double* __restrict__ v0) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NQ = D::NQ, NX = D::NX, NU = D::NU;
  __shared__ double dx[NX], du[NU], dxn[NX];
  const OcpProblem* __restrict__ P = B.prob;
  const int M = P->M;
  const OcpNode* __restrict__ nodes = B.nodes;
  const int lane = threadIdx.x;
  const long b = blockIdx.x;
  const long base = b * P->NS;
  const double* __restrict__ s0 = B.sol + (base + nodes[0].slot) * L::SOL;
  // RiccatiRecursionSolver::computeInitialStateDirection (riccati_recursion_solver.cpp:110-126)
  if (lane == 0) {
    double R[9], p[3], d6[6];
    lieRelative(s0 + L::S_Q, q0 + b * NQ, R, p);         // q (-) s[0].q
    lieLog6(R, p, d6);
    const double* __restrict__ Fi = B.exp + (base + nodes[0].slot) * L::EXP + L::E_FQQPI;
    for (int r = 0; r < 6; ++r) { double acc = 0.0; for (int m = 0; m < 6; ++m) acc += Fi[r + 6 * m] * d6[m]; dx[r] = -acc; }
  }
  if (lane >= 6 && lane < NV) dx[lane] = q0[b * NQ + lane + 1] - s0[L::S_Q + lane + 1];
  if (lane < NV) dx[NV + lane] = v0[b * NV + lane] - s0[L::S_V + lane];
  __syncthreads();
  for (int i = 0; i < M - 1; ++i) {                 // forwardRiccatiRecursion along the chain (riccati_recursion_solver.cpp:129-162)
    const long rec = base + nodes[i].slot;
    const double dt = nodes[i].dtq;
    const double* __restrict__ gg = B.gain + rec * L::GAIN;
    const double* __restrict__ kk = B.kkt + rec * L::KKT;
    double* __restrict__ dd = B.dir + rec * L::DIR;
    if (lane < NU) {
      double acc = gg[L::G_k + lane];
      for (int c = 0; c < NX; ++c) acc += gg[L::G_K + lane + NU * c] * dx[c];
      du[lane] = acc;
      dd[L::D_U + lane] = acc;
    }
    if (lane < NV) { dd[L::D_Q + lane] = dx[lane]; dd[L::D_V + lane] = dx[NV + lane]; }
    __syncthreads();
    if (lane < NV) {
      const int r = lane;
      double dq = kk[L::K_FX + r], dv = kk[L::K_FX + NV + r];
      if (r < 6) {
        for (int m = 0; m < 6; ++m) dq += kk[L::K_FQQ + r + 6 * m] * dx[m] + kk[L::K_FQV + r + 6 * m] * dx[NV + m];
      } else {
        dq += dx[r] + dt * dx[NV + r];
      }
      for (int c = 0; c < NV; ++c) dv += kk[L::K_FVQ + r + NV * c] * dx[c] + kk[L::K_FVV + r + NV * c] * dx[NV + c];
      for (int j = 0; j < NU; ++j) dv += kk[L::K_FVU + r + NV * j] * du[j];
      dxn[r] = dq; dxn[NV + r] = dv;
    }
    __syncthreads();
    if (lane < NX) dx[lane] = dxn[lane];
    __syncthreads();
  }
  double* __restrict__ dd = B.dir + (base + nodes[M - 1].slot) * L::DIR;
  if (lane < NV) { dd[L::D_Q + lane] = dx[lane]; dd[L::D_V + lane] = dx[NV + lane]; }
}

template <typename D>
void OcpLaunch<D>::riccatiBackward(const OcpBuffers& B, long batch, int M, bool hybrid, hipStream_t st) {
  (void)M;
  const size_t smem = RiccatiSmem<D>::TOTAL * sizeof(double), smem_h = smem;      // the Schur blocks alias dead ones
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)ocp_riccati_backward_kernel<D, 256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_riccati_backward_kernel<D, 128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_riccati_backward_kernel<D, 64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_riccati_backward_kernel<D, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_h);
    configured = true;
  }
  // A chain with switching constraints takes the HYBRID instantiation (same LDS footprint).
  if (hybrid) {
    hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 128, true>), dim3((unsigned)batch), dim3(128), smem_h, st, B);
    return;
  }
  // Throughput mode (many instances): two wavefronts per instance, four instances per CU (39.8 kB
  // of LDS each) -- measured best at batch 1024 (2.48 ms vs 3.22 ms with one and 2.96 ms with four
  // wavefronts).  Latency mode (few instances): four wavefronts per instance.
  // IDOCP_RICCATI_NT={64,128,256} overrides the choice (tuning aid).
  static const int nt_env = getenv("IDOCP_RICCATI_NT") ? atoi(getenv("IDOCP_RICCATI_NT")) : 0;
  if (nt_env == 128) hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 128, false>), dim3((unsigned)batch), dim3(128), smem, st, B);
  else if (nt_env == 256) hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 256, false>), dim3((unsigned)batch), dim3(256), smem, st, B);
  else if (nt_env == 64) hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 64, false>), dim3((unsigned)batch), dim3(64), smem, st, B);
  else if (batch >= 512) hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 128, false>), dim3((unsigned)batch), dim3(128), smem, st, B);
  else hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 256, false>), dim3((unsigned)batch), dim3(256), smem, st, B);
}
template <typename D>
void OcpLaunch<D>::riccatiForward(const OcpBuffers& B, long batch, int M, const double* q0, const double* v0, hipStream_t st) {
  (void)M;
  hipLaunchKernelGGL((ocp_riccati_forward_kernel<D>), dim3((unsigned)batch), dim3(64), 0, st, B, q0, v0);
}

template void OcpLaunch<LeggedDims<4, 3>>::riccatiBackward(const OcpBuffers&, long, int, bool, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::riccatiForward(const OcpBuffers&, long, int, const double*, const double*, hipStream_t);

}  // namespace idocp_dev
