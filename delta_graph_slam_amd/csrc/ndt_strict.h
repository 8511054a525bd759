// ndt_strict_order = 1 (UPSTREAM): computeDerivatives / updateDerivatives in upstream's own operation order, as ONE launch per
// evaluation (derivatives + the pair's optimiser step in its closing workgroup).  Included by ndt_align.hip below the optimiser.
//
// What the mode promises (include/dgs_reg.h): every FLOAT operation of upstream's per-voxel update -- the CPU
// checker states the same sequence -- individually rounded, in upstream's order; the float increments are
// converted and added to DOUBLE sums.  Upstream adds them to a per-point total first and sums the points' totals in index order; here
// every increment goes straight into the thread's running double sum, the threads' sums are added in a fixed order per workgroup, the
// workgroups' rows in slice order.  A sum of <= 27 float increments is exact in double unless their exponents differ by more than
// 29 bits, so the only difference to a CPU run is the association of the double additions (~1e-14 relative per evaluation) --
// ndt_strict_order = 2 removes that too.
//
// Round 4 rewrite of round 2's validation kernel (which kept a point's 43 double totals AND the thread's 43 double totals in
// registers -- 256 VGPRs, spills, 98 us per launch -- and left the optimiser to a second launch with a serial SVD):
//   * voxel records as upstream's floats (VoxelStrictRec: the nine float(icov) entries are made once per target, not per visit);
//   * one set of double accumulators per thread, a point's voxels visited through a per-lane list of its VALID slots (a wave runs as
//     many rounds as its fullest point has voxels, not as many as there are slots with any taker);
//   * score + gradient evaluations (More-Thuente trials) skip the Hessian code altogether (wave-uniform branch);
//   * the closing workgroup (ticket hand-off as in the default order's fused kernel, common.h) sums the rows and advances the optimiser,
//     with Eigen's two-sided JacobiSVD laid out across the wave (solve6.h);
//   * evaluation kind 2: PCL's double-precision computeHessian / updateHessian pass (dgs_params.ndt_hessian_recompute_double).
// (no namespace of its own: included inside namespace dgs)

// std::exp(double) as glibc >= 2.28 computes it on an FMA-capable x86-64 (sysdeps/ieee754/dbl-64/e_exp.c: N = 128, a 2 x 128 table of scale / tail, a
// degree-5 polynomial, scale + scale * tmp, the -mfma build's contractions spelled out): the exponential of PCL's updateHessian.  The CPU checker
// carries the same sequence and compares it with its libm (0 differences on 4e8 arguments).  `tab` = kGlibcExpTab.
__device__ __forceinline__ double glibc_exp_dev(double x, const unsigned long long* __restrict__ tab) {
#pragma clang fp contract(off)
  constexpr double N = 128.0;
  constexpr double kInvLn2N = 0x1.71547652b82fep0 * N, kNegLn2hiN = -0x1.62e42fefa0000p-8, kNegLn2loN = -0x1.cf79abc9e3b3ap-47, kShift = 0x1.8p52;
  constexpr double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
  const unsigned long long xb = (unsigned long long)__double_as_longlong(x);
  const unsigned abstop = (unsigned)(xb >> 52) & 0x7ffu;
  bool special = false;
  if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {                 // |x| < 2^-54 or |x| >= 512
    if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;
    if (abstop >= 0x409u) {                                 // |x| >= 1024
      if (xb == 0xfff0000000000000ull) return 0.0;
      if (abstop >= 0x7ffu) return 1.0 + x;
      return (xb >> 63) ? 0.0 : __builtin_inf();
    }
    special = true;
  }
  double kd = __builtin_fma(kInvLn2N, x, kShift);
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd -= kShift;
  const double r = __builtin_fma(kd, kNegLn2loN, __builtin_fma(kd, kNegLn2hiN, x));
  const unsigned idx = 2u * (unsigned)(ki & 127ull);
  const double tail = __longlong_as_double((long long)tab[idx]);
  unsigned long long sbits = tab[idx + 1] + (ki << 45);
  const double r2 = r * r;
  double tmp = __builtin_fma(r2, __builtin_fma(r, C3, C2), tail + r);
  tmp = __builtin_fma(r2 * r2, __builtin_fma(r, C5, C4), tmp);
  if (special) {
    if ((ki & 0x80000000ull) == 0) {
      sbits -= 1009ull << 52;
      const double scale = __longlong_as_double((long long)sbits);
      return 0x1p1009 * __builtin_fma(scale, tmp, scale);
    }
    sbits += 1022ull << 52;
    const double scale = __longlong_as_double((long long)sbits);
    const double st = scale * tmp;
    double y = scale + st;
    if (y < 1.0) {
      double lo = scale - y + st;
      const double hi = 1.0 + y;
      lo = 1.0 - hi + y + lo;
      y = (hi + lo) - 1.0;
      if (y == 0.0) y = 0.0;
    }
    return 0x1p-1022 * y;
  }
  const double scale = __longlong_as_double((long long)sbits);
  return __builtin_fma(scale, tmp, scale);
}

// exp(double) as the CPU checker states it: the fixed sequence of det_expf without the rounding to float, the
// scaling by 2^k split in two so that subnormal results round once.
__device__ __forceinline__ double det_exp(double x) {
#pragma clang fp contract(off)
  if (x != x) return x;
  if (x < -746.0) return 0.0;
  if (x > 710.0) return __builtin_inf();
  const double kd = floor(x * 1.4426950408889634 + 0.5);
  const double r = (x - kd * 0x1.62e42fefa38p-1) - kd * 0x1.ef35793c7673p-45;
  double p = 1.0 / 6227020800.0;
  p = p * r + 1.0 / 479001600.0;
  p = p * r + 1.0 / 39916800.0;
  p = p * r + 1.0 / 3628800.0;
  p = p * r + 1.0 / 362880.0;
  p = p * r + 1.0 / 40320.0;
  p = p * r + 1.0 / 5040.0;
  p = p * r + 1.0 / 720.0;
  p = p * r + 1.0 / 120.0;
  p = p * r + 1.0 / 24.0;
  p = p * r + 1.0 / 6.0;
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  const long long k = (long long)kd;
  const long long k1 = k / 2, k2 = k - k1;
  const double s1 = __longlong_as_double((k1 + 1023) << 52), s2 = __longlong_as_double((k2 + 1023) << 52);
  return (p * s1) * s2;
}

// Valid neighbour voxels of a transformed point, in upstream's visiting order: bit k of the result = slot k holds a voxel (vids[k]).
template <int SEARCH>
__device__ __forceinline__ unsigned strict_neighbourhood(const float (&xt)[3], const VoxelGrid& g, const int leaf_pow2, int (&vids)[Offsets<SEARCH>::N]) {
  constexpr int NB = Offsets<SEARCH>::N;
  const int c0 = (int)floorf(leaf_pow2 ? xt[0] * g.inv_leaf : xt[0] / g.leaf);
  const int c1 = (int)floorf(leaf_pow2 ? xt[1] * g.inv_leaf : xt[1] / g.leaf);
  const int c2 = (int)floorf(leaf_pow2 ? xt[2] * g.inv_leaf : xt[2] / g.leaf);
  unsigned mask = 0;
  // interior cells (every neighbour inside the grid) need no per-neighbour bounds test: base pointer + fixed offsets
  const bool interior = c0 > g.min_b[0] && c0 < g.max_b[0] && c1 > g.min_b[1] && c1 < g.max_b[1] && c2 > g.min_b[2] && c2 < g.max_b[2];
  if (interior) {
    const int* __restrict__ base = g.cell2vox + ((c0 - g.min_b[0]) + (c1 - g.min_b[1]) * g.mul1 + (c2 - g.min_b[2]) * g.mul2);
#pragma unroll
    for (int k = 0; k < NB; k++) {
      int dx, dy, dz;
      neighbour_offset<SEARCH>(k, dx, dy, dz);
      vids[k] = base[dx + dy * g.mul1 + dz * g.mul2];
    }
  } else {
#pragma unroll
    for (int k = 0; k < NB; k++) {
      int dx, dy, dz;
      neighbour_offset<SEARCH>(k, dx, dy, dz);
      const int a0 = c0 + dx, a1 = c1 + dy, a2 = c2 + dz;
      const bool inb = a0 >= g.min_b[0] && a0 <= g.max_b[0] && a1 >= g.min_b[1] && a1 <= g.max_b[1] && a2 >= g.min_b[2] && a2 <= g.max_b[2];
      vids[k] = inb ? g.cell2vox[(a0 - g.min_b[0]) + (a1 - g.min_b[1]) * g.mul1 + (a2 - g.min_b[2]) * g.mul2] : -1;
    }
  }
  if (SEARCH == DGS_NDT_KDTREE) {
    const float r2 = g.leaf * g.leaf;
#pragma unroll
    for (int k = 0; k < NB; k++) {
      if (vids[k] < 0) continue;
      const float4 ce = g.centroid[vids[k]];
      const float ex = ce.x - xt[0], ey = ce.y - xt[1], ez = ce.z - xt[2];
      if (!(ex * ex + ey * ey + ez * ez < r2)) vids[k] = -1;
    }
  }
#pragma unroll
  for (int k = 0; k < NB; k++) mask |= (vids[k] >= 0) ? (1u << k) : 0u;
  return mask;
}

template <int NB>
__device__ __forceinline__ int strict_pick(const int (&vids)[NB], const int k) {   // vids stays in registers: a chain of selects, not scratch
  int vid = vids[0];
#pragma unroll
  for (int j = 1; j < NB; j++) vid = (k == j) ? vids[j] : vid;
  return vid;
}

// ---- kinds 0 / 1: updateDerivatives in float for ONE (point, voxel) item; acc = [score, g 0..5, H 0..35 row-major].
// xt: the transformed point; xj / xh: the point's 8 + 15 products with the angle tables (computePointDerivatives).
// Two halves, so that the item-compacted kernel can issue the NEXT item's voxel record between them (the Hessian block behind covers its L2 latency):
// front = Mahalanobis term, exp, the weight test, gradient increments, the x^T C H vectors; back = the 36 Hessian increments and the score.
struct StrictMid {
  float C[3][3], cPG[3][3], g6[6], xch[6], e, score_inc;
};

// the 64-byte voxel record as four 16-byte words: mean (3 doubles), then the 9 floats of float(icov) row-major, one pad
struct StrictRecWords {
  float4 a, b, c, d;
};
__device__ __forceinline__ StrictRecWords strict_load_rec(const VoxelStrictRec* __restrict__ rec) {
  const float4* __restrict__ r4 = reinterpret_cast<const float4*>(rec);
  return StrictRecWords{r4[0], r4[1], r4[2], r4[3]};
}

template <bool NEED_H, bool GLIBC_ONLY = false>
__device__ __forceinline__ bool strict_item_front(const float (&xt)[3], const float (&xj)[8], const float (&xh)[15], const StrictRecWords& w,
                                                  const double gauss_d1, const float gd2, double (&acc)[kStrictAccum], StrictMid& m,
                                                  const unsigned long long* __restrict__ exptab) {
  const float pg13 = xj[0], pg23 = xj[1];
  const float pg4[3] = {xj[2], xj[3], xj[4]}, pg5[3] = {xj[5], xj[6], xj[7]};
  const float4 ra = w.a, rb = w.b, rc = w.c, rd = w.d;
  const double m0 = __hiloint2double(__float_as_int(ra.y), __float_as_int(ra.x)), m1 = __hiloint2double(__float_as_int(ra.w), __float_as_int(ra.z)),
               m2 = __hiloint2double(__float_as_int(rb.y), __float_as_int(rb.x));
  const float q0 = (float)((double)xt[0] - m0), q1 = (float)((double)xt[1] - m1), q2 = (float)((double)xt[2] - m2);
  m.C[0][0] = rb.z; m.C[0][1] = rb.w; m.C[0][2] = rc.x;
  m.C[1][0] = rc.y; m.C[1][1] = rc.z; m.C[1][2] = rc.w;
  m.C[2][0] = rd.x; m.C[2][1] = rd.y; m.C[2][2] = rd.z;
  float qC[3];
#pragma unroll
  for (int c = 0; c < 3; c++) qC[c] = q0 * m.C[0][c] + q1 * m.C[1][c] + q2 * m.C[2][c];
  const float e_arg = -gd2 * (q0 * qC[0] + q1 * qC[1] + q2 * qC[2]) * 0.5f;
  float e = (GLIBC_ONLY || exptab) ? glibc_expf_dev(e_arg, exptab) : det_expf(e_arg);   // std::exp(float) as glibc computes it / rounds 1-3's polynomial (NdtConsts::exp_libm)
  m.score_inc = (float)(-gauss_d1 * (double)e);
  e = gd2 * e;
  if (e > 1 || e < 0 || e != e) return false;
  m.e = (float)((double)e * gauss_d1);
  // C * point gradient: columns 0..2 are C itself (unit columns), column 3 has a zero first factor
#pragma unroll
  for (int r = 0; r < 3; r++) {
    m.cPG[r][0] = m.C[r][1] * pg13 + m.C[r][2] * pg23;
    m.cPG[r][1] = m.C[r][0] * pg4[0] + m.C[r][1] * pg4[1] + m.C[r][2] * pg4[2];
    m.cPG[r][2] = m.C[r][0] * pg5[0] + m.C[r][1] * pg5[1] + m.C[r][2] * pg5[2];
  }
  m.g6[0] = qC[0]; m.g6[1] = qC[1]; m.g6[2] = qC[2];   // q^T (C * unit column) is q^T C: the same operations
#pragma unroll
  for (int c = 0; c < 3; c++) m.g6[3 + c] = q0 * m.cPG[0][c] + q1 * m.cPG[1][c] + q2 * m.cPG[2][c];
#pragma unroll
  for (int c = 0; c < 6; c++) acc[1 + c] += (double)(m.e * m.g6[c]);
  if (NEED_H) {
    // x^T C H for the six distinct vectors: a = (0, xh0, xh1) b = (0, xh2, xh3) c = (0, xh4, xh5) d = xh6..8 e = xh9..11 f = xh12..14
    m.xch[0] = qC[1] * xh[0] + qC[2] * xh[1];
    m.xch[1] = qC[1] * xh[2] + qC[2] * xh[3];
    m.xch[2] = qC[1] * xh[4] + qC[2] * xh[5];
    m.xch[3] = qC[0] * xh[6] + qC[1] * xh[7] + qC[2] * xh[8];
    m.xch[4] = qC[0] * xh[9] + qC[1] * xh[10] + qC[2] * xh[11];
    m.xch[5] = qC[0] * xh[12] + qC[1] * xh[13] + qC[2] * xh[14];
  }
  return true;
}

// (A packed-FP32 form of this block -- v_pk_mul_f32 / v_pk_add_f32, entries (i, 2m) and (i, 2m + 1) of a row in one instruction, 383 -> 320
//  instructions, bit-identical -- measured 5.25-5.28 ms on the bench step against 5.22 for this scalar form and was removed: the item loop was
//  not bound by issue slots.)
template <bool NEED_H>
__device__ __forceinline__ void strict_item_back(const float (&xj)[8], const StrictMid& m, const float gd2, double (&acc)[kStrictAccum]) {
  if (NEED_H) {
    const float pg13 = xj[0], pg23 = xj[1];
    const float pg4[3] = {xj[2], xj[3], xj[4]}, pg5[3] = {xj[5], xj[6], xj[7]};
    // full C * J (3 x 6) as a lookup: column i < 3 -> C[r][i], else cPG[r][i - 3]
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const float ng = -gd2 * m.g6[i];
      const float cj0 = (i < 3) ? m.C[0][i < 3 ? i : 0] : m.cPG[0][i < 3 ? 0 : i - 3];
      const float cj1 = (i < 3) ? m.C[1][i < 3 ? i : 0] : m.cPG[1][i < 3 ? 0 : i - 3];
      const float cj2 = (i < 3) ? m.C[2][i < 3 ? i : 0] : m.cPG[2][i < 3 ? 0 : i - 3];
#pragma unroll
      for (int j = 0; j < 6; j++) {
        float t = ng * m.g6[j];
        if (i >= 3 && j >= 3) {
          const int lo = (i < j ? i : j) - 3, hi = (i < j ? j : i) - 3;
          t = t + m.xch[lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)];
        }
        // J_j^T (C J_i): column j of J is a unit vector for j < 3 and has a zero first entry for j == 3
        const float pcp = (j == 0) ? cj0 : (j == 1) ? cj1 : (j == 2) ? cj2 : (j == 3) ? (pg13 * cj1 + pg23 * cj2)
                        : (j == 4) ? (pg4[0] * cj0 + pg4[1] * cj1 + pg4[2] * cj2) : (pg5[0] * cj0 + pg5[1] * cj1 + pg5[2] * cj2);
        acc[7 + i * 6 + j] += (double)(m.e * (t + pcp));
      }
    }
  }
  acc[0] += (double)m.score_inc;
}

// One function, straight line (the front / back halves above are the same operations cut in two for the hand-pipelined A/B build).
template <bool NEED_H, bool GLIBC_ONLY = false>
__device__ __forceinline__ void strict_item(const float (&xt)[3], const float (&xj)[8], const float (&xh)[15], const VoxelStrictRec* __restrict__ rec,
                                            const double gauss_d1, const float gd2, double (&acc)[kStrictAccum], const unsigned long long* __restrict__ exptab) {
  const float pg13 = xj[0], pg23 = xj[1];
  const float pg4[3] = {xj[2], xj[3], xj[4]}, pg5[3] = {xj[5], xj[6], xj[7]};
  const float4* __restrict__ r4 = reinterpret_cast<const float4*>(rec);
  const float4 ra = r4[0], rb = r4[1], rc = r4[2], rd = r4[3];
  const double m0 = __hiloint2double(__float_as_int(ra.y), __float_as_int(ra.x)), m1 = __hiloint2double(__float_as_int(ra.w), __float_as_int(ra.z)),
               m2 = __hiloint2double(__float_as_int(rb.y), __float_as_int(rb.x));
  const float q0 = (float)((double)xt[0] - m0), q1 = (float)((double)xt[1] - m1), q2 = (float)((double)xt[2] - m2);
  const float C[3][3] = {{rb.z, rb.w, rc.x}, {rc.y, rc.z, rc.w}, {rd.x, rd.y, rd.z}};
  float qC[3];
#pragma unroll
  for (int c = 0; c < 3; c++) qC[c] = q0 * C[0][c] + q1 * C[1][c] + q2 * C[2][c];
  const float e_arg = -gd2 * (q0 * qC[0] + q1 * qC[1] + q2 * qC[2]) * 0.5f;
#ifdef DGS_AB_DETF   // timing A/B only (`make ab AB=-DDGS_AB_DETF`): the rounds 1-3 polynomial in the item-compacted kernel -- 5.39-5.41 ms per bench step
                     // against 5.38 with glibc's expf (same box, two runs each): the exact libm exponential costs nothing
  float e = det_expf(e_arg);
#else
  float e = (GLIBC_ONLY || exptab) ? glibc_expf_dev(e_arg, exptab) : det_expf(e_arg);   // std::exp(float) as glibc computes it / rounds 1-3's polynomial
#endif
  const float score_inc = (float)(-gauss_d1 * (double)e);
  e = gd2 * e;
  if (e > 1 || e < 0 || e != e) return;
  e = (float)((double)e * gauss_d1);
  // C * point gradient: columns 0..2 are C itself (unit columns), column 3 has a zero first factor
  float cPG[3][3];   // columns 3, 4, 5
#pragma unroll
  for (int r = 0; r < 3; r++) {
    cPG[r][0] = C[r][1] * pg13 + C[r][2] * pg23;
    cPG[r][1] = C[r][0] * pg4[0] + C[r][1] * pg4[1] + C[r][2] * pg4[2];
    cPG[r][2] = C[r][0] * pg5[0] + C[r][1] * pg5[1] + C[r][2] * pg5[2];
  }
  float g6[6];
  g6[0] = qC[0]; g6[1] = qC[1]; g6[2] = qC[2];   // q^T (C * unit column) is q^T C: the same operations
#pragma unroll
  for (int c = 0; c < 3; c++) g6[3 + c] = q0 * cPG[0][c] + q1 * cPG[1][c] + q2 * cPG[2][c];
#pragma unroll
  for (int c = 0; c < 6; c++) acc[1 + c] += (double)(e * g6[c]);
  if (NEED_H) {
    // x^T C H for the six distinct vectors: a = (0, xh0, xh1) b = (0, xh2, xh3) c = (0, xh4, xh5) d = xh6..8 e = xh9..11 f = xh12..14
    float xch[6];
    xch[0] = qC[1] * xh[0] + qC[2] * xh[1];
    xch[1] = qC[1] * xh[2] + qC[2] * xh[3];
    xch[2] = qC[1] * xh[4] + qC[2] * xh[5];
    xch[3] = qC[0] * xh[6] + qC[1] * xh[7] + qC[2] * xh[8];
    xch[4] = qC[0] * xh[9] + qC[1] * xh[10] + qC[2] * xh[11];
    xch[5] = qC[0] * xh[12] + qC[1] * xh[13] + qC[2] * xh[14];
    // full C * J (3 x 6) as a lookup: column i < 3 -> C[r][i], else cPG[r][i - 3]
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const float ng = -gd2 * g6[i];
      const float cj0 = (i < 3) ? C[0][i < 3 ? i : 0] : cPG[0][i < 3 ? 0 : i - 3];
      const float cj1 = (i < 3) ? C[1][i < 3 ? i : 0] : cPG[1][i < 3 ? 0 : i - 3];
      const float cj2 = (i < 3) ? C[2][i < 3 ? i : 0] : cPG[2][i < 3 ? 0 : i - 3];
#pragma unroll
      for (int j = 0; j < 6; j++) {
        float t = ng * g6[j];
        if (i >= 3 && j >= 3) {
          const int lo = (i < j ? i : j) - 3, hi = (i < j ? j : i) - 3;
          t = t + xch[lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)];
        }
        // J_j^T (C J_i): column j of J is a unit vector for j < 3 and has a zero first entry for j == 3
        const float pcp = (j == 0) ? cj0 : (j == 1) ? cj1 : (j == 2) ? cj2 : (j == 3) ? (pg13 * cj1 + pg23 * cj2)
                        : (j == 4) ? (pg4[0] * cj0 + pg4[1] * cj1 + pg4[2] * cj2) : (pg5[0] * cj0 + pg5[1] * cj1 + pg5[2] * cj2);
        acc[7 + i * 6 + j] += (double)(e * (t + pcp));
      }
    }
  }
  acc[0] += (double)score_inc;
}

// the point's products with the float angle tables (computePointDerivatives).  As in the default order's kernel: rows 5..7 of the first
// table and rows 4, 5, 9..14 of the second have an exact zero z entry (the term adds +-0: skipped), and nine of the fifteen second-table rows
// are first-table rows again -- the same double expressions or their exact negations (a2 = -b, a3 = a, b2 = -e, b3 = d, c2 = -h, c3 = g,
// f1 / f2 / f3 = the xy parts of d1 / a2 / a3), so the float products are the same bits or their negations (IEEE rounding is symmetric).
template <bool NEED_H>
__device__ __forceinline__ void strict_point_tables(const float4 x, const NdtPair& st, float (&xj)[8], float (&xh)[15]) {
  const float jxy0 = st.jang[0][0] * x.x + st.jang[0][1] * x.y, jxy1 = st.jang[1][0] * x.x + st.jang[1][1] * x.y;
  xj[0] = jxy0 + st.jang[0][2] * x.z;
  xj[1] = jxy1 + st.jang[1][2] * x.z;
#pragma unroll
  for (int i = 2; i < 5; i++) xj[i] = st.jang[i][0] * x.x + st.jang[i][1] * x.y + st.jang[i][2] * x.z;
#pragma unroll
  for (int i = 5; i < 8; i++) xj[i] = st.jang[i][0] * x.x + st.jang[i][1] * x.y;
  if (NEED_H) {
    xh[0] = -xj[1]; xh[1] = xj[0]; xh[2] = -xj[4]; xh[3] = xj[3]; xh[4] = -xj[7]; xh[5] = xj[6];
    const float hxy6 = st.hang[6][0] * x.x + st.hang[6][1] * x.y;
    xh[6] = hxy6 + st.hang[6][2] * x.z;
#pragma unroll
    for (int i = 7; i < 9; i++) xh[i] = st.hang[i][0] * x.x + st.hang[i][1] * x.y + st.hang[i][2] * x.z;
#pragma unroll
    for (int i = 9; i < 12; i++) xh[i] = st.hang[i][0] * x.x + st.hang[i][1] * x.y;
    xh[12] = hxy6; xh[13] = -jxy1; xh[14] = jxy0;
  } else {
#pragma unroll
    for (int i = 0; i < 15; i++) xh[i] = 0.f;
  }
}

// one point, its valid voxels one after the other (a wave runs as many rounds as its fullest point has voxels)
template <int SEARCH, bool NEED_H>
__device__ __forceinline__ void strict_point(const float4 x, const float (&xt)[3], const int (&vids)[Offsets<SEARCH>::N], unsigned mask, const NdtPair& st,
                                             const VoxelStrictRec* __restrict__ vs, const double gauss_d1, const float gd2, double (&acc)[kStrictAccum],
                                             const unsigned long long* __restrict__ exptab) {
  float xj[8], xh[15];
  strict_point_tables<NEED_H>(x, st, xj, xh);
  while (mask) {
    const int k = __ffs(mask) - 1;
    mask &= mask - 1u;
    strict_item<NEED_H>(xt, xj, xh, vs + strict_pick(vids, k), gauss_d1, gd2, acc, exptab);
  }
}

// ---- kind 2: computeHessian / updateHessian in PCL's double form, ONE (point, voxel) item.
// xj / xh: the point's products with the DOUBLE angle vectors.  Returns false when the voxel's weight fails upstream's test.
// ROWS = false: the 36 terms are added to acc[7 ..]; ROWS = true (ndt_strict_order 2): written to rows[entry * row_stride].
template <bool ROWS>
__device__ __forceinline__ bool strict_item_hd(const float (&xt)[3], const double (&xj)[8], const double (&xh)[15], const double* __restrict__ rec,
                                               const double gauss_d1, const double gauss_d2, double (&acc)[kStrictAccum], double* __restrict__ rows, const size_t row_stride,
                                               const unsigned long long* __restrict__ exptab_d) {
  const double pg13 = xj[0], pg23 = xj[1];
  const double pg4[3] = {xj[2], xj[3], xj[4]}, pg5[3] = {xj[5], xj[6], xj[7]};
  double q[3], C[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++) q[r] = (double)xt[r] - rec[r];   // mean[3], icov[9] row-major
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) C[r][c] = rec[3 + r * 3 + c];
  double Cq[3];
#pragma unroll
  for (int r = 0; r < 3; r++) Cq[r] = C[r][0] * q[0] + C[r][1] * q[1] + C[r][2] * q[2];
  const double e_arg = -gauss_d2 * (q[0] * Cq[0] + q[1] * Cq[1] + q[2] * Cq[2]) / 2;
  double e = gauss_d2 * (exptab_d ? glibc_exp_dev(e_arg, exptab_d) : det_exp(e_arg));   // std::exp(double) as glibc computes it (table: kGlibcExpTab or its LDS copy) / rounds 1-3's polynomial
  if (e > 1 || e < 0 || e != e) return false;
  e *= gauss_d1;
  // x_trans . (c_inv * point_hessian block) for the six distinct vectors a = (0, xh0, xh1) b c d = xh6..8 e f
  double xch[6];
#pragma unroll
  for (int v = 0; v < 6; v++) {
    double Ch[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
      Ch[r] = (v < 3) ? (C[r][1] * xh[v < 3 ? 2 * v : 0] + C[r][2] * xh[v < 3 ? 2 * v + 1 : 0])
                      : (C[r][0] * xh[v < 3 ? 0 : 3 * v - 3] + C[r][1] * xh[v < 3 ? 0 : 3 * v - 2] + C[r][2] * xh[v < 3 ? 0 : 3 * v - 1]);
    xch[v] = q[0] * Ch[0] + q[1] * Ch[1] + q[2] * Ch[2];
  }
  // cov_dxd_pi = c_inv * point_gradient.col(i): column i of C for i < 3
  double cd[6][3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    cd[0][r] = C[r][0]; cd[1][r] = C[r][1]; cd[2][r] = C[r][2];
    cd[3][r] = C[r][1] * pg13 + C[r][2] * pg23;
    cd[4][r] = C[r][0] * pg4[0] + C[r][1] * pg4[1] + C[r][2] * pg4[2];
    cd[5][r] = C[r][0] * pg5[0] + C[r][1] * pg5[1] + C[r][2] * pg5[2];
  }
  double A[6];   // x_trans . (c_inv * point_gradient.col(i))
#pragma unroll
  for (int i = 0; i < 6; i++) A[i] = q[0] * cd[i][0] + q[1] * cd[i][1] + q[2] * cd[i][2];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const double nA = -gauss_d2 * A[i];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double t = nA * A[j];
      if (i >= 3 && j >= 3) {
        const int lo = (i < j ? i : j) - 3, hi = (i < j ? j : i) - 3;
        t = t + xch[lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)];
      }
      const double D = (j < 3) ? cd[i][j < 3 ? j : 0] : (j == 3) ? (pg13 * cd[i][1] + pg23 * cd[i][2])
                     : (j == 4) ? (pg4[0] * cd[i][0] + pg4[1] * cd[i][1] + pg4[2] * cd[i][2]) : (pg5[0] * cd[i][0] + pg5[1] * cd[i][1] + pg5[2] * cd[i][2]);
      const double term = e * (t + D);
      if (ROWS) rows[(size_t)(i * 6 + j) * row_stride] = term;
      else acc[7 + i * 6 + j] += term;
    }
  }
  return true;
}

__device__ __forceinline__ void strict_point_tables_hd(const float4 xf, const NdtPair& st, double (&xj)[8], double (&xh)[15]) {
  const double x[3] = {(double)xf.x, (double)xf.y, (double)xf.z};
#pragma unroll
  for (int i = 0; i < 8; i++) xj[i] = x[0] * st.jang_d[i][0] + x[1] * st.jang_d[i][1] + x[2] * st.jang_d[i][2];
#pragma unroll
  for (int i = 0; i < 15; i++) xh[i] = x[0] * st.hang_d[i][0] + x[1] * st.hang_d[i][1] + x[2] * st.hang_d[i][2];
}

// One point.  ROWS = true: every slot is visited and written -- zeros where a slot holds no voxel or the voxel's weight fails upstream's
// test (adding +0.0 changes nothing) -- entry-major: rows[entry * row_stride + slot k of this point].
template <int SEARCH, bool ROWS>
__device__ __forceinline__ void strict_point_hd(const float4 xf, const float (&xt)[3], const int (&vids)[Offsets<SEARCH>::N], const unsigned mask_in, const NdtPair& st,
                                                const double* __restrict__ vtab, const double gauss_d1, const double gauss_d2, double (&acc)[kStrictAccum],
                                                double* __restrict__ rows, const size_t row_stride, const unsigned long long* __restrict__ exptab_d) {
  constexpr int NB = Offsets<SEARCH>::N;
  double xj[8], xh[15];
  strict_point_tables_hd(xf, st, xj, xh);
  unsigned mask = ROWS ? ((NB >= 32) ? 0xFFFFFFFFu : ((1u << NB) - 1u)) : mask_in;
  while (mask) {
    const int k = __ffs(mask) - 1;
    mask &= mask - 1u;
    bool ok = ((mask_in >> k) & 1u) != 0;
    if (ok) ok = strict_item_hd<ROWS>(xt, xj, xh, vtab + (size_t)strict_pick(vids, k) * 12, gauss_d1, gauss_d2, acc, ROWS ? rows + k : nullptr, row_stride, exptab_d);
    if (ROWS && !ok) {
      for (int en = 0; en < 36; en++) rows[(size_t)en * row_stride + k] = 0.0;
    }
  }
}

// Block reduction of the 43 per-thread totals into one row of kStrictPad doubles (ncol = 7 for a score + gradient evaluation): every wave
// transposes through LDS, 11 values at a time (lane l stores value k at row k, then lane k adds the 64 entries of row k in lane order),
// the four waves' sums are added in wave order.  Same construction as ndt_block_row, same reason: a DPP / shuffle butterfly over 43
// doubles costs ~1,500 wave-instructions.
constexpr int kStrictRowScratch = 11 * 65;   // doubles of per-wave transposition scratch
template <bool COHERENT, bool OWN_SCRATCH = true>
__device__ __forceinline__ void strict_block_row(const double (&acc)[kStrictAccum], const int ncol, double* __restrict__ row_of_slice, double* wave_scratch = nullptr) {
  constexpr int CH = 11, RS = 65, NCH = 4;
  __shared__ double tr[OWN_SCRATCH ? kBlock / kWave : 1][OWN_SCRATCH ? CH * RS : 1];
  __shared__ double sm[kBlock / kWave][kStrictPad];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* my = OWN_SCRATCH ? tr[wave] : wave_scratch;   // !OWN_SCRATCH: this wave's own LDS region (>= kStrictRowScratch doubles), free by now
#pragma unroll
  for (int h = 0; h < NCH; h++) {
    if (h * CH >= ncol) break;   // wave-uniform
#pragma unroll
    for (int k = 0; k < CH; k++) my[k * RS + lane] = (h * CH + k < kStrictAccum) ? acc[h * CH + k < kStrictAccum ? h * CH + k : 0] : 0.0;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    if (lane < CH) {
      double v = 0.0;
#pragma unroll 8
      for (int j = 0; j < 64; j++) v += my[lane * RS + j];
      sm[wave][h * CH + lane] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
  __syncthreads();
  if (threadIdx.x < kStrictPad) {
    double v = 0.0;
    if ((int)threadIdx.x < ncol) v = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
    double* row = row_of_slice + threadIdx.x;
    if (COHERENT) handoff_store_row(row, v);
    else *row = v;
  }
}

// The same reduction cut in two for the item-compacted kernel, whose workgroups write one row per SLICE and several slices each: the per-wave half
// (no workgroup barrier: the four waves of a workgroup run through their tiles independently -- with a barrier per slice the step took 6.8 ms
// instead of 5.4, the waves waiting for the slowest of the four at every slice) leaves the wave's column sums in LDS, slot by slot; the rows of
// up to kStrictSlots slices are then written together behind ONE barrier, each row = ((wave 0 + wave 1) + wave 2) + wave 3 as before.
constexpr int kStrictSlots = 4;
__device__ __forceinline__ void strict_wave_cols(const double (&acc)[kStrictAccum], const int ncol, double* __restrict__ my, double* __restrict__ cols_of_wave) {
  constexpr int CH = 11, RS = 65, NCH = 4;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int h = 0; h < NCH; h++) {
    if (h * CH >= ncol) break;   // wave-uniform
#pragma unroll
    for (int k = 0; k < CH; k++) my[k * RS + lane] = (h * CH + k < kStrictAccum) ? acc[h * CH + k < kStrictAccum ? h * CH + k : 0] : 0.0;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    if (lane < CH) {
      double v = 0.0;
#pragma unroll 8
      for (int j = 0; j < 64; j++) v += my[lane * RS + j];
      cols_of_wave[h * CH + lane] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
}
// rows of slices q_first, q_first + q_step, ... (n_slots of them) from the waves' column sums; every thread of the workgroup calls it
template <bool COHERENT>
__device__ __forceinline__ void strict_rows_flush(const double (*cols)[kBlock / kWave][kStrictPad], const int n_slots, const int ncol, double* __restrict__ rows_of_pair,
                                                  const int q_first, const int q_step) {
  __syncthreads();
  const int slot = threadIdx.x / kStrictPad, col = threadIdx.x % kStrictPad;
  if (slot < n_slots) {
    double v = 0.0;
    if (col < ncol) v = ((cols[slot][0][col] + cols[slot][1][col]) + cols[slot][2][col]) + cols[slot][3][col];
    double* row = rows_of_pair + (size_t)(q_first + slot * q_step) * kStrictPad + col;
    if (COHERENT) handoff_store_row(row, v);
    else *row = v;
  }
  __syncthreads();
}

// Sums a pair's rows in slice order and advances its optimiser by one evaluation (one whole workgroup; launch >= 0: inside the fused
// launch, rows read with the hand-off's coherent loads, the pair leaves through last_launch + its host flag).
// HD / ONE_KERNEL: which kernel of the round runs this closing -- see NdtPair::serve.  ONE_KERNEL (ndt_strict3_kernel): every evaluation
// kind is served by the round's only launch.
#ifdef DGS_CLOSE_STAMPS   // diagnostic build (make dbg): 100 MHz wall-clock stamps of the closing phases while the pair is in its second iteration
#define STRICT_STAMP(k) if (threadIdx.x == 0 && st->s.nr_iterations == 1) st->traj[kTrajCap - 1][k] = (double)wall_clock64();
#else
#define STRICT_STAMP(k)
#endif
#ifdef DGS_STRICT_CLOSE_NOINLINE
#define DGS_CLOSE_INLINE __noinline__
#else
#define DGS_CLOSE_INLINE __forceinline__
#endif
template <bool HD, bool ONE_KERNEL = false>
__device__ DGS_CLOSE_INLINE void ndt_close_strict(NdtPair* st, const double* rows_of_pair, const int blocks_per_pair, const NdtConsts& c, int* done_flag, const int launch,
                                                 const int hd_lag = 1, const bool defer_solve = false, const bool speculate = false) {
  STRICT_STAMP(0)
  __shared__ NdtSolver s_lds;
  NdtSolver& s = s_lds;
  static_assert(sizeof(NdtSolver) % 8 == 0 && sizeof(NdtSolver) / 8 <= kBlock, "state words");
  constexpr int kWords = (int)(sizeof(NdtSolver) / 8);
  double word = 0.0;
  // a speculated step behind this evaluation (NdtPair::spec_s): the exact state and its verdict were left by a wave of THIS launch -> coherent loads
  const int spec_pending = st->spec_pending;
  const int spec_result = spec_pending ? __hip_atomic_load(&st->spec_result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
  if (threadIdx.x < kWords)
    word = (spec_pending && spec_result != 0) ? __hip_atomic_load(reinterpret_cast<const double*>(&st->spec_s) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                        : reinterpret_cast<const double*>(&st->s)[threadIdx.x];
  const int kind = st->need_hessian;
  __shared__ double tot[kStrictPad];
  constexpr int G = kBlock / kStrictPad;   // 5 groups of 48 columns; threads 240.. idle
  __shared__ double sm[G][kStrictPad];
  const int col = threadIdx.x % kStrictPad, grp = threadIdx.x / kStrictPad;
  if (grp < G) {
    double v = 0.0;
    for (int b0 = grp; b0 < blocks_per_pair; b0 += 4 * G) {
      double r[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int b = b0 + k * G;
        const double* ptr = rows_of_pair + (size_t)min(b, blocks_per_pair - 1) * kStrictPad + col;
        const double x = (launch >= 0) ? handoff_load_row(ptr) : *ptr;
        r[k] = (b < blocks_per_pair) ? x : 0.0;
      }
      v = (((v + r[0]) + r[1]) + r[2]) + r[3];
    }
    sm[grp][col] = v;
  }
  if (threadIdx.x < kWords) reinterpret_cast<double*>(&s_lds)[threadIdx.x] = word;
  __syncthreads();
  if (threadIdx.x < kStrictPad) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < G; k++) t += sm[k][threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
  STRICT_STAMP(1)
  if (threadIdx.x >= kWave) return;
  const bool writer = threadIdx.x == 0;
  // spec_result 1: the state just loaded IS the exact one (its step yields the published header): consume the evaluation as ever.
  // spec_result 2 (or a missing verdict): the evaluation was made from a header the exact step does not yield, or the exact step ends the
  // registration: drop its sums and take the step exactly, here (the state loaded is the exact wave's copy; its inputs are the iteration's).
  const bool redo = spec_pending && spec_result != 1;
  if (!redo) {
    const int t = threadIdx.x;
    if (t < 36) {
      if (kind) s.hess[t] = tot[7 + t];   // upstream's full 6 x 6 (not exactly symmetric)
    } else if (t < 42) {
      if (kind != 2) s.grad[t - 36] = tot[1 + t - 36];
    } else if (t == 42) {
      if (kind != 2) s.score = tot[0];    // computeHessian alone (kind 2) leaves score and gradient as the last trial left them
    }
  } else if (threadIdx.x == 0) {
    s.phase = PH_SOLVE_PENDING;           // ndt_advance resumes in front of the Newton step, without counting an evaluation
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  STRICT_STAMP(2)
  bool speculated = false;
  // (a redo takes ITS step exactly and publishes that: speculating again would yield the very header that was just refused, for ever --
  //  seen on a 2,048-point planar pair whose rotation components of the step, ~1e-8 rad, differ in the 6th digit between the two solvers)
  ndt_advance<false, false, true>(st, st, s, c, writer, defer_solve, speculate && !redo, &speculated);
  STRICT_STAMP(3)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  for (int w = threadIdx.x; w < kWords; w += kWave) reinterpret_cast<double*>(&st->s)[w] = reinterpret_cast<const double*>(&s_lds)[w];
  STRICT_STAMP(4)
  if (writer) {
    st->spec_pending = speculated ? 1 : 0;
    if (speculated) st->spec_result = 0;
    if (s.phase == PH_DONE) {
      st->active = 0;
      if (launch >= 0) {
        st->last_launch = launch;
        __hip_atomic_store(done_flag, launch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        atomicAdd(done_flag, 1);
      }
    } else if (launch >= 0) {
      // which kernel of which round serves the evaluation just queued (NdtPair::serve); `launch` is the round number
      if (s.phase == PH_SOLVE_PENDING) st->serve[1] = launch;   // ndt_strict_solve_kernel of this round (its own stream) takes the Newton step
      else if (ONE_KERNEL) st->serve[0] = launch + 1;
      else if (st->need_hessian == 2) st->serve[1] = HD ? launch + hd_lag : launch;   // the double computeHessian kernel of this round follows this launch
      else if (HD) st->serve[2] = launch + hd_lag;   // see NdtPair::serve
      else st->serve[0] = launch + 1;
    }
  }
}

// FUSED: round number `launch` of the align; the workgroup that takes the pair's last ticket closes the evaluation (ndt_close_strict).
// Otherwise derivatives only, rows summed by ndt_solve_kernel (DGS_NDT_FUSED=0, the test hook).
// HD = false: the pairs whose evaluation in flight is of kind 0 / 1 (updateDerivatives in float); HD = true: the pairs waiting for kind 2
// (computeHessian in double).  Two instantiations, launched one behind the other in every round, because the double pass keeps ~130
// registers of double tables next to the accumulators: in one kernel it pushed every path to 256 VGPRs and 87 spilled registers.  A pair
// is served by exactly one of the two in a round -- or by both when the first one's closing asks for kind 2, which the second one then
// evaluates in the same round (the kernel boundary orders it; results do not depend on it).
template <int SEARCH, bool FUSED, bool HD>
__global__ __launch_bounds__(kBlock, 2) void ndt_strict_kernel(const float4* const* __restrict__ src_ptrs, const int* __restrict__ src_sizes, NdtPair* __restrict__ pairs,
                                                              const VoxelGrid g, const VoxelStrictRec* __restrict__ vs, const double* __restrict__ vtab,
                                                              const double gauss_d1, const double gauss_d2, const int leaf_pow2, double* __restrict__ partials,
                                                              const int n_pairs, const int cap_blocks, int* __restrict__ pair_blocks, const NdtConsts consts,
                                                              int* __restrict__ done_flags, const int launch, const int hd_lag) {
  int pair, slice, blocks_per_pair;
  if (!deal_workgroup(n_pairs, cap_blocks, [&](int pi) {
        if (!FUSED) return pairs[pi].active != 0 && (pairs[pi].need_hessian == 2) == HD;
        return HD ? (launch == pairs[pi].serve[1]) : (launch <= pairs[pi].serve[0] || launch == pairs[pi].serve[2]);
      }, pair, slice, blocks_per_pair)) return;
  if (slice == 0 && threadIdx.x == 0) pair_blocks[pair] = blocks_per_pair;
  const NdtPair& st = pairs[pair];
  const float4* __restrict__ src = src_ptrs[pair];
  const int n = src_sizes[pair];
  const int kind = HD ? 2 : st.need_hessian;   // 0: score + gradient, 1: + Hessian (float), 2: Hessian alone in double (computeHessian)
  const float gd2 = (float)gauss_d2;
  const unsigned long long* __restrict__ exptab = consts.exp_libm ? kGlibcExp2fTab : nullptr;
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = st.T[k];
  double acc[kStrictAccum];
#pragma unroll
  for (int k = 0; k < kStrictAccum; k++) acc[k] = 0.0;
  constexpr int NB = Offsets<SEARCH>::N;
  for (int i = slice * kBlock + threadIdx.x; i < n; i += blocks_per_pair * kBlock) {
    const float4 x = src[i];
    float xt[3];
    xt[0] = affine_row_rn(T[0], T[1], T[2], T[3], x.x, x.y, x.z);
    xt[1] = affine_row_rn(T[4], T[5], T[6], T[7], x.x, x.y, x.z);
    xt[2] = affine_row_rn(T[8], T[9], T[10], T[11], x.x, x.y, x.z);
    int vids[NB];
    const unsigned mask = strict_neighbourhood<SEARCH>(xt, g, leaf_pow2, vids);
    if (!mask) continue;
    if (HD) strict_point_hd<SEARCH, false>(x, xt, vids, mask, st, vtab, gauss_d1, gauss_d2, acc, nullptr, 0, consts.exp_libm ? kGlibcExpTab : nullptr);
    else if (kind == 1) strict_point<SEARCH, true>(x, xt, vids, mask, st, vs, gauss_d1, gd2, acc, exptab);
    else strict_point<SEARCH, false>(x, xt, vids, mask, st, vs, gauss_d1, gd2, acc, exptab);
  }
  strict_block_row<FUSED>(acc, kind ? kStrictAccum : 7, partials + ((size_t)pair * cap_blocks + slice) * kStrictPad);
  if (!FUSED) return;
  __shared__ int s_last;
  if (threadIdx.x < kStrictPad) handoff_drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) s_last = handoff_take_ticket(&pairs[pair].ticket, blocks_per_pair) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
#ifdef DGS_CLOSE_STAMPS
  if (threadIdx.x == 0 && pairs[pair].s.nr_iterations == 1) pairs[pair].traj[kTrajCap - 1][5] = (double)wall_clock64();
#endif
  ndt_close_strict<HD>(pairs + pair, partials + (size_t)pair * cap_blocks * kStrictPad, blocks_per_pair, consts, done_flags + pair, launch, hd_lag);
}

// ================================================================================================ item-compacted kernel (v3)
// The kernel above gives a lane a POINT and lets it walk the point's valid voxels: a wave runs as many rounds as its fullest point has
// voxels (5.5-6 on scan data against 4.0 on average), and the double-precision computeHessian pass needs a kernel of its own because its
// per-point double tables do not fit the register file next to the accumulators.  Here a lane gets an ITEM -- one (point, voxel) pair:
//   * a wave takes a tile of its points (128, or 64 for the 27-slot searches and the double pass), lane = point: transform, neighbourhood,
//     the point's products with the angle tables -> LDS, field-major ([field][slot]: conflict-free both ways); every valid (slot, voxel)
//     is appended to the wave's item queue in LDS at a position made from ballots, so the order of the items -- and with it the order of
//     every double addition -- is a function of the data alone;
//   * then lane l of round r processes item 64 r + l: it reads its point's tables from LDS and the voxel's record from L2 and adds the
//     item's float increments to ITS double accumulators.  Every round but the last of a tile has all 64 lanes busy.
// Which thread adds which item differs from the kernel above, i.e. the association of the double sums differs once more (~1e-14); every
// float operation is the same (the same strict_item / strict_item_hd).  All three evaluation kinds run in ONE launch per round.
// WITH_HD = false: the kernel serves the float kinds only (64-point tiles, a third of the LDS, no double tables in registers: three waves
// per SIMD instead of two); the pairs waiting for kind 2 are served by ndt_strict_kernel<SEARCH, FUSED, HD = true> as the round's second launch.

// The float items of one tile, 64 per round, lane <-> item.
// A/B build `make ab AB=-DDGS_STRICT_ITEMS=1`: the loop software-pipelined by hand -- the NEXT round's queue entry read at the top, its voxel record
// (four 16-byte gathers from L2) requested between the two halves of the current item, the 36-entry Hessian block behind covering the latency.
// Measured on the bench step (same box, two runs each): 5.65 ms against 5.48 for the plain loop -- the 15 more live registers turn 4 spilled
// registers into 29 in a kernel that sits at 256; the wave next door on the SIMD was hiding most of that latency already.  Not the default.
template <bool NEED_H, int PTS>
__device__ __forceinline__ void strict_items_float(const float* __restrict__ tf, const unsigned* __restrict__ queue, const int qn, const int lane,
                                                   const VoxelStrictRec* __restrict__ vs, const double gauss_d1, const float gd2, double (&acc)[kStrictAccum],
                                                   const unsigned long long* __restrict__ exptab) {
#if !defined(DGS_STRICT_ITEMS) || DGS_STRICT_ITEMS != 1   // the plain loop: the record loaded where it is used
#pragma unroll 1
  for (int h = 0; h < qn; h += 64) {
    const int idx = h + lane;
    if (idx < qn) {
      const unsigned entry = queue[idx];
      const int slot = (int)(entry >> 25);
      float xt[3], xj[8], xh[15];
#pragma unroll
      for (int f = 0; f < 3; f++) xt[f] = tf[f * PTS + slot];
#pragma unroll
      for (int f = 0; f < 8; f++) xj[f] = tf[(3 + f) * PTS + slot];
#pragma unroll
      for (int f = 0; f < 15; f++) xh[f] = NEED_H ? tf[(11 + f) * PTS + slot] : 0.f;
      strict_item<NEED_H, true>(xt, xj, xh, vs + (entry & 0x1FFFFFFu), gauss_d1, gd2, acc, exptab);
    }
  }
#else
  bool have = lane < qn;
  unsigned entry = have ? queue[lane] : 0u;
  StrictRecWords w = strict_load_rec(vs + (entry & 0x1FFFFFFu));   // (entry 0: voxel 0, a valid address; its words are not used)
#pragma unroll 1
  for (int h = 0; h < qn; h += 64) {
    const int idx_n = h + 64 + lane;
    const bool have_n = idx_n < qn;
    const unsigned entry_n = have_n ? queue[idx_n] : 0u;
    float xj[8];
    StrictMid m;
    bool alive = false;
    if (have) {
      const int slot = (int)(entry >> 25);
      float xt[3], xh[15];
#pragma unroll
      for (int f = 0; f < 3; f++) xt[f] = tf[f * PTS + slot];
#pragma unroll
      for (int f = 0; f < 8; f++) xj[f] = tf[(3 + f) * PTS + slot];
      if (NEED_H) {
#pragma unroll
        for (int f = 0; f < 15; f++) xh[f] = tf[(11 + f) * PTS + slot];
      } else {
#pragma unroll
        for (int f = 0; f < 15; f++) xh[f] = 0.f;
      }
      alive = strict_item_front<NEED_H, true>(xt, xj, xh, w, gauss_d1, gd2, acc, m, exptab);
    }
    StrictRecWords wn = w;
    if (have_n) wn = strict_load_rec(vs + (entry_n & 0x1FFFFFFu));
    if (alive) strict_item_back<NEED_H>(xj, m, gd2, acc);
    w = wn;
    entry = entry_n;
    have = have_n;
  }
#endif
}

template <int SEARCH, bool WITH_HD>
struct StrictTile {
  static constexpr int NB = Offsets<SEARCH>::N;
#ifndef DGS_STRICT_PTS_FLOAT_ONLY
#define DGS_STRICT_PTS_FLOAT_ONLY 64
#endif
  static constexpr int PTS = (NB <= 7) ? (WITH_HD ? 128 : DGS_STRICT_PTS_FLOAT_ONLY) : 64;   // points per wave and tile (float kinds)
  static constexpr int PTS_HD = 64;                  // double pass: 23 doubles + 3 floats per point
  static constexpr int kFields = 26;                 // xt[3], xj[8], xh[15]
  static constexpr int kTableBytes = (!WITH_HD || kFields * PTS * 4 > (23 * 8 + 3 * 4) * PTS_HD) ? kFields * PTS * 4 : (23 * 8 + 3 * 4) * PTS_HD;
  static constexpr int kQueue = PTS * NB;            // items of a tile at most
};

// DGS_NDT_FIXED_SLICES=1 (dgs_handle::ndt_fixed_slices; off by default): the slices of a pair as a function of its OWN point count -- 512 points
// per slice, at least 64 slices while every thread still has a point, at most the rows reserved per pair -- instead of "one slice per workgroup
// the launch could spare".  A slice is the unit whose double sums make one row, so with fixed slices neither the workgroup that serves a slice
// nor the composition of the launch enters the association of the sums: the same pose gives the same doubles in every launch, and a pair's
// result does not depend on what else is in its batch (tests/test_round4_gpu.py).  It costs 20 % of the bench step (6.5 against 5.4 ms: four
// per-slice reductions per workgroup instead of one while 32 pairs iterate), and the case that made it necessary -- More-Thuente comparing
// the values of a REPEATED trial point, found by the soak at transformation_epsilon = 0.1 -- is closed for free by NdtSolver::trial_x, so it
// stays an option for callers that need batch-independent bits.
__device__ __forceinline__ int strict_slices_of(const int n, const int cap_blocks) {
#ifndef DGS_STRICT_SLICE_PPT
#define DGS_STRICT_SLICE_PPT 2   // points per thread and slice (A/B: make ab AB=-DDGS_STRICT_SLICE_PPT=8)
#endif
  const int by_points = (n + DGS_STRICT_SLICE_PPT * kBlock - 1) / (DGS_STRICT_SLICE_PPT * kBlock), at_least = min(64 * 2 / DGS_STRICT_SLICE_PPT, (n + kBlock - 1) / kBlock);
  return max(1, min(max(by_points, at_least), cap_blocks));
}

template <int SEARCH, bool FUSED, bool WITH_HD, bool FIXED = false>
__global__ __launch_bounds__(kBlock, 2) void ndt_strict3_kernel(const float4* const* __restrict__ src_ptrs, const int* __restrict__ src_sizes, NdtPair* __restrict__ pairs,
                                                               const VoxelGrid g, const VoxelStrictRec* __restrict__ vs, const double* __restrict__ vtab,
                                                               const double gauss_d1, const double gauss_d2, const int leaf_pow2, double* __restrict__ partials,
                                                               const int n_pairs, const int cap_blocks, int* __restrict__ pair_blocks, const NdtConsts consts,
                                                               int* __restrict__ done_flags, const int launch, const int solve_min_active, const int speculate) {
  using TL = StrictTile<SEARCH, WITH_HD>;
  constexpr int NB = TL::NB;
  int pair, slice, blocks_per_pair;
  int n_active = 0;
  // speculate: the first n_pairs workgroups of the grid are SOLVER workgroups, one per pair -- dispatched first, so that the exact Newton step
  // of a pair whose evaluation was published from the speculated direction (NdtPair::spec_s) runs beside the derivative work from the very
  // start of the launch; the remaining workgroups are dealt to the pairs of the round as ever.  A solver workgroup takes a ticket like a slice.
  const int n_solvers = (FUSED && speculate) ? n_pairs : 0;
  const bool solver = FUSED && (int)blockIdx.x < n_solvers;   // (FUSED spelled out: the unfused kernels carry no solver role, tests/test_isa_handoff.py)
  auto in_round = [&](int pi) {
    return FUSED ? (launch <= pairs[pi].serve[0] || launch == pairs[pi].serve[2]) : (pairs[pi].active != 0 && (WITH_HD || pairs[pi].need_hessian != 2));
  };
  if (solver) {
    pair = (int)blockIdx.x;
    if (!in_round(pair) || !pairs[pair].spec_pending) return;
    slice = -1;
    const int lane_id = threadIdx.x & 63;
    for (int c0 = 0; c0 < n_pairs; c0 += 64) n_active += __popcll(__ballot(c0 + lane_id < n_pairs && in_round(c0 + lane_id)));
    blocks_per_pair = max(1, min(((int)gridDim.x - n_solvers) / n_active, cap_blocks));   // as deal_workgroup derives it
  } else {
    if (!deal_workgroup(n_pairs, cap_blocks, in_round, pair, slice, blocks_per_pair, &n_active, (int)blockIdx.x - n_solvers, (int)gridDim.x - n_solvers)) return;
  }
  const NdtPair& st = pairs[pair];
  const float4* __restrict__ src = src_ptrs[pair];
  const int n = src_sizes[pair];
  // rows of the pair: the workgroups it was dealt in THIS launch (default), or -- dgs_handle::ndt_fixed_slices -- a function of its own size
  const int n_slices = FIXED ? strict_slices_of(n, cap_blocks) : blocks_per_pair;   // (FIXED: its own instantiation -- as a run-time switch the default path lost 5 %)
  if (!solver && slice == 0 && threadIdx.x == 0) pair_blocks[pair] = n_slices;
  const int kind = WITH_HD ? st.need_hessian : (st.need_hessian != 0 ? 1 : 0);
  const float gd2 = (float)gauss_d2;
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = st.T[k];
  double acc[kStrictAccum];
#pragma unroll
  for (int k = 0; k < kStrictAccum; k++) acc[k] = 0.0;

  __shared__ __attribute__((aligned(16))) unsigned char s_tab[kBlock / kWave][TL::kTableBytes];
  __shared__ unsigned s_queue[kBlock / kWave][TL::kQueue];
  // glibc's 2^(i/32) table in LDS: the item loop looks it up per lane (an LDS read instead of a gather through the vector cache).  Measured
  // alternatives on the bench step, same box: one copy per wave written by the wave itself, no workgroup barrier: 5.45-5.46 ms against 5.37-5.39
  // for this shared copy; the double pass's 2 x 128 table in LDS as well: 5.37 / 3.342 against 5.38 / 3.342 ms (bench step / 8 x 200,000
  // points) for constant memory, where it therefore stays.
  __shared__ unsigned long long s_exp2f[32];
  if (threadIdx.x < 32) s_exp2f[threadIdx.x] = kGlibcExp2fTab[threadIdx.x];
  __syncthreads();
  const unsigned long long* __restrict__ exptab = s_exp2f;   // this kernel serves consts.exp_libm = 1 only (strict_kernel_version, ndt_align.hip)
  const unsigned long long* __restrict__ exptab_d = kGlibcExpTab;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* tf = reinterpret_cast<float*>(s_tab[wave]);            // float kinds: [26][PTS]
  double* td = reinterpret_cast<double*>(s_tab[wave]);          // double pass: [23][PTS_HD] doubles, then [3][PTS_HD] floats
  float* tdx = reinterpret_cast<float*>(s_tab[wave] + 23 * 8 * TL::PTS_HD);
  unsigned* queue = s_queue[wave];
  // wave-uniform.  (Keeping it opaque to the optimiser -- asm volatile("" : "+s"(pts)) -- takes the kernel from 256 VGPRs + 4 spilled to 239
  // without spills, and the step from 6.25 to 6.6 ms: the loops specialised per tile size are worth more than the registers.)
  const int pts = (WITH_HD && kind == 2) ? TL::PTS_HD : TL::PTS;
  if (solver && wave == 0) {
    // ---- the deferred exact Newton step (NdtPair::spec_s).  The evaluation this launch computes for the pair was published from the
    // Gauss-Jordan direction; this wave takes the step the way the reference does -- JacobiSVD(H).solve(-g), the same begin_iteration code,
    // every lane alike -- on a copy of the optimiser state, writing the header it yields into LDS, and compares: 82 words, bit for bit.
    __shared__ NdtSolver s_x;
    NdtPair* scratch = reinterpret_cast<NdtPair*>(s_tab[0]);   // this wave's table region: free until its first tile
    constexpr int kWords = (int)(sizeof(NdtSolver) / 8);
    for (int w = lane; w < kWords; w += kWave) reinterpret_cast<double*>(&s_x)[w] = reinterpret_cast<const double*>(&st.s)[w];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const bool queued = begin_iteration<true, false, false>(scratch, scratch, s_x, consts, lane == 0);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    bool same = queued;
    const int* pub = reinterpret_cast<const int*>(&st);            // T[12], jang[8][3], hang[15][3], need_hessian: the first 82 words
    const int* mine = reinterpret_cast<const int*>(scratch);
    for (int w = lane; w < 82; w += kWave) same = same && (pub[w] == mine[w]);
    same = __all(same) != 0;
    for (int w = lane; w < kWords; w += kWave)
      __hip_atomic_store(reinterpret_cast<double*>(&pairs[pair].spec_s) + w, reinterpret_cast<const double*>(&s_x)[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) __hip_atomic_store(&pairs[pair].spec_result, same ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_wave_barrier();
  }
  const int subs = pts / 64;
  const int stride = n_slices * kBlock;
  static_assert(TL::kTableBytes >= kStrictRowScratch * 8, "the wave's table region doubles as its reduction scratch");
  // this workgroup's slices q = slice, slice + blocks_per_pair, ...: each its own sums and its own row
  __shared__ double s_cols[FIXED ? kStrictSlots : 1][FIXED ? kBlock / kWave : 1][FIXED ? kStrictPad : 1];
  static_assert(kStrictSlots * kStrictPad <= kBlock, "one thread per entry of the rows written together");
  int slot = 0, q_first = slice;
#pragma unroll 1
  for (int q = solver ? n_slices : slice; FIXED ? (q < n_slices) : (q == slice && !solver); q += blocks_per_pair) {
#pragma unroll
  for (int k = 0; k < kStrictAccum; k++) acc[k] = 0.0;
  // the wave's points of slice q: i = first + lane + sub * stride, tiles of `subs` strides
#pragma unroll 1
  for (int first = q * kBlock + wave * 64; first < n; first += subs * stride) {
    int qn = 0;
#pragma unroll 1
    for (int sub = 0; sub < subs; sub++) {
      const int i = first + sub * stride + lane;
      const int slot = sub * 64 + lane;
      unsigned mask = 0;
      int vids[NB];
      if (i < n) {
        const float4 x = src[i];
        float xt[3];
        xt[0] = affine_row_rn(T[0], T[1], T[2], T[3], x.x, x.y, x.z);
        xt[1] = affine_row_rn(T[4], T[5], T[6], T[7], x.x, x.y, x.z);
        xt[2] = affine_row_rn(T[8], T[9], T[10], T[11], x.x, x.y, x.z);
        mask = strict_neighbourhood<SEARCH>(xt, g, leaf_pow2, vids);
        if (mask) {
          if (WITH_HD && kind == 2) {
            double xj[8], xh[15];
            strict_point_tables_hd(x, st, xj, xh);
#pragma unroll
            for (int f = 0; f < 8; f++) td[f * TL::PTS_HD + slot] = xj[f];
#pragma unroll
            for (int f = 0; f < 15; f++) td[(8 + f) * TL::PTS_HD + slot] = xh[f];
#pragma unroll
            for (int f = 0; f < 3; f++) tdx[f * TL::PTS_HD + slot] = xt[f];
          } else {
            float xj[8], xh[15];
            if (kind == 1) strict_point_tables<true>(x, st, xj, xh);
            else strict_point_tables<false>(x, st, xj, xh);
#pragma unroll
            for (int f = 0; f < 3; f++) tf[f * TL::PTS + slot] = xt[f];
#pragma unroll
            for (int f = 0; f < 8; f++) tf[(3 + f) * TL::PTS + slot] = xj[f];
            if (kind == 1) {
#pragma unroll
              for (int f = 0; f < 15; f++) tf[(11 + f) * TL::PTS + slot] = xh[f];
            }
          }
        }
      }
      // append this sub-tile's items, slot-major: positions from ballots
#pragma unroll
      for (int k = 0; k < NB; k++) {
        const bool has = (mask >> k) & 1u;
        const unsigned long long b = __ballot(has);
        if (has) queue[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u))] = ((unsigned)slot << 25) | (unsigned)vids[k];
        qn += __popcll(b);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    // ---- the items, 64 at a time.  The float kinds run strict_items_float, a loop specialised per kind (the default, DGS_STRICT_ITEMS = 2); the
    // loop below serves the double pass -- and, in the A/B build DGS_STRICT_ITEMS = 0, all three kinds with the kind tested per round: that form
    // has 7 spilled registers instead of 25 and is 1 % SLOWER on the bench step (5.35 against 5.27-5.30 ms, same box, two runs each).
#ifndef DGS_STRICT_ITEMS
#define DGS_STRICT_ITEMS 2
#endif
#if DGS_STRICT_ITEMS != 0
    if (!(WITH_HD && kind == 2)) {
      if (kind == 1) strict_items_float<true, TL::PTS>(tf, queue, qn, lane, vs, gauss_d1, gd2, acc, exptab);
      else strict_items_float<false, TL::PTS>(tf, queue, qn, lane, vs, gauss_d1, gd2, acc, exptab);
      qn = 0;
    }
#endif
#pragma unroll 1
    for (int h = 0; h < qn; h += 64) {
      const int idx = h + lane;
      if (idx < qn) {
        const unsigned entry = queue[idx];
        const int slot = (int)(entry >> 25), vid = (int)(entry & 0x1FFFFFFu);
        if (WITH_HD && kind == 2) {
          double xj[8], xh[15];
          float xt[3];
#pragma unroll
          for (int f = 0; f < 8; f++) xj[f] = td[f * TL::PTS_HD + slot];
#pragma unroll
          for (int f = 0; f < 15; f++) xh[f] = td[(8 + f) * TL::PTS_HD + slot];
#pragma unroll
          for (int f = 0; f < 3; f++) xt[f] = tdx[f * TL::PTS_HD + slot];
          (void)strict_item_hd<false>(xt, xj, xh, vtab + (size_t)vid * 12, gauss_d1, gauss_d2, acc, nullptr, 0, exptab_d);
        } else {
          float xt[3], xj[8], xh[15];
#pragma unroll
          for (int f = 0; f < 3; f++) xt[f] = tf[f * TL::PTS + slot];
#pragma unroll
          for (int f = 0; f < 8; f++) xj[f] = tf[(3 + f) * TL::PTS + slot];
          if (kind == 1) {
#pragma unroll
            for (int f = 0; f < 15; f++) xh[f] = tf[(11 + f) * TL::PTS + slot];
            strict_item<true, true>(xt, xj, xh, vs + vid, gauss_d1, gd2, acc, exptab);
          } else {
#pragma unroll
            for (int f = 0; f < 15; f++) xh[f] = 0.f;
            strict_item<false, true>(xt, xj, xh, vs + vid, gauss_d1, gd2, acc, exptab);
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();   // the next tile overwrites the tables and the queue (LDS operations of one wave stay in order)
  }
  if (FIXED) {
    strict_wave_cols(acc, kind ? kStrictAccum : 7, td, s_cols[slot][wave]);
    slot++;
    if (slot == kStrictSlots || q + blocks_per_pair >= n_slices) {   // workgroup-uniform
      strict_rows_flush<FUSED>(reinterpret_cast<const double (*)[kBlock / kWave][kStrictPad]>(&s_cols[0][0][0]), slot, kind ? kStrictAccum : 7,
                               partials + (size_t)pair * cap_blocks * kStrictPad, q_first, blocks_per_pair);
      slot = 0;
      q_first = q + blocks_per_pair;
    }
  } else {
    strict_block_row<FUSED, false>(acc, kind ? kStrictAccum : 7, partials + ((size_t)pair * cap_blocks + slice) * kStrictPad, td);
  }
  }
  if (!FUSED) return;
  __shared__ int s_last;
  if (threadIdx.x < kStrictPad) handoff_drain_stores();   // (solver: the exact state's write-through stores)
  __syncthreads();
  // tickets of the pair in this launch: its slices, and its solver workgroup when a speculated step is pending (read before anybody can close)
  const int n_tickets = blocks_per_pair + ((speculate && st.spec_pending) ? 1 : 0);
  if (threadIdx.x == 0) s_last = handoff_take_ticket(&pairs[pair].ticket, n_tickets) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
#ifdef DGS_CLOSE_STAMPS
  if (threadIdx.x == 0 && pairs[pair].s.nr_iterations == 1) pairs[pair].traj[kTrajCap - 1][5] = (double)wall_clock64();
#endif
  // with enough other pairs to keep the chip busy the Newton step (a ~40 us dependent chain on one wave) leaves the launch: solve_min_active > 0
  ndt_close_strict<false, WITH_HD>(pairs + pair, partials + (size_t)pair * cap_blocks * kStrictPad, n_slices, consts, done_flags + pair, launch, 1,
                                   solve_min_active > 0 && n_active >= solve_min_active, speculate != 0);
}

// The Newton steps that the closings of round `launch` left behind (NdtPair::serve[1] == launch, phase PH_SOLVE_PENDING): one wave per pair,
// on its own stream beside the next round's derivative launch.  The pair re-enters the derivative launches at round launch + lag.
__global__ __launch_bounds__(kWave) void ndt_strict_solve_kernel(NdtPair* __restrict__ pairs, const int n_pairs, const NdtConsts c, int* __restrict__ done_flags, const int launch,
                                                               const int lag) {
  const int pair = blockIdx.x;
  if (pair >= n_pairs) return;
  NdtPair* st = pairs + pair;
  if (st->serve[1] != launch || st->s.phase != PH_SOLVE_PENDING) return;
  __shared__ NdtSolver s_lds;
  NdtSolver& s = s_lds;
  constexpr int kWords = (int)(sizeof(NdtSolver) / 8);
  for (int w = threadIdx.x; w < kWords; w += kWave) reinterpret_cast<double*>(&s_lds)[w] = reinterpret_cast<const double*>(&st->s)[w];
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  __syncthreads();
  const bool writer = threadIdx.x == 0;
  ndt_advance<false, false, true>(st, st, s, c, writer, false);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  for (int w = threadIdx.x; w < kWords; w += kWave) reinterpret_cast<double*>(&st->s)[w] = reinterpret_cast<const double*>(&s_lds)[w];
  if (writer) {
    if (s.phase == PH_DONE) {
      st->active = 0;
      st->last_launch = launch;
      __hip_atomic_store(done_flags + pair, launch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      st->serve[2] = launch + lag;
    }
  }
}
