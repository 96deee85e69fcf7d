// Layout of the small per-level device array holding the FGMRES Hessenberg data (K = max iterations).
#pragma once
struct HsLayout {
  int K;
  int beta, tt, cs, sn, grs, y, hd, total;
  __host__ __device__ explicit HsLayout(int K_) : K(K_) {
    beta = 0;                 // |r0|
    tt = 1;                   // current |w|
    cs = 8 + K * (K + 1);     // Givens cosines (K)
    sn = cs + K;              // Givens sines (K)
    grs = sn + K;             // rotated rhs (K+1)
    y = grs + K + 1;          // least-squares solution (K)
    hd = y + K;               // CGS dot products of the current column (K+1)
    total = hd + K + 1;
  }
  __host__ __device__ int H(int j) const { return 8 + j * (K + 1); }  // column j of the Hessenberg (K+1 entries)
};

#ifdef __HIPCC__
// ---- one-thread pieces of the FGMRES smoother, used by the kernels of kernels_vec.hip -------------------------------
// column j of the Hessenberg: h[0..j] = the CGS dots, h[j+1] = tt = sqrt(sum of nblocks partials); Givens update
// (KSPFGMRESUpdateHessenberg [3P]).  Partitioned levels pass the all-reduced values (nblocks = 1).
// ww != nullptr (partitioned levels, one all-reduce per iteration): |w_new|^2 = |w|^2 - sum_i h_i^2 with the all-reduced
// *ww = |w|^2 (w before the projection) and h; V is orthonormal, so this is the same number up to O(eps |w|^2).
__device__ inline double pythagoras_norm2(const double* ww, const double* h, int j) {
  double s = *ww;
  for (int i = 0; i <= j; ++i) s -= h[i] * h[i];
  return s > 0.0 ? s : 0.0;
}

// one thread: column j of the Hessenberg from the CGS dots h[0..j] and tt = |w_new|, Givens update, rotated rhs
__device__ inline void hessenberg_column(double* __restrict__ hs, int K, int j, const double* __restrict__ h,
                                                  double tt) {
  HsLayout L(K);
  hs[L.tt] = tt;
  double* hcol = hs + L.H(j);
  for (int i = 0; i <= j; ++i) hcol[i] = h[i];
  hcol[j + 1] = tt;
  double* cs = hs + L.cs;
  double* sn = hs + L.sn;
  double* grs = hs + L.grs;
  for (int i = 0; i < j; ++i) {
    const double t = hcol[i];
    hcol[i] = cs[i] * t + sn[i] * hcol[i + 1];
    hcol[i + 1] = -sn[i] * t + cs[i] * hcol[i + 1];
  }
  const double den = hypot(hcol[j], hcol[j + 1]);
  if (den != 0.0) {
    cs[j] = hcol[j] / den;
    sn[j] = hcol[j + 1] / den;
  } else {
    cs[j] = 1.0;
    sn[j] = 0.0;
  }
  grs[j + 1] = -sn[j] * grs[j];
  grs[j] = cs[j] * grs[j];
  hcol[j] = den;
  hcol[j + 1] = 0.0;
}


// back substitution on the triangularised Hessenberg (KSPFGMRESBuildSoln [3P]) -> y
__device__ inline void fgmres_back_substitution(double* __restrict__ hs, int k, int K) {
  HsLayout L(K);
  double* y = hs + L.y;
  const double* grs = hs + L.grs;
  for (int i = k - 1; i >= 0; --i) {
    double s = grs[i];
    for (int q = i + 1; q < k; ++q) s -= hs[L.H(q) + i] * y[q];
    const double d = hs[L.H(i) + i];
    y[i] = d != 0.0 ? s / d : 0.0;
  }
}
#endif
