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
