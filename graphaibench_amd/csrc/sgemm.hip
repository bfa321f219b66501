// sgemm.hip -- fp32 dense update on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// replaces matmul -> sgemm_gpu -> cublasSgemm (src/utilities/math_functions.cu:321-343;
// CPU: matmul -> cblas_sgemm, math_functions.cpp:142-171):
//     row-major  C[M x N] (=|+=) op(A)[M x K] . op(B)[K x N]
// The three shapes on the GNN layer path:
//     NN  forward        X[N_v x din]  . W[din x dout]
//     NT  input gradient G[N_v x dout] . W[din x dout]^T
//     TN  weight grad    X[N_v x din]^T . G[N_v x dout]      (K = N_v: split-K, fixed-order reduce)
//
// f32-input MFMA is exact fp32 (a k-ordered fmaf chain, one rounding per product) at the
// fp32 vector peak (157 TF); there is no xf32/TF32 on gfx950, and the north-star tolerance
// (1e-4 vs the OpenMP path) rules out bf16 operands.
//
// Kernel: 256 threads = 4 waves; each wave owns WM x WN tiles of 32x32; operands are staged
// global -> registers (16-B loads, prefetched one K-step ahead) -> LDS.  LDS images:
//   "x-major" [rows][BK+4]  (operand stored with k contiguous): fragments by ds_read_b128,
//             conflict-free at row stride 36 floats;
//   "k-major" [BK][cols+4]  (operand stored with k as the slow index): fragments by 4 ds_read_b32.
// Lane (i = l&31, h = l>>5) of MFMA t in k-group kk consumes k = kk*8 + 4h + t for BOTH
// operands (the sum over k is order-free as long as A and B agree).
#include "common.h"

// sgemm_skinny.hip: *handled = 1 when the product was launched there
int gaib_sgemm_skinny_try(gaib_ctx* ctx, int transA, int transB, int64_t M, int64_t N, int64_t K, const float* d_A,
                          const float* d_B, int flags, float* d_C, int* handled);
int gaib_sgemm_wide_tn_try(gaib_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* d_A, float* d_G, const float* d_mask,
                           int flags, float* d_C, int* handled);

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
// a float4 whose address is only 4-byte aligned (rows of an odd width: 47 floats = 188 B): global loads of 16 B need dword
// alignment only
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int BK = 32;
constexpr int THREADS = 256;

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;        // output (or split-K partial slab base)
  int64_t M, N, K;
  int64_t k_chunk;  // K range per blockIdx.y (multiple of BK); == K when not split
  int accum;        // C += (ignored when writing split-K partials)
  int relu;         // clamp at 0 in the epilogue (ignored when writing split-K partials)
  int64_t slab;     // M*N when writing partials (C + blockIdx.y*slab), else 0
  int tiles_n;
  // BMASK kernels (weight gradient with d_relu folded in): op(B) = B where bmask > 0 else 0, and the masked
  // B is written back through bwrite (by the workgroups of the first M tile only)
  const float* bmask;
  float* bwrite;
  int interleave;  // register-resident TN kernel: waves take register sets round robin (a compact moving window over K)
};

// a dense product's work -- flops 2 M N K; bytes: both operands once, C written (and read when C +=),
// the mask read and the masked operand written back by the d_relu form
inline double gemm_flops(const GemmArgs& g) { return 2.0 * (double)g.M * (double)g.N * (double)g.K; }
struct GemmTag {  // "M x N x K" for the work table (the price of a product depends on its shape: a 47-wide output is no MFMA-bound job)
  char s[28];
  explicit GemmTag(const GemmArgs& g) { snprintf(s, sizeof(s), "%lldx%lldx%lld", (long long)g.M, (long long)g.N, (long long)g.K); }
};
inline double gemm_bytes(const GemmArgs& g, int accum) {
  return 4.0 * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N * (accum ? 2.0 : 1.0) +
                (g.bmask ? 2.0 * (double)g.K * g.N : 0.0));
}

// Load a TR x TC tile (TC % 4 == 0) of a row-major matrix [R][Cc] (leading dim ld) starting at
// (r0, c0) into registers; element (r, c) outside the matrix reads as 0.
template <int TR, int TC, bool VEC4>
__device__ __forceinline__ void tile_load(const float* __restrict__ src, int64_t R, int64_t Cc,
                                          int64_t ld, int64_t r0, int64_t c0,
                                          f4 (&regs)[TR * TC / 4 / THREADS]) {
  constexpr int CPR = TC / 4;  // float4 chunks per tile row
  constexpr int NCH = TR * TC / 4 / THREADS;
#pragma unroll
  for (int s = 0; s < NCH; ++s) {
    const int q = threadIdx.x + s * THREADS;
    const int r = q / CPR, cq = q % CPR;
    const int64_t gr = r0 + r, gc = c0 + cq * 4;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (gr < R) {
      const float* p = src + gr * ld + gc;
      if (VEC4 && gc + 3 < Cc) {
        // (f4u: 4-byte alignment is all a 16-byte global load needs -- rows of an odd width, 47 or 1433 floats, are loaded
        // with the same instruction as aligned ones; round 5: they used to take four 4-byte loads each)
        v = *reinterpret_cast<const f4u*>(p);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (gc + e < Cc) v[e] = p[e];
      }
    }
    regs[s] = v;
  }
}

// the mirror of tile_load: registers -> the same TR x TC window of a row-major matrix
template <int TR, int TC, bool VEC4>
__device__ __forceinline__ void tile_store_global(float* __restrict__ dst, int64_t R, int64_t Cc, int64_t ld,
                                                  int64_t r0, int64_t c0,
                                                  const f4 (&regs)[TR * TC / 4 / THREADS]) {
  constexpr int CPR = TC / 4;
  constexpr int NCH = TR * TC / 4 / THREADS;
#pragma unroll
  for (int s = 0; s < NCH; ++s) {
    const int q = threadIdx.x + s * THREADS;
    const int r = q / CPR, cq = q % CPR;
    const int64_t gr = r0 + r, gc = c0 + cq * 4;
    if (gr < R) {
      float* p = dst + gr * ld + gc;
      if (VEC4 && gc + 3 < Cc) {
        *reinterpret_cast<f4u*>(p) = regs[s];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (gc + e < Cc) p[e] = regs[s][e];
      }
    }
  }
}

template <int TR, int TC>
__device__ __forceinline__ void tile_store_lds(float* lds, const f4 (&regs)[TR * TC / 4 / THREADS]) {
  constexpr int CPR = TC / 4;
  constexpr int NCH = TR * TC / 4 / THREADS;
  constexpr int LD = TC + 4;
#pragma unroll
  for (int s = 0; s < NCH; ++s) {
    const int q = threadIdx.x + s * THREADS;
    const int r = q / CPR, cq = q % CPR;
    *reinterpret_cast<f4*>(&lds[r * LD + cq * 4]) = regs[s];
  }
}

// WAVES_M x WAVES_N waves, each WM x WN tiles of 32x32.
// A_KMAJOR: op(A) is stored [K][M] (transA);  B_KMAJOR: op(B) is stored [K][N] (no transB).
// OCC: workgroups per CU the register allocation is held to -- an HONEST figure (round 6): the 256-row tilings (WAVES_M = 4)
// stage 32-40 floats of A per lane next to 32-64 accumulators and cannot live in the 168 registers three workgroups per CU leave
// a wave (hipcc either missed the target or spilled 47-61 registers trying, -Rpass-analysis=kernel-resource-usage); they are
// held to two.  The build runs with -Werror=pass-failed: a tiling that misses its target does not compile.
template <int WAVES_M, int OCC>
constexpr int honest_occ() { return (WAVES_M == 4 && OCC > 2) ? 2 : OCC; }

template <int WAVES_M, int WAVES_N, int WM, int WN, bool A_KMAJOR, bool B_KMAJOR, bool AVEC, bool BVEC, int OCC,
          bool BMASK = false, bool DB = false>
__global__ __launch_bounds__(THREADS, (honest_occ<WAVES_M, OCC>())) void sgemm_mfma_kernel(GemmArgs g) {
  static_assert(!BMASK || B_KMAJOR, "the B mask is implemented for row-major [K][N] B operands");
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int BM = WAVES_M * WM * 32;
  constexpr int BN = WAVES_N * WN * 32;
  constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK;
  constexpr int A_LD = A_KMAJOR ? BM + 4 : BK + 4;
  constexpr int B_LD = B_KMAJOR ? BN + 4 : BK + 4;
  constexpr int A_LDS = A_KMAJOR ? BK * A_LD : BM * A_LD;
  constexpr int B_LDS = B_KMAJOR ? BK * B_LD : BN * B_LD;
  // DB: two LDS images, one barrier per K-step (the next tile's LDS stores and global loads are issued before
  // the MFMAs of the current one); otherwise one image and two barriers
  __shared__ __attribute__((aligned(16))) float lds[(DB ? 2 : 1) * (A_LDS + B_LDS)];
  float* As = lds;
  float* Bs = lds + A_LDS;

  const int tile = blockIdx.x;
  const int64_t m0 = (int64_t)(tile / g.tiles_n) * BM;
  const int64_t n0 = (int64_t)(tile % g.tiles_n) * BN;
  const int64_t kbeg = (int64_t)blockIdx.y * g.k_chunk;
  const int64_t kend = (kbeg + g.k_chunk < g.K) ? kbeg + g.k_chunk : g.K;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 31, lh = lane >> 5;

  f16v acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  f4 ra[A_ELEMS / 4 / THREADS], rb[B_ELEMS / 4 / THREADS];

  auto load_tiles = [&](int64_t k0) {
    if constexpr (A_KMAJOR) tile_load<BK, BM, AVEC>(g.A, g.K, g.M, g.M, k0, m0, ra);
    else tile_load<BM, BK, AVEC>(g.A, g.M, g.K, g.K, m0, k0, ra);
    if constexpr (B_KMAJOR) tile_load<BK, BN, BVEC>(g.B, g.K, g.N, g.N, k0, n0, rb);
    else tile_load<BN, BK, BVEC>(g.B, g.N, g.K, g.K, n0, k0, rb);
    if constexpr (BMASK) {
      f4 rm[B_ELEMS / 4 / THREADS];
      tile_load<BK, BN, BVEC>(g.bmask, g.K, g.N, g.N, k0, n0, rm);
#pragma unroll
      for (int s = 0; s < B_ELEMS / 4 / THREADS; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) rb[s][e] = rm[s][e] > 0.f ? rb[s][e] : 0.f;  // d_relu (math_functions.cu:258-268)
      if (g.bwrite && m0 == 0) tile_store_global<BK, BN, BVEC>(g.bwrite, g.K, g.N, g.N, k0, n0, rb);
    }
  };
  auto store_tiles = [&]() {
    if constexpr (A_KMAJOR) tile_store_lds<BK, BM>(As, ra);
    else tile_store_lds<BM, BK>(As, ra);
    if constexpr (B_KMAJOR) tile_store_lds<BK, BN>(Bs, rb);
    else tile_store_lds<BN, BK>(Bs, rb);
  };

  // note: rows of the K range beyond kend must read as zero -> clamp through the R/Cc bound
  // (g.K is the true extent; a split's kend <= g.K and k_chunk % BK == 0, so a tile never
  // straddles two splits).
  auto compute_tile = [&](const float* As_, const float* Bs_) {
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float af[WM][4], bf[WN][4];
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int m = (wm * WM + a) * 32 + li;
        if constexpr (A_KMAJOR) {
#pragma unroll
          for (int t = 0; t < 4; ++t) af[a][t] = As_[(kk * 8 + 4 * lh + t) * A_LD + m];
        } else {
          const f4 v = *reinterpret_cast<const f4*>(&As_[m * A_LD + kk * 8 + 4 * lh]);
#pragma unroll
          for (int t = 0; t < 4; ++t) af[a][t] = v[t];
        }
      }
#pragma unroll
      for (int b = 0; b < WN; ++b) {
        const int n = (wn * WN + b) * 32 + li;
        if constexpr (B_KMAJOR) {
#pragma unroll
          for (int t = 0; t < 4; ++t) bf[b][t] = Bs_[(kk * 8 + 4 * lh + t) * B_LD + n];
        } else {
          const f4 v = *reinterpret_cast<const f4*>(&Bs_[n * B_LD + kk * 8 + 4 * lh]);
#pragma unroll
          for (int t = 0; t < 4; ++t) bf[b][t] = v[t];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
          for (int b = 0; b < WN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][t], bf[b][t], acc[a][b], 0, 0, 0);
    }
  };

  if constexpr (DB) {
    auto store_to = [&](int buf) {
      float* Ad = lds + buf * (A_LDS + B_LDS);
      float* Bd = Ad + A_LDS;
      if constexpr (A_KMAJOR) tile_store_lds<BK, BM>(Ad, ra);
      else tile_store_lds<BM, BK>(Ad, ra);
      if constexpr (B_KMAJOR) tile_store_lds<BK, BN>(Bd, rb);
      else tile_store_lds<BN, BK>(Bd, rb);
    };
    if (kbeg < kend) {
      load_tiles(kbeg);
      store_to(0);
      if (kbeg + BK < kend) load_tiles(kbeg + BK);
    }
    __syncthreads();
    int cur = 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
      if (k0 + BK < kend) {
        store_to(cur ^ 1);  // registers hold tile k0 + BK; image cur^1 was last read before the previous barrier
        if (k0 + 2 * BK < kend) load_tiles(k0 + 2 * BK);
      }
      const float* Ac = lds + cur * (A_LDS + B_LDS);
      compute_tile(Ac, Ac + A_LDS);
      __syncthreads();
      cur ^= 1;
    }
  } else {
    // note: rows of the K range beyond kend must read as zero -> clamp through the R/Cc bound
    // (g.K is the true extent; a split's kend <= g.K and k_chunk % BK == 0, so a tile never
    // straddles two splits).
    if (kbeg < kend) {
      load_tiles(kbeg);
      store_tiles();
    }
    __syncthreads();
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
      const bool more = k0 + BK < kend;
      if (more) load_tiles(k0 + BK);
      compute_tile(As, Bs);
      __syncthreads();
      if (more) {
        store_tiles();
        __syncthreads();
      }
    }
  }

  // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* C = g.C + (g.slab ? (int64_t)blockIdx.y * g.slab : 0);
  const bool accum = g.accum && !g.slab;
  const bool relu = g.relu && !g.slab;
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) {
      const int64_t nn = n0 + (wn * WN + b) * 32 + li;
      if (nn < g.N) {
        const int64_t mt = m0 + (wm * WM + a) * 32;
        if (mt + 32 <= g.M) {
          // whole tile: the 16 old values of the C += form are requested together (one dependent load per element,
          // behind a bounds branch each, cost the accumulating products of the SAGE layers ~0.5 ms per launch)
          float* pc = C + (mt + 4 * lh) * g.N + nn;
          float cv[16];
          if (accum) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cv[r] = pc[((r & 3) + 8 * (r >> 2)) * g.N];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = accum ? (cv[r] + acc[a][b][r]) : acc[a][b][r];
            if (relu && !(v > 0.f)) v = 0.f;
            pc[((r & 3) + 8 * (r >> 2)) * g.N] = v;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t mm = mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (mm < g.M) {
              float* p = C + mm * g.N + nn;
              float v = accum ? (*p + acc[a][b][r]) : acc[a][b][r];
              if (relu && !(v > 0.f)) v = 0.f;
              *p = v;
            }
          }
        }
      }
    }
}

// C[i] = (accum ? C[i] : 0) + sum_s partial[s][i], s in order (deterministic)
__global__ void splitk_reduce_kernel(int64_t n, int splits, const float* partial, int accum, int relu,
                                     float* C) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float s = accum ? C[i] : 0.f;
#pragma unroll 8
    for (int k = 0; k < splits; ++k) s += partial[(int64_t)k * n + i];
    C[i] = (relu && !(s > 0.f)) ? 0.f : s;
  }
}

// ---- weight gradient, register-resident split-K ------------------------------------------------------------
//   C[M x N] = A^T . B   for A [K x M], B [K x N] row-major, M, N <= 128 (multiples of 4), K = number of vertices.
// Both operands are k-major, so an MFMA operand fragment is a COALESCED global read: lane (i = l&31, h = l>>5) loads
// the float4 A[k+h][4i..4i+3] -- 32 lanes cover the 512-byte row -- and uses its four elements as the A fragments of
// the four 32-row tiles (tile t holds the rows m = 4i + t; the assignment of rows to tiles is free as long as the
// epilogue knows it).  B likewise.  No LDS, no barriers: every wave owns a contiguous K range, keeps the whole
// 128x128 output in 256 accumulator registers (one wave per SIMD, 512-register budget) and streams its rows through
// two register sets of TN_PD row pairs each: the next set's loads are issued before the current set's 128 MFMAs.
// Partials go to a slab per wave and are summed in fixed order by splitk_reduce_kernel.
// BMASK: B <- B where mask > 0 else 0 on the way (d_relu folded in), written back in place -- K ranges are
// disjoint, so every element is written exactly once.
// pairs of rows per register set: 8 plain, 4 with the mask rows travelling along (register budget)
template <bool BMASK> struct TnDepth { static constexpr int PD = BMASK ? 4 : 8; };

// Outputs of 129..256 rows / columns (the hidden width 256 of scripts/run-sage-products.sh): QM x QN waves of ONE
// workgroup form a team that owns the QM x QN 128x128 quadrants of the output and walks the SAME K schedule, each wave
// reading its 512-byte half of the A row and of the B row.  A line of A, B or the mask therefore leaves HBM once: its
// second reader sits on the same CU (L1) or at worst behind the same XCD's L2 -- the LDS kernel's 128-row tiles re-read
// every A and B / mask line from HBM twice at 256 x 256 (17.6 GB of traffic for 10 GB).  The masked
// B is written back by the waves of the first quadrant row only.  <1, 1> is the M, N <= 128 kernel unchanged.
// NUNAL (round 5: the weight gradient of a 47-wide output layer, 128 x 47 and 256 x 47 with K = 2.45 M, ran through the LDS-tiled
// kernel with 4-byte loads at 0.26 of its roof): N is no multiple of 4, so B's rows are not 16-B aligned and the last lane of a
// row would read past its end.  Rows are still 4-B aligned, which is all a 16-byte global load needs; the one lane whose four
// columns straddle the end of the row takes the four columns that END at the row's end instead (never past the matrix): which
// columns a lane carries is free in this kernel as long as the epilogue stores them where they belong, so that lane and its
// neighbour both produce the overlapping columns -- the same sums in the same order, stored twice.  Plain form only (an output
// layer has no activation, hence no mask).
// NB < 4 (round 6: the 47 output classes, N <= 32 NB or 64): the 4 i + t column map gives a 47-wide B twelve useful lanes of 32 in
// every one of its four tiles -- 16 MFMAs per row pair for what two tiles hold, 0.51 ms of matrix-core time at 128 x 47 x 2.45 M
// where the bytes need 0.29 (profiles/r06/gemm_narrow_pmc.json: 0.27 of the roof).  Here tile b of B holds the columns 32 b + i:
// a lane reads B[k][i] and B[k][32 + i] with two 4-byte loads (coalesced: 128 + 60 bytes of the 188-byte row), the row pair
// costs 4 x NB MFMAs, and the accumulators -- 128 registers instead of 256 -- leave room for two waves per SIMD.  Plain form.
template <bool BMASK, int QM, int QN, bool NUNAL = false, int NB = 4, bool NTS = false>
__global__ __launch_bounds__(256, (NB < 4 ? 2 : 1)) void sgemm_tn_reg_kernel(GemmArgs g) {
  static_assert(!(BMASK && NUNAL), "the masked form writes float4s back: N must be a multiple of 4 there");
  static_assert(NB == 4 || (!BMASK && !NUNAL && QN == 1), "the narrow-B form: plain, one column quadrant");
  constexpr int TN_PD = NB < 4 ? 4 : TnDepth<BMASK>::PD;  // (narrow B: two waves per SIMD overlap each other -- half the prefetch depth, no spills)
  constexpr int TS = QM * QN;  // waves per team (1, 2 or 4); 4 / TS teams per workgroup
  static_assert(TS == 1 || TS == 2 || TS == 4, "a team is 1, 2 or 4 waves of one workgroup");
  const int lane = threadIdx.x & 63;
  const int i = lane & 31, h = lane >> 5;
  const int wv = threadIdx.x >> 6;
  const int qm = (wv % TS) / QN, qn = (wv % TS) % QN;  // this wave's quadrant
  const int64_t wid = (int64_t)blockIdx.x * (4 / TS) + wv / TS;  // team id: position in the K schedule, partial slab
  constexpr int64_t GRP = 2 * TN_PD;  // rows per register set
  // Two ways to deal K out.  Contiguous: wave w owns rows [w * k_chunk, (w + 1) * k_chunk).  Interleaved: set q of
  // 2 * TN_PD rows goes to wave q mod W, so at any time the W waves sweep one compact window of the operands (W sets =
  // a few MB) instead of W places spread over 1.25 GB each.  Below, k and kp are positions inside the wave's own
  // sequence of rows; phys() turns the start of a set into its row in the matrices.
  const int64_t W = (int64_t)gridDim.x * (4 / TS);
  int64_t kbeg, kend;
  if (g.interleave) {
    const int64_t total = (g.K + GRP - 1) / GRP;
    const int64_t mine = wid < total ? (total - wid + W - 1) / W : 0;
    kbeg = 0;
    kend = mine * GRP;
    if (mine > 0 && wid == (total - 1) % W) kend = (mine - 1) * GRP + (g.K - (total - 1) * GRP);  // the short last set
  } else {
    kbeg = wid * g.k_chunk;  // k_chunk is a multiple of 2 * TN_PD
    kend = (kbeg + g.k_chunk < g.K) ? kbeg + g.k_chunk : g.K;
  }
  auto phys = [&](int64_t kp) -> int64_t { return g.interleave ? ((kp / GRP) * W + wid) * GRP : kp; };
  f16v acc[4][NB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int mcol = 128 * qm + 4 * i, ncol = 128 * qn + 4 * i;
  const bool mok = mcol < g.M, nok = ncol < g.N;  // (M, N multiples of 4 -- N unless NUNAL --: a lane's float4 is all in or all out)
  // NUNAL: columns ncol .. ncol + 3 with only 4 - nsh of them inside the row: the lane carries the columns nsh further left
  const int nsh = (NUNAL && nok && ncol + 4 > g.N) ? (int)(ncol + 4 - g.N) : 0;
  const int mo = mok ? mcol : 0, no = nok ? ncol - nsh : 0;
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f4 sa[2][TN_PD], sb[2][TN_PD];
  f4 sm[BMASK ? 2 : 1][BMASK ? TN_PD : 1];  // BMASK: the mask rows travel with the set and are applied when it is consumed
  // NB < 4: sb[set][s][b] = B[k][32 b + i] (columns past N: column 0 of the row -- a valid address, a tile column never stored)
  int ncl[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) ncl[b] = (b < NB && 32 * b + i < g.N) ? 32 * b + i : 0;

  // load the TN_PD row pairs starting at row kp into register set `set`.  full: every row exists -> no selects on the
  // loaded values (a select makes the compiler wait for each load where it is issued).  Lanes beyond M / N read
  // column 0: what they contribute lands in rows / columns of C that are never stored.
  auto fetch = [&](int set, int64_t kp, bool full) {
#pragma unroll
    for (int s = 0; s < TN_PD; ++s) {
      const int64_t k = kp + 2 * s + h;
      const bool kok = full || k < kend;
      const int64_t kr = kok ? phys(kp) + 2 * s + h : 0;
      f4 va = NTS ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(g.A + kr * g.M + mo)) : *reinterpret_cast<const f4*>(g.A + kr * g.M + mo);
      f4 vb;
      if constexpr (NB < 4) {
#pragma unroll
        for (int b = 0; b < NB; ++b) vb[b] = g.B[kr * g.N + ncl[b]];
      } else if constexpr (NUNAL) vb = *reinterpret_cast<const f4u*>(g.B + kr * g.N + no);
      else vb = NTS ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(g.B + kr * g.N + no)) : *reinterpret_cast<const f4*>(g.B + kr * g.N + no);
      if constexpr (BMASK) {
        f4 mk = NTS ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(g.bmask + kr * g.N + no)) : *reinterpret_cast<const f4*>(g.bmask + kr * g.N + no);
        if (!full) mk = kok ? mk : zero4;  // (a zero mask also keeps the row out of the write-back below)
        sm[set][s] = mk;
      }
      if (!full) {  // rows past the end of this wave's range must not contribute
        va = kok ? va : zero4;
        if constexpr (NB < 4) {  // (only the NB components that exist: the others are never read and must not cost registers)
#pragma unroll
          for (int b = 0; b < NB; ++b) vb[b] = kok ? vb[b] : 0.f;
        } else {
          vb = kok ? vb : zero4;
        }
      }
      sa[set][s] = va;
      sb[set][s] = vb;
    }
    __builtin_amdgcn_sched_barrier(0);  // all loads of the set are issued before the MFMAs that follow
  };
  // BMASK: d_relu on the set that is about to be consumed (its loads landed a whole set ago), masked rows written back
  auto apply_mask = [&](int set, int64_t kp, bool full) {
    if constexpr (BMASK) {
#pragma unroll
      for (int s = 0; s < TN_PD; ++s) {
        f4 vb = sb[set][s];
        const f4 mk = sm[set][s];
#pragma unroll
        for (int e = 0; e < 4; ++e) vb[e] = mk[e] > 0.f ? vb[e] : 0.f;  // d_relu (math_functions.cu:258-268)
        sb[set][s] = vb;
        const int64_t k = kp + 2 * s + h;
        if (nok && qm == 0 && (full || k < kend)) {
          if constexpr (NTS) __builtin_nontemporal_store(vb, reinterpret_cast<f4*>(g.bwrite + (phys(kp) + 2 * s + h) * g.N + no));
          else *reinterpret_cast<f4*>(g.bwrite + (phys(kp) + 2 * s + h) * g.N + no) = vb;
        }
      }
    }
  };
  auto compute = [&](int set) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < TN_PD; ++s)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[set][s][a], sb[set][s][b], acc[a][b], 0, 0, 0);
  };
  constexpr int64_t GROUP = 2 * TN_PD;  // rows per register set
  if (kbeg < kend) {
    int64_t kp = kbeg;
    const bool f0 = kp + GROUP <= kend;
    fetch(0, kp, f0);
    bool full0 = f0;
    // steady state: two full groups per trip, no predicates on the loads
    while (kp + 3 * GROUP <= kend) {
      fetch(1, kp + GROUP, true);
      apply_mask(0, kp, true);
      compute(0);
      fetch(0, kp + 2 * GROUP, true);
      apply_mask(1, kp + GROUP, true);
      compute(1);
      kp += 2 * GROUP;
      full0 = true;
    }
    // tail: set 0 holds the group at kp; at most two more (possibly partial) groups follow
    const bool more1 = kp + GROUP < kend, more2 = kp + 2 * GROUP < kend;
    const bool full1 = kp + 2 * GROUP <= kend;
    if (more1) fetch(1, kp + GROUP, full1);
    apply_mask(0, kp, full0 && kp + GROUP <= kend);
    compute(0);
    if (more1) {
      if (more2) fetch(0, kp + 2 * GROUP, false);
      apply_mask(1, kp + GROUP, full1);
      compute(1);
      if (more2) {
        apply_mask(0, kp + 2 * GROUP, false);
        compute(0);
      }
    }
  }
  // partial slab of this team.  C/D map of the 32x32 MFMA: col j = lane&31, row r_ = (r&3) + 8*(r>>2) + 4*(lane>>5);
  // tile (a, b) of quadrant (qm, qn) holds C[128*qm + 4*r_ + a][128*qn + 4*j + b]
  float* P = g.C + wid * g.slab;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      // (nsh: the columns this lane carries, see above; 0 unless NUNAL.  NB < 4: tile b holds the columns 32 b + j)
      const int64_t nn = NB < 4 ? (int64_t)(32 * b + i) : (int64_t)(128 * qn + 4 * i - nsh + b);
      if (nn < g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t mm = 128 * qm + 4 * ((r & 3) + 8 * (r >> 2) + 4 * h) + a;
          if (mm < g.M) P[mm * g.N + nn] = acc[a][b][r];
        }
      }
    }
}

// ---- weight gradient of the 129..256-wide layers through an LDS ring (round 3) -------------------------------------------
// The register-resident kernel prefetches ONE register set ahead: 8 rows (plain) or 4 row pairs (masked -- the mask rows
// need registers too) = 1.7-3.4 us of MFMA work, which under 5 TB/s of streaming load is about one memory latency; the
// masked 256 x 256 product ran at 0.60 of the matrix peak with it.  Here the prefetch does not live in registers: the four
// waves of a workgroup (one per SIMD, the accumulator tile of a 128 x 128 output quadrant each, as before) share a ring of
// STAGES stages in LDS, a stage = TG_R rows of A, G and the mask (256 floats each: one 1-KiB LDS-DMA piece per row and
// array, `global_load_lds_dwordx4`, no VGPR destination), filled STAGES - 1 stages (7-10 us) ahead.  Every A / G / mask
// line leaves HBM once and reaches the CU once.
// One wave per SIMD means nothing hides what a wave does between its MFMAs, so the loop is software-pipelined by hand:
// while the 64 MFMAs of stage t run out of one register set, the SAME wave -- in the shadow of those MFMAs -- waits for its
// own pieces of stage t + 1 (counted vmcnt: loads return in order, so "at most NG * (STAGES - 2) outstanding" means stage
// t + 1 has landed; outstanding stores can only lengthen the wait), passes the ONE raw s_barrier of the stage (everybody's
// pieces have landed, everybody has read stage t out of its slot), refills that slot with stage t + STAGES, reads stage
// t + 1's fragments with ds_read_b128 into the other register set (lane (i, h): row 2 s + h, columns 128 q + 4 i .. + 3:
// the register kernel's fragment shape, so the epilogue is shared), applies the mask and writes the masked G back (the two
// waves of a column half hold the same values: each writes half of the stage's row pairs).  First version without the
// interleave (all of that between the stages): 3.06 ms masked / 2.49 plain at 2.45 M x 256 x 256; interleaved and with the
// wave index scalar: 2.87-2.90 / 2.53.  Timing-only builds without the barrier, without the counted wait and without both
// ran the plain form in the same 2.52-2.53 ms, i.e. neither memory latency nor barrier skew is what is left; without
// the write-back the masked form took 2.75.
constexpr int TG_R = 8;  // rows per stage
template <bool BMASK> struct TgCfg {
  static constexpr int ARR = BMASK ? 3 : 2;
  static constexpr int STAGES = BMASK ? 6 : 8;               // 6 x 24 KB = 144 KB, 8 x 16 KB = 128 KB of the 160 KB
  static constexpr int NG = ARR * TG_R / 4;                  // LDS-DMA pieces per wave and stage
  static constexpr int STAGE_F = ARR * TG_R * 256;           // floats per stage
};

template <bool BMASK, bool NTS = false>  // NTS: non-temporal LDS-DMA pieces and write-back (every line is touched once)
__global__ __launch_bounds__(256, 1) void sgemm_tn_glds_kernel(GemmArgs g) {
  using Cfg = TgCfg<BMASK>;
  constexpr int STAGES = Cfg::STAGES, NG = Cfg::NG, STAGE_F = Cfg::STAGE_F;
  constexpr int NS = TG_R / 2;  // row pairs per stage
  extern __shared__ __attribute__((aligned(16))) float tg_ring[];  // [STAGES][ARR][TG_R][256]  (the ONLY LDS object)
  const int lane = threadIdx.x & 63;
  // the wave index as a SCALAR: everything derived from it (the rows a wave loads, its quadrant, the LDS-DMA destinations)
  // then is scalar arithmetic.  As a plain threadIdx.x >> 6 the compiler kept it per lane, and the 64-bit row addresses of
  // the six pieces cost ~25 vector instructions each -- more than fits in the shadow of the MFMAs they sit between
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int qm = wv >> 1, qn = wv & 1;  // output quadrant
  const int team = blockIdx.x, W = gridDim.x;
  const int total = (int)((g.K + TG_R - 1) / TG_R);                 // sets of TG_R rows, dealt round robin to the teams
  const int mine = team < total ? (total - team + W - 1) / W : 0;
  // loader role: lane l brings floats 4l .. 4l + 3 of a 256-float row image; columns past M / N re-read column 0 (never used)
  const int lca = 4 * lane < g.M ? 4 * lane : 0, lcb = 4 * lane < g.N ? 4 * lane : 0;
  // piece j (compile time) of stage t of this team -> slot t % STAGES.  Past the team's last set: its last row again
  // (a valid address; keeps the number of outstanding pieces constant, which is what the counted wait counts on)
  const float* pa = g.A + lca;
  const float* pb = g.B + lcb;
  const float* pm = BMASK ? g.bmask + lcb : nullptr;
  const int last_set = mine > 0 ? (mine - 1) * W + team : 0;
  auto issue_piece = [&](int t, int j) {
    const int set = t < mine ? t * W + team : last_set;
    float* slot = tg_ring + (t % STAGES) * STAGE_F;
    const int arr = j >> 1;  // pieces 2 arr, 2 arr + 1 of every wave belong to array arr
    const int row = 4 * (j & 1) + wv;
    int64_t k = (int64_t)set * TG_R + row;
    k = k < g.K ? k : g.K - 1;  // rows past K: a valid address, zeroed by the reader
    const float* src = arr == 0 ? pa + k * g.M : (arr == 1 ? pb + k * g.N : pm + k * g.N);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(slot + (arr * TG_R + row) * 256), 16, 0, NTS ? 2 : 0);
  };
  f16v acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int mcol = 128 * qm + 4 * i, ncol = 128 * qn + 4 * i;
  const bool nok = ncol < g.N;
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // fragment reads of stage t: pair s -> (fa[s], fb[s], mk[s]); finish(): mask, rows past K, write-back
  auto read_pair = [&](int t, int s, f4 (&fa)[NS], f4 (&fb)[NS], f4 (&mk)[NS]) {
    const float* st = tg_ring + (t % STAGES) * STAGE_F;
    const int row = 2 * s + h;
    fa[s] = *reinterpret_cast<const f4*>(st + (0 * TG_R + row) * 256 + mcol);
    fb[s] = *reinterpret_cast<const f4*>(st + (1 * TG_R + row) * 256 + ncol);
    if constexpr (BMASK) mk[s] = *reinterpret_cast<const f4*>(st + (2 * TG_R + row) * 256 + ncol);
  };
  auto finish_pair = [&](int t, int s, f4 (&fa)[NS], f4 (&fb)[NS], f4 (&mk)[NS]) {
    const int64_t k = ((int64_t)t * W + team) * TG_R + 2 * s + h;
    const bool live = t < mine && k < g.K;
    fa[s] = live ? fa[s] : zero4;  // rows past K (the matrix's last set) and stages past the team's end contribute nothing
    if constexpr (BMASK) {
#pragma unroll
      for (int e = 0; e < 4; ++e) fb[s][e] = mk[s][e] > 0.f ? fb[s][e] : 0.f;  // d_relu (math_functions.cu:258-268)
      if ((s >> 1) == qm && nok && live) {
        if constexpr (NTS) __builtin_nontemporal_store(fb[s], reinterpret_cast<f4*>(g.bwrite + k * g.N + ncol));
        else *reinterpret_cast<f4*>(g.bwrite + k * g.N + ncol) = fb[s];
      }
    }
  };
  // one stage: the 64 MFMAs of stage t out of (ca, cb), everything stage t + 1 needs in their shadow, into (na, nb)
  auto stage = [&](int t, f4 (&ca)[NS], f4 (&cb)[NS], f4 (&na)[NS], f4 (&nb)[NS]) {
    f4 mk[NS];
#pragma unroll
    for (int m = 0; m < 16 * NS; ++m) {
      const int s = m >> 4, a = (m >> 2) & 3, b = m & 3;
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[s][a], cb[s][b], acc[a][b], 0, 0, 0);
      if (m == 7) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG * (STAGES - 2)) : "memory");  // this wave's pieces of stage t + 1 have landed
        __builtin_amdgcn_s_barrier();  // ... and everybody else's; everybody has read stage t out of its slot
        __builtin_amdgcn_sched_barrier(0);
      }
      if (m >= 8 && m < 8 + 2 * NG && ((m - 8) & 1) == 0) {  // refill the slot of stage t, a piece every other MFMA
        issue_piece(t + STAGES, (m - 8) >> 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (m >= 24 && m < 24 + 2 * NS && ((m - 24) & 1) == 0) {
        read_pair(t + 1, (m - 24) >> 1, na, nb, mk);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (m >= 40 && m < 40 + 4 * NS && ((m - 40) & 3) == 0) {  // every fourth MFMA: a pair's selects and its store
        finish_pair(t + 1, (m - 40) >> 2, na, nb, mk);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  f4 fa0[NS], fb0[NS], fa1[NS], fb1[NS];
#pragma unroll
  for (int t = 0; t < STAGES - 1; ++t)
#pragma unroll
    for (int j = 0; j < NG; ++j) issue_piece(t, j);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG * (STAGES - 2)) : "memory");  // stage 0
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < NG; ++j) issue_piece(STAGES - 1, j);
  {
    f4 mk[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) read_pair(0, s, fa0, fb0, mk);
#pragma unroll
    for (int s = 0; s < NS; ++s) finish_pair(0, s, fa0, fb0, mk);
  }
  for (int t = 0; t < mine; t += 2) {  // (an odd count runs one stage of zeros: finish_pair zeroes stages past the end)
    stage(t, fa0, fb0, fa1, fb1);
    stage(t + 1, fa1, fb1, fa0, fb0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the pieces issued past the end must not outlive the workgroup's LDS
  // partial slab of this team.  Same C/D map as sgemm_tn_reg_kernel.
  float* P = g.C + team * g.slab;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int64_t nn = 128 * qn + 4 * i + b;
      if (nn < g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t mm = 128 * qm + 4 * ((r & 3) + 8 * (r >> 2) + 4 * h) + a;
          if (mm < g.M) P[mm * g.N + nn] = acc[a][b][r];
        }
      }
    }
}

// 129..256 x 129..256 weight gradients: the LDS-ring kernel (sgemm_variant 34 keeps the register-resident teams)
int launch_tn_glds(gaib_ctx* ctx, GemmArgs g) {
  const int64_t total = cdiv64(g.K, TG_R);
  const unsigned blocks = (unsigned)std::min<int64_t>(ctx->num_cus, total);
  float* Cout = g.C;
  const int accum = g.accum;
  g.slab = g.M * g.N;
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * (size_t)blocks * g.slab));
  g.C = (float*)ctx->ws;
  ProfScope ps(ctx, "sgemm", gemm_bytes(g, accum), gemm_flops(g), 0, GemmTag(g).s);
#define GAIB_TG(MASK_, NTS_)                                                                                           \
  do {                                                                                                                 \
    const size_t lds = sizeof(float) * (size_t)TgCfg<MASK_>::STAGES * TgCfg<MASK_>::STAGE_F;                           \
    GAIB_HIP(hipFuncSetAttribute((const void*)sgemm_tn_glds_kernel<MASK_, NTS_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 160 * 1024));                                                                         \
    sgemm_tn_glds_kernel<MASK_, NTS_><<<blocks, 256, lds, ctx->stream>>>(g);                                           \
  } while (0)
  // (round 6: non-temporal pieces and write-back, which took 0.09 ms off the register-resident masked form at 128 x 128, change
  // nothing here -- 11.23 vs 11.26 ms for the four products of the SAGE 256 layer step; sgemm_variant 28 runs them for the record)
  if (g.bmask && ctx->sgemm_variant == 28) GAIB_TG(true, true);
  else if (g.bmask) GAIB_TG(true, false);
  else GAIB_TG(false, false);
#undef GAIB_TG
  GAIB_LAUNCH_CHECK();
  const int64_t n = g.M * g.N;
  unsigned rg = (unsigned)(cdiv64(n, 256) < 1024 ? cdiv64(n, 256) : 1024);
  splitk_reduce_kernel<<<rg, 256, 0, ctx->stream>>>(n, (int)blocks, (const float*)ctx->ws, accum, g.relu, Cout);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

// M, N <= 256 weight gradients with a long K: one wave per SIMD of the whole chip, in teams of QM x QN quadrant waves
int launch_tn_reg(gaib_ctx* ctx, GemmArgs g) {
  // 256 x 256, measured at K = 2.45 M (scripts/tn256.py): masked -- LDS ring 2.87-2.90 ms, register teams 3.36, LDS-tiled
  // kernel 3.58; plain -- register teams 2.39-2.41 ms (0.85 of the matrix peak), LDS ring 2.53, LDS-tiled 2.74-2.80.  So the
  // ring where the mask rows would have halved the register prefetch, the registers elsewhere (sgemm_variant 34 / 35 force
  // the register teams / the ring for both forms).
  if (g.M > 128 && g.N > 128 && ctx->sgemm_variant != 34 && (g.bmask || ctx->sgemm_variant == 35)) return launch_tn_glds(ctx, g);
  const int qm = g.M > 128 ? 2 : 1, qn = g.N > 128 ? 2 : 1, ts = qm * qn;
  // narrow B (N <= 64, plain): two column tiles instead of four, two waves per SIMD (sgemm_variant 38: the four-tile form)
  const bool narrow = !g.bmask && g.N <= 64 && ctx->sgemm_variant != 38;
  const int64_t teams = (int64_t)ctx->num_cus * 4 / ts * (narrow ? 2 : 1);
  const int64_t group = 2 * (narrow ? 4 : (g.bmask ? TnDepth<true>::PD : TnDepth<false>::PD));  // rows per register set (TN_PD of the kernel)
  int64_t chunk = cdiv64(cdiv64(g.K, teams), group) * group;  // whole register sets per team
  int64_t active = cdiv64(g.K, chunk);             // teams that own rows
  // sgemm_variant 33: contiguous K ranges per team; default: sets dealt round robin
  g.interleave = ctx->sgemm_variant == 33 ? 0 : 1;
  if (g.interleave) active = std::min<int64_t>(teams, cdiv64(g.K, group));
  const unsigned blocks = (unsigned)cdiv64(active, 4 / ts);
  const int n_slabs = (int)blocks * (4 / ts);
  float* Cout = g.C;
  const int accum = g.accum;
  g.k_chunk = chunk;
  g.slab = g.M * g.N;
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * (size_t)n_slabs * g.slab));
  g.C = (float*)ctx->ws;
  {
    ProfScope ps(ctx, "sgemm", gemm_bytes(g, accum), gemm_flops(g), 0, GemmTag(g).s);
  // NTS (round 6): the masked form streams three arrays and writes one back, each touched once: non-temporal loads and stores
  // (1.00 -> 0.91 ms at 128 x 128 x 2.45 M inside the GCN layer step, four alternations; sgemm_variant 39 = without).  The
  // plain form is bound by the matrix cores: sgemm_variant 29 runs it non-temporal for the comparison.
#define GAIB_TN(QM_, QN_)                                                                          \
  do {                                                                                             \
    if (g.bmask && ctx->sgemm_variant != 39) sgemm_tn_reg_kernel<true, QM_, QN_, false, 4, true><<<blocks, 256, 0, ctx->stream>>>(g); \
    else if (g.bmask) sgemm_tn_reg_kernel<true, QM_, QN_><<<blocks, 256, 0, ctx->stream>>>(g);    \
    else if (g.N % 4 != 0) sgemm_tn_reg_kernel<false, QM_, QN_, true><<<blocks, 256, 0, ctx->stream>>>(g); \
    else if (ctx->sgemm_variant == 29) sgemm_tn_reg_kernel<false, QM_, QN_, false, 4, true><<<blocks, 256, 0, ctx->stream>>>(g); \
    else sgemm_tn_reg_kernel<false, QM_, QN_><<<blocks, 256, 0, ctx->stream>>>(g);                 \
  } while (0)
    if (narrow && g.N <= 32) {
      if (qm == 1) sgemm_tn_reg_kernel<false, 1, 1, false, 1><<<blocks, 256, 0, ctx->stream>>>(g);
      else sgemm_tn_reg_kernel<false, 2, 1, false, 1><<<blocks, 256, 0, ctx->stream>>>(g);
    } else if (narrow) {
      if (qm == 1) sgemm_tn_reg_kernel<false, 1, 1, false, 2><<<blocks, 256, 0, ctx->stream>>>(g);
      else sgemm_tn_reg_kernel<false, 2, 1, false, 2><<<blocks, 256, 0, ctx->stream>>>(g);
    } else if (qm == 1 && qn == 1) GAIB_TN(1, 1);
    else if (qm == 2 && qn == 2) GAIB_TN(2, 2);
    else if (qm == 1) GAIB_TN(1, 2);
    else GAIB_TN(2, 1);
#undef GAIB_TN
    GAIB_LAUNCH_CHECK();
    const int64_t n = g.M * g.N;
    unsigned rg = (unsigned)(cdiv64(n, 256) < 1024 ? cdiv64(n, 256) : 1024);
    splitk_reduce_kernel<<<rg, 256, 0, ctx->stream>>>(n, n_slabs, (const float*)ctx->ws, accum, g.relu, Cout);
    GAIB_LAUNCH_CHECK();
  }
  return GAIB_OK;
}

template <int WAVES_M, int WAVES_N, int WM, int WN, bool AK, bool BKM>
int launch(gaib_ctx* ctx, GemmArgs g, bool avec, bool bvec) {
  constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
  const int64_t tiles_m = cdiv64(g.M, BM), tiles_n = cdiv64(g.N, BN);
  g.tiles_n = (int)tiles_n;
  const int64_t tiles = tiles_m * tiles_n;
  GAIB_CHECK(tiles < (int64_t)1 << 31, "gaib_sgemm: too many tiles");
  // split K when the output has too few tiles to fill 256 CUs and K is long (weight grads)
  // Also for the small graphs (cora: 2 708 vertices, 1 433 features): without a split the products of a layer are a
  // handful of workgroups walking K one 32-column step after another, two barriers and a memory latency per step --
  // 214 + 181 + 120 us of a 690 us epoch (profiles/r02/cora_kernel_stats_before_split.csv).  From K = 512 on, K is
  // dealt out in pieces of at least two steps.
  int splits = 1;
  if (tiles < 2 * ctx->num_cus && g.K >= 512 && ctx->sgemm_variant != 50) {
    int64_t want = cdiv64(2 * (int64_t)ctx->num_cus, tiles);
    int64_t maxs = g.K >= 8192 ? g.K / (8 * BK) : g.K / (2 * BK);
    splits = (int)(want < maxs ? want : maxs);
    if (splits < 1) splits = 1;
  }
  float* Cout = g.C;
  const int accum = g.accum;
  if (splits > 1) {
    g.k_chunk = cdiv64(cdiv64(g.K, splits), BK) * BK;
    splits = (int)cdiv64(g.K, g.k_chunk);
    g.slab = g.M * g.N;
    GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * (size_t)splits * g.slab));
    g.C = (float*)ctx->ws;
  } else {
    g.k_chunk = cdiv64(g.K > 0 ? g.K : 1, BK) * BK;
    g.slab = 0;
  }
  dim3 grid((unsigned)tiles, (unsigned)splits);
  ProfScope ps(ctx, "sgemm", gemm_bytes(g, accum), gemm_flops(g), 0, GemmTag(g).s);
  // OCC = workgroups per CU the register allocation is held to (__launch_bounds__ 2nd argument):
  // more resident blocks hide the staging / epilogue phases of one block under another's MFMAs.
#define GAIB_GEMM_LAUNCH(AV, BV, OCC)                                                    \
  sgemm_mfma_kernel<WAVES_M, WAVES_N, WM, WN, AK, BKM, AV, BV, OCC><<<grid, THREADS, 0, ctx->stream>>>(g)
  // (sgemm_variant 2: two workgroups per CU; default three.  The four-per-CU and LDS-double-buffered experiments of rounds 1-2
  // are gone: the former spilled on every 128-column tiling, the latter's 70-90 KB of LDS held every tiling to one or two
  // workgroups per CU whatever was asked for -- 21 instantiations that missed their own occupancy target)
  const int occ = ctx->sgemm_variant == 2 ? 2 : 3;
  if constexpr (AK && BKM) {
    if (g.bmask) {  // (the entry point only takes this path with 16-B aligned operands)
      sgemm_mfma_kernel<WAVES_M, WAVES_N, WM, WN, AK, BKM, true, true, 3, true><<<grid, THREADS, 0, ctx->stream>>>(g);
      GAIB_LAUNCH_CHECK();
      if (splits > 1) {
        const int64_t n = g.M * g.N;
        unsigned rg = (unsigned)(cdiv64(n, 256) < 1024 ? cdiv64(n, 256) : 1024);
        splitk_reduce_kernel<<<rg, 256, 0, ctx->stream>>>(n, splits, (const float*)ctx->ws, accum, g.relu, Cout);
        GAIB_LAUNCH_CHECK();
      }
      return GAIB_OK;
    }
  }
  if (avec && bvec) {
    if (occ == 3) GAIB_GEMM_LAUNCH(true, true, 3);
    else GAIB_GEMM_LAUNCH(true, true, 2);
  } else if (avec) GAIB_GEMM_LAUNCH(true, false, 2);
  else if (bvec) GAIB_GEMM_LAUNCH(false, true, 2);
  else GAIB_GEMM_LAUNCH(false, false, 2);
#undef GAIB_GEMM_LAUNCH
  GAIB_LAUNCH_CHECK();
  if (splits > 1) {
    const int64_t n = g.M * g.N;
    unsigned rg = (unsigned)(cdiv64(n, 256) < 1024 ? cdiv64(n, 256) : 1024);
    splitk_reduce_kernel<<<rg, 256, 0, ctx->stream>>>(n, splits, (const float*)ctx->ws, accum, g.relu, Cout);
    GAIB_LAUNCH_CHECK();
  }
  return GAIB_OK;
}

template <bool AK, bool BKM>
int dispatch_shape(gaib_ctx* ctx, const GemmArgs& g, bool avec, bool bvec) {
  // experimental tilings (sgemm_variant 10..12): smaller per-wave tiles, more resident waves
  if (ctx->sgemm_variant == 10) return launch<2, 2, 2, 1, AK, BKM>(ctx, g, avec, bvec);  // 128 x 64
  if (ctx->sgemm_variant == 11) return launch<2, 2, 1, 2, AK, BKM>(ctx, g, avec, bvec);  // 64 x 128
  if (ctx->sgemm_variant == 12) return launch<2, 2, 1, 1, AK, BKM>(ctx, g, avec, bvec);  // 64 x 64
  if (g.N > 64) {
    // measured at 2.45 M x 128 x 128 (scripts/microbench.py): 64 x 128 blocks 0.86-0.87 ms vs 128 x 128 0.93-0.97 ms
    // for the streaming-A shapes; the split-K weight gradient keeps the square tile (0.84 ms)
    if (!AK && ctx->sgemm_variant != 13) return launch<2, 2, 1, 2, AK, BKM>(ctx, g, avec, bvec);  // 64 x 128
    return launch<2, 2, 2, 2, AK, BKM>(ctx, g, avec, bvec);  // 128 x 128
  }
  if (g.N > 32) return launch<4, 1, 2, 2, AK, BKM>(ctx, g, avec, bvec);  // 256 x 64
  return launch<4, 1, 2, 1, AK, BKM>(ctx, g, avec, bvec);                // 256 x 32
}

}  // namespace

extern "C" int gaib_sgemm(gaib_ctx* ctx, int transA, int transB, int64_t M, int64_t N, int64_t K,
                          const float* d_A, const float* d_B, int accum, float* d_C) {
  return gaib_sgemm_ex(ctx, transA, transB, M, N, K, d_A, d_B, accum ? GAIB_ACCUMULATE : 0, d_C);
}

// ---- streaming products C[M x N] = A[M x K] . op(B), M in the millions, K <= 256 ----------------------------------------
// The LDS-tiled kernel re-stages the same small B for every 64-row tile (9.8 GB of L2 -> LDS traffic for 2.45 M x 256 x
// 256 against 2.5 GB of A) and synchronises every 32 columns of K; every tiling of it lands at 103-106 TF/s.  Here a
// persistent workgroup keeps a 128-column slab of op(B) in LDS, k-major [K][128 + 4], for its whole life, and every wave
// streams its own 32-row tiles of A straight from global memory into MFMA operand registers: lane (i, h) reads the 16
// bytes A[row i][k0 + 4h .. k0 + 4h + 3] and pairs them, step by step, with B rows k0 + 4h + s from LDS (the two lane
// halves of v_mfma_f32_32x32x2_f32 may carry any two k's as long as both operands agree).  One float4 load feeds 16
// MFMAs; no barrier after the slab is staged.
namespace {
constexpr int NNP_WAVES = 8;
constexpr int NNP_LDB = 128 + 4;
template <int NT, bool BT>  // NT = 32-column MFMA tiles per wave (N slab = 32 * NT <= 128); BT: B is [N][K] (op = transpose)
__global__ __launch_bounds__(NNP_WAVES * 64) void sgemm_stream_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float nnp_lds[];  // [K][NNP_LDB]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: the tile loop and its bounds run on the SALU
  const int li = lane & 31, lh = lane >> 5;
  const int64_t n0 = (int64_t)blockIdx.y * (32 * NT);
  const int K = (int)g.K;
  const int Kfull = K & ~7;        // whole 8-column steps; K % 8 == 4 (the 100 input features of the products shape)
  const int Kpad = (K + 7) & ~7;   // leaves a half step: its upper four B rows are zero and lane half 1 feeds zeros
  // stage the slab: Bs[k][n] = op(B)[k][n0 + n], zero past N and past K
  for (int t = threadIdx.x; t < Kpad * 32 * NT; t += NNP_WAVES * 64) {
    int k, n;
    if constexpr (BT) { n = t / Kpad; k = t - n * Kpad; }  // coalesced along k of B[n][:]
    else { k = t / (32 * NT); n = t - k * (32 * NT); }
    const int64_t nn = n0 + n;
    float v = 0.f;
    if (nn < g.N && k < K) v = BT ? g.B[nn * g.K + k] : g.B[(int64_t)k * g.N + nn];
    nnp_lds[k * NNP_LDB + n] = v;
  }
  __syncthreads();
  const int64_t ntiles = (g.M + 31) / 32;
  const float* bcol = nnp_lds + li;
  for (int64_t t = (int64_t)blockIdx.x * NNP_WAVES + wave; t < ntiles; t += (int64_t)gridDim.x * NNP_WAVES) {
    const int64_t m0 = t * 32;
    int64_t row = m0 + li;
    if (row >= g.M) row = g.M - 1;  // clamped: loaded, never stored
    const float* arow = g.A + row * g.K + 4 * lh;
    f16v acc[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    // two operand sets: the load of step k0 + 8 is in flight while step k0 runs on the matrix cores.  (Measured at
    // 2.45 M x 256 x 256: this form 2.82 ms; 16 waves + two steps ahead 2.97 ms; 32-wide K blocks with whole-line
    // loads 2.94 ms; the LDS-tiled kernel 3.0-3.1 ms in every tiling; rocBLAS 2.63 ms.)
    // (the half step's operand: both lane halves read the row's last four columns -- inside the row -- and half 1 is
    // zeroed before use)
    const float* atail = g.A + row * g.K + Kfull;
    const float* atail_or_first = Kfull < K ? atail : arow;  // what the last whole step prefetches (the row's first operand again when there is no half step: a valid address, unused)
    f4 a_cur = *reinterpret_cast<const f4*>(Kfull ? arow : atail);
    for (int k0 = 0; k0 < Kfull; k0 += 8) {
      // the next step's operand is requested now and used after this step's 16 MFMAs.  One unconditional load from a
      // selected address: with the load under a branch (two of them once the half step came in) the compiler moved it
      // to the end of the step and waited for it at once -- 3.05 instead of 2.65 ms at 2.45 M x 256 x 256
      const float* pn = arow + k0 + 8;
      if (k0 + 8 >= Kfull) pn = atail_or_first;
      const f4 a_nxt = *reinterpret_cast<const f4*>(pn);
      __builtin_amdgcn_sched_barrier(0);  // (... and without this it sinks the load behind the step's MFMAs to save four registers)
      const float* bk = bcol + (k0 + 4 * lh) * NNP_LDB;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s], bk[s * NNP_LDB + 32 * b], acc[b], 0, 0, 0);
      }
      a_cur = a_nxt;
    }
    if (Kfull < K) {
      const float* bk = bcol + (Kfull + 4 * lh) * NNP_LDB;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float av = lh ? 0.f : a_cur[s];
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bk[s * NNP_LDB + 32 * b], acc[b], 0, 0, 0);
      }
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const bool full_rows = m0 + 32 <= g.M;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int64_t nn = n0 + 32 * b + li;
      if (nn < g.N) {
        float* pc = g.C + (m0 + 4 * lh) * g.N + nn;
        if (full_rows) {
          // whole tile: the 16 old values (C += form) are requested together, not one dependent load per element.
          // (Round 3 tried to request them one tile AHEAD and seed the accumulators with them: 4.24 vs 2.85 ms -- vmcnt
          // returns in order, so every wait for an A operand behind the 64 prefetch loads waits for them too.)
          float cv[16];
          if (g.accum) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cv[r] = pc[((r & 3) + 8 * (r >> 2)) * g.N];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = g.accum ? (cv[r] + acc[b][r]) : acc[b][r];
            if (g.relu && !(v > 0.f)) v = 0.f;
            pc[((r & 3) + 8 * (r >> 2)) * g.N] = v;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t mm = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (mm < g.M) {
              float* p = g.C + mm * g.N + nn;
              float v = g.accum ? (*p + acc[b][r]) : acc[b][r];
              if (g.relu && !(v > 0.f)) v = 0.f;
              *p = v;
            }
          }
        }
      }
    }
  }
}

}  // namespace

template <bool BT>
int launch_stream(gaib_ctx* ctx, const GemmArgs& g) {
  const int nt = g.N > 96 ? 4 : (g.N > 64 ? 3 : (g.N > 32 ? 2 : 1));
  const unsigned slabs = (unsigned)cdiv64(g.N, 32 * nt);
  const size_t lds = sizeof(float) * (size_t)((g.K + 7) & ~(int64_t)7) * NNP_LDB;
  const int64_t ntiles = cdiv64(g.M, 32);
  unsigned gx = (unsigned)std::min<int64_t>((int64_t)ctx->num_cus / slabs > 0 ? ctx->num_cus / slabs : 1, cdiv64(ntiles, NNP_WAVES));
  if (slabs == 1) gx = (unsigned)std::min<int64_t>(ctx->num_cus, cdiv64(ntiles, NNP_WAVES));
  ProfScope ps(ctx, "sgemm", gemm_bytes(g, g.accum), gemm_flops(g), 0, GemmTag(g).s);
#define GAIB_STREAM(NTT)                                                                                        \
  do {                                                                                                          \
    GAIB_HIP(hipFuncSetAttribute((const void*)sgemm_stream_kernel<NTT, BT>,                                     \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                      \
    sgemm_stream_kernel<NTT, BT><<<dim3(gx, slabs), NNP_WAVES * 64, lds, ctx->stream>>>(g);                     \
  } while (0)
  if (nt == 4) GAIB_STREAM(4);
  else if (nt == 3) GAIB_STREAM(3);
  else if (nt == 2) GAIB_STREAM(2);
  else GAIB_STREAM(1);
#undef GAIB_STREAM
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_sgemm_ex(gaib_ctx* ctx, int transA, int transB, int64_t M, int64_t N, int64_t K,
                             const float* d_A, const float* d_B, int flags, float* d_C) {
  const int accum = (flags & GAIB_ACCUMULATE) ? 1 : 0;
  GAIB_CHECK(ctx, "gaib_sgemm: ctx is NULL");
  GAIB_CHECK(M >= 0 && N >= 0 && K >= 0, "gaib_sgemm: negative dimension");
  if (M == 0 || N == 0) return GAIB_OK;
  GAIB_CHECK(d_C, "gaib_sgemm: C is NULL");
  GAIB_CHECK(K == 0 || (d_A && d_B), "gaib_sgemm: A/B is NULL");
  GAIB_HIP(hipSetDevice(ctx->device));
  if (K == 0) {
    if (!accum) return gaib_fill_f32(ctx, M * N, 0.f, d_C);
    if (flags & GAIB_RELU) return gaib_relu(ctx, M * N, d_C, d_C);
    return GAIB_OK;
  }
  if (transA && transB) {
    gaib_set_error("gaib_sgemm: transA && transB is not on the GNN path (unsupported)");
    return GAIB_ERR_UNSUPPORTED;
  }
  {  // round 6: the products whose output or inner width is the class / input feature count (sgemm_skinny.hip)
    int handled = 0;
    GAIB_TRY(gaib_sgemm_skinny_try(ctx, transA, transB, M, N, K, d_A, d_B, flags, d_C, &handled));
    if (handled) return GAIB_OK;
    if (transA && !transB && ctx->sgemm_variant != 67) {  // weight gradients with a 100- / 128-wide input side (wide_tn_kernel)
      GAIB_TRY(gaib_sgemm_wide_tn_try(ctx, M, N, K, d_A, const_cast<float*>(d_B), nullptr, flags, d_C, &handled));
      if (handled) return GAIB_OK;
    }
  }
  GemmArgs g;
  g.A = d_A;
  g.B = d_B;
  g.C = d_C;
  g.M = M;
  g.N = N;
  g.K = K;
  g.k_chunk = K;
  g.accum = accum ? 1 : 0;
  g.relu = (flags & GAIB_RELU) ? 1 : 0;
  g.slab = 0;
  g.tiles_n = 1;
  g.bmask = nullptr;
  g.bwrite = nullptr;
  g.interleave = 0;
  // 16-B loads need an aligned base and a leading dimension that keeps rows aligned
  const int64_t lda = transA ? M : K;
  const int64_t ldb = transB ? K : N;
  const bool avec = (((uintptr_t)d_A & 15) == 0) && (lda % 4 == 0);
  const bool bvec = (((uintptr_t)d_B & 15) == 0) && (ldb % 4 == 0);
  // weight gradients of the layer widths: register-resident split-K (sgemm_variant 30 keeps the LDS kernel)
  // (M, N <= 128: one wave per output; up to 256: teams of quadrant waves; sgemm_variant 32: only up to 128, the round-2 rule)
  const int64_t tn_max = ctx->sgemm_variant == 32 ? 128 : 256;
  // (N no multiple of 4 -- the 47-wide output layer: B's rows are 4-byte aligned only, the NUNAL form; sgemm_variant 36 keeps
  // the LDS-tiled kernel there)
  // measured (scripts/gemm_odd_width.py, K = 2.45 M): 128 x 47 0.82 -> 0.69-0.72 ms, 100 x 47 0.65; at 256 x 47 the quadrant
  // teams compute two 128 x 128 tiles for 47 columns and lose to the tiled kernel (1.22 vs 0.80 ms): M <= 128 only
  // round 6: N <= 64 takes the narrow-B form (4-byte loads of B: any N, any alignment of its rows; M up to 256)
  const bool b_rows_ok = (N % 4 == 0 && bvec) || (N <= 64 && ctx->sgemm_variant != 38 && ctx->sgemm_variant != 36) ||
                         (N % 4 != 0 && N >= 4 && M <= 128 && (((uintptr_t)d_B & 3) == 0) && ctx->sgemm_variant != 36);
  if (transA && !transB && M <= tn_max && N <= tn_max && M % 4 == 0 && avec && b_rows_ok && K >= 32768 &&
      ctx->sgemm_variant != 30)
    return launch_tn_reg(ctx, g);
  // streaming products (rows in the millions, 128 < K <= 256): persistent workgroups with the op(B) slab in LDS
  // (sgemm_variant 40 keeps the LDS-tiled kernel, 41 forces this one wherever the shape allows).  Inside the SAGE
  // 256 -> 256 layer step: 17.9 vs 19.5 ms for the six GEMMs (scripts/ab_gemm_in_layer.py); at K = 128 a tile is too
  // short for this form (1.00 vs 0.90 ms).
  const bool stream_shape = !transA && K % 4 == 0 && K >= 8 && K <= 256 && avec;
  // Round 2, measured again after the whole-tile C += epilogue (scripts/gemm_variants.py, 2.45 M rows): K = 128 NN 0.78-0.80
  // vs 0.89 ms for the LDS-tiled kernel (torch.mm / rocBLAS 0.98), K = 256 NN 2.63-2.66 vs 3.05-3.11 (rocBLAS 2.51-2.53),
  // so the streaming form is the default from K = 128 on (sgemm_variant 44: only above 128, the round-1 rule).
  // Tried on top of it and dropped: the next tile's first operand requested before the epilogue (no change: 2.63 vs
  // 2.65); 64-column operand groups in two register sets with the prefetch running on across tiles and the B
  // fragments double-buffered out of LDS (184 VGPRs, every wait in the ISA where it should be -- and 2.82 ms vs 2.68);
  // the second wave of every SIMD started half a tile late, so that the two waves' epilogues do not fall together
  // (2.63 vs 2.68 NN, nothing at NT or K = 128); 64-row wave tiles, so that every B fragment read from LDS feeds two
  // MFMAs (212 VGPRs, four MFMAs per ds_read2 in the listing): 2.97-3.00 ms vs 2.60-2.63.
  const int64_t kmin = ctx->sgemm_variant == 44 ? 129 : (ctx->sgemm_variant == 45 ? 128 : 96);  // (45: the rule before K = 100 was measured)
  const int sv = ctx->sgemm_variant;
  const bool auto_rule = sv == 0 || sv == 28 || sv == 29 || sv == 44 || sv == 45 || (sv >= 30 && sv <= 39) || (sv >= 60 && sv <= 67);  // (60 .. 67 switch sgemm_skinny.hip's family only)  // (30 .. 36 concern the weight gradient only, 37 the tiled kernel's loads)
  if (stream_shape && (sv == 41 || (auto_rule && M >= 65536 && K >= kmin)))
    return transB ? launch_stream<true>(ctx, g) : launch_stream<false>(ctx, g);
  // The LDS-tiled kernel loads a ROW-MAJOR A ([M][K], K odd: dX = G [N x 47] . W^T of the output layer) with 16-byte
  // instructions at 4-byte alignment (tile_load: f4u): 2.45 M x 256 x 47 1.21 -> 1.04 ms, x 128 x 47 0.63 -> 0.55
  // (scripts/gemm_odd_width.py; sgemm_variant 37: 4-byte loads, the rule until round 4).  NOT the k-major operands of a weight
  // gradient: the same loads on B [K][47] made the tiled 128 x 47 product 2.18 ms against 0.82 with 4-byte loads.
  const bool av4 = (transA || ctx->sgemm_variant == 37) ? avec : (((uintptr_t)d_A & 3) == 0);
  const bool bv4 = bvec;
  if (!transA && !transB) return dispatch_shape<false, true>(ctx, g, av4, bv4);
  if (!transA && transB) return dispatch_shape<false, false>(ctx, g, av4, bv4);
  return dispatch_shape<true, true>(ctx, g, av4, bv4);
}

// weight gradient with the layer's d_relu folded in (graph_conv_layer backward: d_relu_gpu on grad_in, then
// matmul(transA) -- gcn_layer.cpp:33-52):  G <- G where mask > 0 else 0 (in place), C (=|+=) A^T . G.
// A is [K x M], G and mask are [K x N].  One pass over G instead of d_relu's read-modify-write plus the GEMM's read.
extern "C" int gaib_sgemm_drelu(gaib_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* d_A, float* d_G,
                                const float* d_mask, int accum, float* d_C) {
  GAIB_CHECK(ctx, "gaib_sgemm_drelu: ctx is NULL");
  GAIB_CHECK(M >= 0 && N >= 0 && K >= 0, "gaib_sgemm_drelu: negative dimension");
  GAIB_CHECK(K == 0 || N == 0 || (d_G && d_mask), "gaib_sgemm_drelu: G/mask is NULL");
  const bool aligned = ((((uintptr_t)d_A | (uintptr_t)d_G | (uintptr_t)d_mask) & 15) == 0) && M % 4 == 0 && N % 4 == 0;
  // long K, M, N <= 128: the register-resident split-K kernel (mask rows travel with the operand sets and are applied
  // when a set is consumed); sgemm_variant 30 / 31 keep the LDS kernel.  With K dealt out in contiguous per-wave ranges
  // this kernel won alone (1.03 vs 1.16 ms) and lost inside the layer step (1.15 vs 1.10); with the register sets dealt
  // round robin (one compact window over K for all waves) it wins there too: 0.97 vs 1.03 ms (scripts/ab_weight_grad.py).
  const int64_t tn_max = ctx->sgemm_variant == 32 ? 128 : 256;
  const bool reg_path = aligned && M <= tn_max && N <= tn_max && K >= 32768 && ctx->sgemm_variant != 30 && ctx->sgemm_variant != 31;
  if (M == 0 || N == 0 || K == 0 || (!reg_path && (!aligned || N <= 64))) {
    // shapes the masked kernel is not built for: the two-step form
    if (K > 0 && N > 0) GAIB_TRY(gaib_d_relu(ctx, K * N, d_G, d_mask, d_G));
    return gaib_sgemm(ctx, 1, 0, M, N, K, d_A, d_G, accum, d_C);
  }
  GAIB_CHECK(d_C && d_A, "gaib_sgemm_drelu: A/C is NULL");
  if (ctx->sgemm_variant != 67) {  // round 6: 16 x 16 tiles, 7 of them for a 100-wide input, a deeper ring (sgemm_skinny.hip)
    int handled = 0;
    GAIB_TRY(gaib_sgemm_wide_tn_try(ctx, M, N, K, d_A, d_G, d_mask, accum ? GAIB_ACCUMULATE : 0, d_C, &handled));
    if (handled) return GAIB_OK;
  }
  GAIB_HIP(hipSetDevice(ctx->device));
  GemmArgs g;
  g.A = d_A;
  g.B = d_G;
  g.C = d_C;
  g.M = M;
  g.N = N;
  g.K = K;
  g.k_chunk = K;
  g.accum = accum ? 1 : 0;
  g.relu = 0;
  g.slab = 0;
  g.tiles_n = 1;
  g.bmask = d_mask;
  g.bwrite = d_G;
  g.interleave = 0;
  if (reg_path) return launch_tn_reg(ctx, g);
  return launch<2, 2, 2, 2, true, true>(ctx, g, true, true);  // 128 x 128 split-K tile, as the plain weight gradient
}
