// sgemm_skinny.hip -- the dense products of a layer whose output or inner width is the class count or the input feature
// count (47 and 100 on the products shape): matmul -> sgemm_gpu -> cublasSgemm (src/utilities/math_functions.cu:321-343; CPU:
// matmul -> cblas_sgemm, math_functions.cpp:142-171) at   rows x 47 x {64, 128, 256},  rows x {64, 128, 256} x 47 (NT),
// {128, 256} x 47 x rows (TN),  rows x 128 x 100  and  rows x 256 x 100 (two column slabs).
//
// These are HBM-stream shapes (<= 17 flop / B, the fp32 MFMA / HBM ridge is at 19.6): both operands leave HBM once, the output
// is written once, and the matrix cores need 0.6-0.9 of the time the bytes need -- IF a tile wastes no columns.  Round 5 ran
// them through kernels built for 128-wide outputs: 32 x 32 tiles (a 47-wide output pays for 64 columns: 27 % of the
// matrix-core time), operands requested ONE 8-column step (200 ns of MFMA work) ahead of their use, i.e. a memory latency
// exposed at every new 128-B line -- 2.1-3.5 TB/s (profiles/r06/gemm_narrow_pmc.json "before": traffic = 1.0-1.15 x the
// algorithmic bytes, SQ_WAIT_INST_ANY 35-60 % of the wave time: not over-fetch, under-issue).  Here:
//   * v_mfma_f32_16x16x4_f32: 47 columns cost 48;
//   * the SMALL operand (W or its transpose, <= 128 x 48 / 48 x 128 / 100 x 128 ... 256 x 48) lives in REGISTERS for the life
//     of a persistent wave, laid out as MFMA operand fragments: the inner loop reads no LDS and has no barrier;
//   * the STREAMED operand comes through buffer loads -- one descriptor per row tile, loop-invariant lane offsets, no 64-bit
//     address arithmetic, rows past the matrix read as zeros -- into a RING of register tiles: NBUF - 1 tiles (3-16 KB each)
//     are in flight while one runs on the matrix cores, and a round of NBUF steps is straight-line code so that the
//     compiler's vmcnt waits are exact (with an exit test between the steps they drained the ring);
//   * row-stream kernel: the product is computed TRANSPOSED (D^T = op(B)^T . A^T), so that a lane ends up with four consecutive
//     COLUMNS of one output row: 16-byte stores, one per 16 x 16 tile instead of four 4-byte ones; a 47-wide output leaves
//     through a compact LDS image as whole 128-B lines.
// Measured at 2.45 M rows (profiles/r06/gemm_narrow_pmc.json "after", profiles/r06/gemm_skinny/): 0.31-0.37 ms for the
// 128 x 47 shapes (round 5: 0.49-0.62), 0.60-0.72 for 256 x 47 (0.88-1.17), 0.57 / 1.12 for x 128 / 256 x 100 (0.71 / 1.45):
// 4.1-5.6 TB/s against an in-run stream copy of 6.0, traffic 1.00-1.05 x the algorithmic bytes.
// f32-input MFMA is a k-ordered chain of exact fp32 fmas; which k a lane pair carries is free as long as both operands agree,
// which is what lets a lane load 16 contiguous bytes of a row and use them in four consecutive MFMAs.
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // a 16-byte access at 4-byte alignment (rows of 47 floats)
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u3_t __attribute__((ext_vector_type(3)));

struct SkinnyArgs {
  const float* A;
  const float* B;
  float* C;
  int64_t M;  // row-stream: rows of A and C.  TN: rows of A and B (the long K)
  int N, K;   // row-stream: C is [M x N], A is [M x K].  TN: C is [K x N] = A[M x K]^T . B[M x N]
  int accum, relu;
  float* slabs;  // TN: per-workgroup partial outputs
  int ldn;       // row-stream, direct stores: row stride of C and of a row-major B when blockIdx.y walks column slabs of N (else == N)
};

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---- row stream:  C[M x N] (=|+=) A[M x K] . op(B),  M in the millions, N <= 16 CT, K = 16 G + 4 E ---------------------------
// Lane (j = l & 15, q = l >> 4).  Chunk c of a row = its columns 4c .. 4c + 3; in group g lane q takes chunk 4g + q with one
// 16-byte load, and the four MFMA steps (g, t) pair element t of it with op(B)[4 (4g + q) + t][:] -- held in registers.  E: one
// more step whose lanes take ONE column each (k = 16 G + q; K = 100 = 6 groups + 1 step).  A chunk that would straddle the end
// of a row (K = 47: chunk 11) starts 4 - K % 4 columns further left instead; the columns it then shares with its neighbour
// meet zeros in the B registers.  Chunks past the row likewise re-read the row's first chunk against zeros.  No load ever
// leaves its row.
// MFMA operands: a = op(B) fragment (i = column 16 ct + j of the output), b = A fragment (j = row of the tile)
//   => D[i][j] in lane (j, q), register r = C[row j][16 ct + 4 q + r].
// Epilogue, two forms, both with a FIXED number of memory instructions per tile (a store under a branch makes the compiler wait
// for vmcnt(0) at the top of the loop, i.e. for the previous tile's stores to be acknowledged before the next loads go out):
//   LDSOUT (N <= 48, any N): the 16 rows of a tile are 16 N CONTIGUOUS floats of C.  The wave drops its D fragments into a
//     compact [16][N] image in LDS (its own 16 x 48 floats, no barrier) and copies the image out -- or adds it to what is
//     there (ACC) -- with 16-byte accesses of whole lines; the ragged column edge never reaches a global store;
//   direct (N % 4 == 0): lane (j, q) stores C[row j][16 ct + 4 q .. + 3] from its registers, one 16-byte store per column tile.
template <int G, int E, int CT, int RT, bool BT, int WPS, bool LDSOUT, bool ACC, int NBUF>
__global__ __launch_bounds__(256, WPS) void skinny_rows_kernel(SkinnyArgs g) {
  __shared__ __attribute__((aligned(16))) float out_lds[LDSOUT ? 4 * (16 * 16 * CT + 4) : 4];  // per wave: image + scrap word
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int K = g.K, N = g.N;  // N: the columns of THIS workgroup's slab (== g.ldn unless the launch has column slabs)
  const int ldn = LDSOUT ? N : g.ldn, n0 = LDSOUT ? 0 : (int)blockIdx.y * N;
  const int K4 = (K + 3) >> 2;
  int kb[G > 0 ? G : 1];
  float breg[G > 0 ? G : 1][4][CT];
  float bx[E ? CT : 1];
#pragma unroll
  for (int gg = 0; gg < G; ++gg) {
    const int c = 4 * gg + q;
    const bool valid = c < K4;
    const int base = valid ? (4 * c + 4 <= K ? 4 * c : K - 4) : 0;
    kb[gg] = base;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = base + t;
      const bool live = valid && k >= 4 * c;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int n = 16 * ct + j;
        const bool on = live && n < N;  // (unconditional load from a clamped address + select: no branch per register)
        const int nn = on ? n : 0, kk = on ? k : 0;
        const float v = BT ? g.B[(int64_t)(n0 + nn) * K + kk] : g.B[(int64_t)kk * ldn + n0 + nn];
        breg[gg][t][ct] = on ? v : 0.f;
      }
    }
  }
  if constexpr (E) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int n = 16 * ct + j, k = 16 * G + q;
      const bool on = n < N && k < K;
      const int nn = on ? n : 0, kk = on ? k : 0;
      const float v = BT ? g.B[(int64_t)(n0 + nn) * K + kk] : g.B[(int64_t)kk * ldn + n0 + nn];
      bx[ct] = on ? v : 0.f;
    }
  }
  const int kx = (16 * G + q < K) ? 16 * G + q : 0;  // E: this lane's single column (past the row: column 0 against a zero)

  constexpr int ROWS = 16 * RT;
  const int64_t ntiles = (g.M + ROWS - 1) / ROWS;
  const int64_t W = (int64_t)gridDim.x * 4;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: the tile loop and its bounds run on the SALU
  const int64_t wid = (int64_t)blockIdx.x * 4 + wv;
  float* img = out_lds + (LDSOUT ? wv * (16 * 16 * CT + 4) : 0);  // this wave's [16][N] image
  const int64_t cend = g.M * N;
  const bool relu = g.relu != 0;
  f4 buf[NBUF][RT][G > 0 ? G : 1];  // the ring: NBUF - 1 tiles in flight while one runs on the matrix cores
  float bufx[NBUF][RT];

  // The streamed operand comes through buffer loads: one descriptor per tile (its base = the tile's first row: scalar
  // arithmetic), the lane's byte offsets inside the tile are loop invariants -- no 64-bit address is computed per load, and
  // rows past the end of the matrix (the last, partial tile) read as zeros instead of faulting.
  int voff[RT][G > 0 ? G : 1];
  int voffx[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
    for (int gg = 0; gg < G; ++gg) voff[rt][gg] = 4 * ((16 * rt + j) * K + kb[gg]);
    voffx[rt] = 4 * ((16 * rt + j) * K + kx);
  }
  const int64_t a_bytes = g.M * (int64_t)K * 4;
  auto fetch = [&](int s, int64_t tile) {
    const int64_t off = tile * ROWS * (int64_t)K * 4;
    const int64_t left = a_bytes - off;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const char*>(g.A) + off), 0, (int)(left < (1 << 30) ? left : (1 << 30)), 0x00020000);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int gg = 0; gg < G; ++gg) {
        const u4_t r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[rt][gg], 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) buf[s][rt][gg][e] = __uint_as_float(r[e]);
      }
      if constexpr (E) bufx[s][rt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voffx[rt], 0, 0));
    }
    __builtin_amdgcn_sched_barrier(0);  // every load of the tile is issued before the MFMAs that follow
  };
  // TAIL: the matrix's last, partial tile (at most one, after the loop): rows and pieces past the end are predicated there.  The
  // loop's tiles are whole: every store is unconditional.
  auto compute = [&](int s, int64_t tile, auto tailc) {
    constexpr bool TAIL = decltype(tailc)::value;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int64_t row0 = tile * ROWS + 16 * rt;
      // C +=: the tile's old values are requested HERE, behind the next tile's operand loads and before this tile's MFMAs, and
      // used after them (3-7 us later).  Requested where they are added they are the youngest loads of the wave: the wait for
      // them drains the ring at every tile (2.45 M x 256 x 256 C += inside the SAGE epoch: 3.07 ms per launch against 2.36 for C =)
      constexpr int NP = LDSOUT ? (16 * 16 * CT / 4 + 63) / 64 : CT;  // 16-byte pieces of old C per lane
      f4 old[ACC ? NP : 1];
      if constexpr (ACC && !TAIL) {
        if constexpr (LDSOUT) {
          const float* cb0 = g.C + row0 * N;
          const int last = 16 * N - 4;
#pragma unroll
          for (int x = 0; x < NP; ++x) {
            const int f = 4 * (lane + 64 * x);
            old[x] = *reinterpret_cast<const f4*>(cb0 + (f < last ? f : last));
          }
        } else {
          const float* pc0 = g.C + (row0 + j) * ldn + n0 + 4 * q;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) old[ct] = *reinterpret_cast<const f4*>(pc0 + 16 * ct);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      f4 acc[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int gg = 0; gg < G; ++gg)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) acc[ct] = mfma16(breg[gg][t][ct], buf[s][rt][gg][t], acc[ct]);
      if constexpr (E) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = mfma16(bx[ct], bufx[s][rt], acc[ct]);
      }
      if constexpr (LDSOUT) {
        // columns past N (the ragged edge of the last column tile) go to a scrap word behind the image: the write itself is
        // unconditional
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = 16 * ct + 4 * q + e;
            img[n < N ? j * N + n : 16 * 16 * CT] = acc[ct][e];
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the image = C[row0 * N .. (row0 + 16) * N), 16-byte pieces (row0 % 16 == 0: the base is 64-byte aligned).  A lane
        // whose piece lies past the image takes the image's last piece again: the same bytes to the same place.
        float* cb = g.C + row0 * N;
        if constexpr (!TAIL) {
          const int last = 16 * N - 4;
#pragma unroll
          for (int x = 0; x < NP; ++x) {
            int f = 4 * (lane + 64 * x);
            f = f < last ? f : last;
            f4 v = *reinterpret_cast<const f4*>(img + f);
            if constexpr (ACC) v = old[x] + v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (relu && !(v[e] > 0.f)) ? 0.f : v[e];
            *reinterpret_cast<f4*>(cb + f) = v;
          }
        } else {
          const int64_t left = cend - row0 * N;
          const int npc = (int)(left < 0 ? 0 : (left < 16 * N ? left : 16 * N));  // floats of this tile that exist
          for (int f = lane; f < npc; f += 64) {
            float v = img[f];
            if constexpr (ACC) v = cb[f] + v;
            cb[f] = (relu && !(v > 0.f)) ? 0.f : v;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the image is rewritten by the next tile only after every lane has read it
      } else {
        // (N == 16 CT: no column edge)
        const int64_t row = row0 + j;
        const bool rok = !TAIL || row < g.M;
        float* pc = g.C + (rok ? row : 0) * ldn + n0 + 4 * q;
        if constexpr (ACC && TAIL) {  // (the one partial tile: rows past the end were pointed at row 0)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) old[ct] = *reinterpret_cast<const f4*>(pc + 16 * ct);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          f4 v = acc[ct];
          if constexpr (ACC) v = old[ct] + v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (relu && !(v[e] > 0.f)) ? 0.f : v[e];
          if constexpr (TAIL) {
            if (rok) *reinterpret_cast<f4*>(pc + 16 * ct) = v;
          } else {
            *reinterpret_cast<f4*>(pc + 16 * ct) = v;
          }
        }
      }
    }
  };
  // Whole tiles dealt round robin: at any time the waves of the chip sweep one compact window of A and C.  This wave's tiles
  // are wid + x W, x = 0 .. nw - 1, taken in ROUNDS of NBUF through the ring: step b of a round requests the tile NBUF - 1
  // ahead into the slot the previous step emptied and runs slot b.  The round is straight-line code -- no exit test between its
  // steps: with one, the compiler's count of outstanding loads at every merge is the minimum over both paths, and the waits it
  // emits drain the ring (measured in the ISA: vmcnt(12) where 36 are allowed).  A request past the wave's last tile takes that
  // last tile again (out of the cache, never consumed); the nw % NBUF tiles after the last whole round are already in their
  // slots.  vmcnt counts 63 at most: (NBUF - 1) (G + E) RT loads + the stores of NBUF - 1 tiles stay below that.
  const int64_t nfull = g.M / ROWS;
  const int64_t nw = wid < nfull ? (nfull - wid + W - 1) / W : 0;
  auto tile_of = [&](int64_t x) { return wid + (x < nw ? x : nw - 1) * W; };
  if (nw > 0) {
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) fetch(b, tile_of(b));
    int64_t x = 0;
    for (; x + NBUF <= nw; x += NBUF) {
#pragma unroll
      for (int b = 0; b < NBUF; ++b) {
        fetch((b + NBUF - 1) % NBUF, tile_of(x + b + NBUF - 1));
        compute(b, wid + (x + b) * W, std::false_type{});
      }
    }
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b)
      if (x + b < nw) compute(b, wid + (x + b) * W, std::false_type{});
  }
  if (nfull < ntiles && wid == nfull % W) {  // the partial tile
    fetch(0, nfull);
    compute(0, nfull, std::true_type{});
  }
}

// ---- TN:  C[MW x N] (=|+=) A[K x MW]^T . B[K x N],  K (= g.M here) in the millions, MW <= 128 NH (a multiple of 4), N <= 16 CT
// -- the weight gradient of the output layer.  Both operands are k-major, so an MFMA fragment is a coalesced global read: lane
// (i = l & 15, q = l >> 4) loads the float4 A[k + q][64 h + 4 i .. + 3] (16 lanes cover 256 contiguous bytes) and uses element
// e as the A fragment of row tile (h, e) -- the tile that holds the rows m = 64 h + 4 i + e: which rows a tile holds is free as
// long as the epilogue knows.  B: lane (j, q) loads B[k + q][16 c + j] for each of the CT column tiles.  A step = 4 rows of k =
// 8 CT MFMAs.  A wave keeps a 128-row half of the output in 32 CT accumulator registers (two waves per SIMD) and walks its own
// sets of 4 S rows (dealt round robin over the waves: one compact window over K), the next set in flight while the current one
// runs on the matrix cores.  NH = 2 (129..256 output rows): waves w and w + 1 of a workgroup take the two halves of the SAME
// sets -- the second reader of a B line sits on the same CU.  The waves of a workgroup add their outputs in LDS in wave order;
// one slab per workgroup goes to memory and the slabs are summed in fixed order by skinny_reduce_kernel.
// B (round 6b): ONE 12-byte load per row instead of three 4-byte ones -- lane j takes the columns 3 j .. 3 j + 2 (16 lanes cover
// 48 columns: 192 contiguous bytes), element c is the lane's column of tile c.  A lane whose three columns would cross the end
// of the row (N = 47: lane 15) starts sh = 3 j + 3 - N columns further left; the columns it then shares with its neighbours
// are computed twice (same operands, same order) and stored by the neighbour only.
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
template <int NH, int S, int WPS, int NBUF>
__global__ __launch_bounds__(256, WPS) void skinny_tn_kernel(SkinnyArgs g) {
  constexpr int CT = 3;
  extern __shared__ __attribute__((aligned(16))) float tn_lds[];  // [128 NH][16 CT]
  const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: the tile loop and its bounds run on the SALU
  const int MW = g.K, N = g.N;  // output rows / columns
  const int64_t KR = g.M;       // rows of A and B
  constexpr int SR = 4 * S;     // rows per set
  constexpr int TPW = 4 / NH;   // K schedules (teams) per workgroup
  const int half = wv % NH;
  const int64_t W = (int64_t)gridDim.x * TPW;
  const int64_t wid = (int64_t)blockIdx.x * TPW + wv / NH;
  const int64_t nsets = (KR + SR - 1) / SR;
  int mo[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) mo[h] = (128 * half + 64 * h + 4 * i < MW) ? 128 * half + 64 * h + 4 * i : 0;  // (MW % 4 == 0)
  const int sh = 3 * i + 3 > N ? 3 * i + 3 - N : 0;  // (N >= 3)
  const int no = 3 * i - sh;                         // this lane's first column of B
  f4 acc[2][4][CT];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[h][e][c] = f4{0.f, 0.f, 0.f, 0.f};
  f4 sa[NBUF][S][2];
  typedef float f3 __attribute__((ext_vector_type(3)));
  f3 sb[NBUF][S];
  // Buffer loads: one descriptor pair per set (base = the set's first row: scalar arithmetic), the lane's byte offsets inside
  // the set are loop invariants; rows past the end of the matrices (the last, partial set) read as zeros.
  int voa[S][2], vob[S];
#pragma unroll
  for (int st = 0; st < S; ++st) {
#pragma unroll
    for (int h = 0; h < 2; ++h) voa[st][h] = 4 * ((4 * st + q) * MW + mo[h]);
    vob[st] = 4 * ((4 * st + q) * N + no);
  }
  const int64_t a_bytes = KR * (int64_t)MW * 4, b_bytes = KR * (int64_t)N * 4;
  auto fetch = [&](int set, int64_t s) {
    const int64_t offa = s * SR * (int64_t)MW * 4, offb = s * SR * (int64_t)N * 4;
    const int64_t la = a_bytes - offa, lb = b_bytes - offb;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const char*>(g.A) + offa), 0, (int)(la < (1 << 30) ? la : (1 << 30)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const char*>(g.B) + offb), 0, (int)(lb < (1 << 30) ? lb : (1 << 30)), 0x00020000);
#pragma unroll
    for (int st = 0; st < S; ++st) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u4_t r = __builtin_amdgcn_raw_buffer_load_b128(ra, voa[st][h], 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) sa[set][st][h][e] = __uint_as_float(r[e]);
      }
      const u3_t r = __builtin_amdgcn_raw_buffer_load_b96(rb, vob[st], 0, 0);
#pragma unroll
      for (int e = 0; e < 3; ++e) sb[set][st][e] = __uint_as_float(r[e]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto compute = [&](int set) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < S; ++st)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < CT; ++c) acc[h][e][c] = mfma16(sa[set][st][h][e], sb[set][st][c], acc[h][e][c]);
  };
  // this team's whole sets: wid + x W, x = 0 .. nw - 1, in rounds of NBUF through a ring of register sets (NBUF - 1 in flight);
  // a round is straight-line code (see skinny_rows_kernel); a request past the team's last set takes that set again (out of the
  // cache, never consumed).  The matrix's last, partial set (at most one) is taken afterwards: the rows it lacks read as zeros.
  const int64_t nfull = KR / SR;
  const int64_t nw = wid < nfull ? (nfull - wid + W - 1) / W : 0;
  auto set_of = [&](int64_t x) { return wid + (x < nw ? x : nw - 1) * W; };
  if (nw > 0) {
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) fetch(b, set_of(b));
    int64_t x = 0;
    for (; x + NBUF <= nw; x += NBUF) {
#pragma unroll
      for (int b = 0; b < NBUF; ++b) {
        fetch((b + NBUF - 1) % NBUF, set_of(x + b + NBUF - 1));
        compute(b);
      }
    }
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b)
      if (x + b < nw) compute(b);
  }
  if (nfull < nsets && wid == nfull % W) {
    fetch(0, nfull);
    compute(0);
  }
  // D[ii][jj] in lane (jj = l & 15, qq = l >> 4), register r: ii = 4 qq + r -> C[128 half + 64 h + 4 ii + e][3 jj - sh + c];
  // a lane stores the columns that are its own (c >= sh).  The teams of a workgroup add their halves in LDS in team order
  // (fixed order: deterministic).
  constexpr int LDN = 16 * CT;
  for (int w = 0; w < TPW; ++w) {
    if (wv / NH == w) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int m = 128 * half + 64 * h + 4 * (4 * q + r) + e;
              if (c >= sh) {
                float* p = tn_lds + m * LDN + no + c;
                *p = (w == 0) ? acc[h][e][c][r] : *p + acc[h][e][c][r];
              }
            }
    }
    __syncthreads();
  }
  float* P = g.slabs + (int64_t)blockIdx.x * MW * N;
  for (int x = threadIdx.x; x < MW * N; x += 256) {
    const int m = x / N, n = x - m * N;
    P[x] = tn_lds[m * LDN + n];
  }
}

// ---- wide TN:  C[MW x N] (=|+=) A[K x MW]^T . G[K x N],  K (= g.M) in the millions, MW <= 16 MT (7 tiles: the 100 input
// features; 8 compiles too and measures like sgemm.hip's kernel: not dispatched), N = 128 NH -- the weight gradients of the layers
// whose INPUT is 100 wide, plain or with the
// layer's d_relu folded in (MASK: G <- G where mask > 0 else 0 on the way, written back in place; math_functions.cu:258-268).
// The 32 x 32 register-resident kernel of sgemm.hip pays 128 rows for 100 (22 % of its MFMAs) and prefetches ONE set of 8 rows;
// here a wave keeps the whole MW x 128 output of its column half in 16 x 16 accumulators (7 x 8 x 4 = 224 registers), A arrives
// as one 4-byte load per 16-row tile and step (lane (i, q): A[k + q][16 mt + i]: 64 contiguous bytes per row and tile), G and the
// mask as two 16-byte loads (lane (j, q): columns 128 half + 64 hb + 4 j .. + 3: which column a tile holds is free as long as the
// epilogue knows), through a ring of NBUF register sets of S steps.  Loads AND the masked write-back go through buffer
// descriptors whose size is what is left of the matrices: rows past K read as zeros and their stores are dropped by the
// hardware, so the last, partial set needs no code of its own.  Every stream is touched once: non-temporal.  NH = 2: the two
// column halves of a set run in waves w, w + 1 of one workgroup.  Partial outputs go to one slab per team (the halves write
// disjoint columns) and are summed in fixed order by skinny_reduce_kernel.
template <int MT, int NH, bool MASK, int S, int NBUF>
__global__ __launch_bounds__(256, 1) void wide_tn_kernel(SkinnyArgs g, const float* mask, float* gwrite) {
  const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int MW = g.K, N = g.N;  // output rows / columns (N == 128 NH)
  const int64_t KR = g.M;
  constexpr int SR = 4 * S;
  constexpr int TPW = 4 / NH;
  const int half = wv % NH;
  const int64_t W = (int64_t)gridDim.x * TPW;
  const int64_t wid = (int64_t)blockIdx.x * TPW + wv / NH;
  const int64_t nsets = (KR + SR - 1) / SR;
  int voa[S][MT], vog[S][2];
#pragma unroll
  for (int st = 0; st < S; ++st) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) voa[st][mt] = 4 * ((4 * st + q) * MW + (16 * mt + i < MW ? 16 * mt + i : 0));  // past MW: column 0, rows never stored
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) vog[st][hb] = 4 * ((4 * st + q) * N + 128 * half + 64 * hb + 4 * i);
  }
  f4 acc[MT][2][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mt][hb][e] = f4{0.f, 0.f, 0.f, 0.f};
  float sa[NBUF][S][MT];
  f4 sg[NBUF][S][2];
  f4 sm[MASK ? NBUF : 1][MASK ? S : 1][2];
  const int64_t a_bytes = KR * (int64_t)MW * 4, g_bytes = KR * (int64_t)N * 4;
  auto rsrc_at = [&](const float* base, int64_t off, int64_t total) {
    const int64_t left = total - off;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(base) + off), 0,
                                            (int)(left < (1 << 30) ? left : (1 << 30)), 0x00020000);
  };
  auto fetch = [&](int set, int64_t s) {
    const int64_t offa = s * SR * (int64_t)MW * 4, offg = s * SR * (int64_t)N * 4;
    const __amdgpu_buffer_rsrc_t ra = rsrc_at(g.A, offa, a_bytes), rg = rsrc_at(g.B, offg, g_bytes);
#pragma unroll
    for (int st = 0; st < S; ++st) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) sa[set][st][mt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, voa[st][mt], 0, 2));
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const u4_t r = __builtin_amdgcn_raw_buffer_load_b128(rg, vog[st][hb], 0, 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) sg[set][st][hb][e] = __uint_as_float(r[e]);
      }
    }
    if constexpr (MASK) {
      const __amdgpu_buffer_rsrc_t rm = rsrc_at(mask, offg, g_bytes);
#pragma unroll
      for (int st = 0; st < S; ++st)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const u4_t r = __builtin_amdgcn_raw_buffer_load_b128(rm, vog[st][hb], 0, 2);
#pragma unroll
          for (int e = 0; e < 4; ++e) sm[set][st][hb][e] = __uint_as_float(r[e]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto compute = [&](int set, int64_t s) {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MASK) {  // d_relu on the set that is about to be consumed (its loads landed NBUF - 1 sets ago), masked rows written back
      const __amdgpu_buffer_rsrc_t rw = rsrc_at(gwrite, s * SR * (int64_t)N * 4, g_bytes);
#pragma unroll
      for (int st = 0; st < S; ++st)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          u4_t w;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = sm[set][st][hb][e] > 0.f ? sg[set][st][hb][e] : 0.f;
            sg[set][st][hb][e] = v;
            w[e] = __float_as_uint(v);
          }
          __builtin_amdgcn_raw_buffer_store_b128(w, rw, vog[st][hb], 0, 2);  // (rows past K: outside the descriptor, dropped)
        }
    }
#pragma unroll
    for (int st = 0; st < S; ++st)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mt][hb][e] = mfma16(sa[set][st][mt], sg[set][st][hb][e], acc[mt][hb][e]);
  };
  // this team's sets wid + x W, x = 0 .. nw - 1 (the matrix's last set may be partial: zeros), in straight-line rounds of NBUF
  const int64_t nw = wid < nsets ? (nsets - wid + W - 1) / W : 0;
  auto set_of = [&](int64_t x) { return wid + (x < nw ? x : nw - 1) * W; };
  if (nw > 0) {
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) fetch(b, set_of(b));
    int64_t x = 0;
    for (; x + NBUF <= nw; x += NBUF) {
#pragma unroll
      for (int b = 0; b < NBUF; ++b) {
        fetch((b + NBUF - 1) % NBUF, set_of(x + b + NBUF - 1));
        compute(b, wid + (x + b) * W);
      }
    }
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b)
      if (x + b < nw) compute(b, wid + (x + b) * W);
  }
  // D[ii][jj] in lane (jj = l & 15, qq = l >> 4), register r: ii = 4 qq + r -> C[16 mt + ii][128 half + 64 hb + 4 jj + e]: the
  // four e of one (mt, hb, r) are four consecutive columns -> one 16-byte store into the team's slab
  float* P = g.slabs + wid * (int64_t)MW * N;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 16 * mt + 4 * q + r;
      if (m < MW) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          f4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[mt][hb][e][r];
          *reinterpret_cast<f4*>(P + (int64_t)m * N + 128 * half + 64 * hb + 4 * i) = v;
        }
      }
    }
}

// C[i] = (accum ? C[i] : 0) + sum_s slab[s][i], s in order (deterministic)
__global__ void skinny_reduce_kernel(int64_t n, int slabs, const float* partial, int accum, int relu, float* C) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += stride) {
    float s = accum ? C[x] : 0.f;
#pragma unroll 8
    for (int k = 0; k < slabs; ++k) s += partial[(int64_t)k * n + x];
    C[x] = (relu && !(s > 0.f)) ? 0.f : s;
  }
}

struct SkinnyTag {
  char s[28];
  SkinnyTag(int64_t M, int64_t N, int64_t K) { snprintf(s, sizeof(s), "%lldx%lldx%lld", (long long)M, (long long)N, (long long)K); }
};

template <int G, int E, int CT, int RT, bool BT, int WPS, bool LDSOUT, int NBUF>
int launch_rows(gaib_ctx* ctx, const SkinnyArgs& a) {
  const int64_t ntiles = cdiv64(a.M, 16 * RT);
  // column slabs (a.ldn / a.N of them, blockIdx.y): the workgroups of a row share the CUs' wave slots, and -- launched side by
  // side -- walk the same tiles of A at the same time: the second reader of a line finds it in the L2
  const unsigned slabs = (unsigned)(a.ldn / a.N);
  const unsigned blocks = (unsigned)std::min<int64_t>((int64_t)ctx->num_cus * WPS / slabs, cdiv64(ntiles, 4));
  const dim3 grid(blocks, slabs);
  if (a.accum) skinny_rows_kernel<G, E, CT, RT, BT, WPS, LDSOUT, true, NBUF><<<grid, 256, 0, ctx->stream>>>(a);
  else skinny_rows_kernel<G, E, CT, RT, BT, WPS, LDSOUT, false, NBUF><<<grid, 256, 0, ctx->stream>>>(a);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

template <int NH, int S, int WPS, int NBUF>
int launch_tn(gaib_ctx* ctx, SkinnyArgs a) {
  const int64_t nsets = cdiv64(a.M, 4 * S);
  const unsigned blocks = (unsigned)std::min<int64_t>((int64_t)ctx->num_cus * WPS, cdiv64(nsets, 4 / NH));
  const int64_t n = (int64_t)a.K * a.N;
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * (size_t)blocks * (size_t)n));
  a.slabs = (float*)ctx->ws;
  const size_t lds = sizeof(float) * (size_t)128 * NH * 48;
  skinny_tn_kernel<NH, S, WPS, NBUF><<<blocks, 256, lds, ctx->stream>>>(a);
  GAIB_LAUNCH_CHECK();
  const unsigned rg = (unsigned)std::min<int64_t>(cdiv64(n, 256), 1024);
  skinny_reduce_kernel<<<rg, 256, 0, ctx->stream>>>(n, (int)blocks, a.slabs, a.accum, a.relu, a.C);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

template <int MT, int NH, bool MASK, int S, int NBUF>
int launch_wide_tn(gaib_ctx* ctx, SkinnyArgs a, const float* mask, float* gwrite) {
  const int64_t nsets = cdiv64(a.M, 4 * S);
  const unsigned blocks = (unsigned)std::min<int64_t>(ctx->num_cus, cdiv64(nsets, 4 / NH));
  const int64_t n = (int64_t)a.K * a.N;
  const int64_t teams = (int64_t)blocks * (4 / NH);
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * (size_t)teams * (size_t)n));
  a.slabs = (float*)ctx->ws;
  // (teams past the last set store a slab of zeros: every slab the reduce reads is written by this launch)
  wide_tn_kernel<MT, NH, MASK, S, NBUF><<<blocks, 256, 0, ctx->stream>>>(a, mask, gwrite);
  GAIB_LAUNCH_CHECK();
  const unsigned rg = (unsigned)std::min<int64_t>(cdiv64(n, 256), 1024);
  skinny_reduce_kernel<<<rg, 256, 0, ctx->stream>>>(n, (int)teams, a.slabs, a.accum, a.relu, a.C);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

}  // namespace

// Called by gaib_sgemm_ex before its own rules.  *handled = 1: the product was launched here.  Shapes (sgemm_variant 61 turns
// the family off, 62 / 63 the row-stream / TN half only, 64 the column-slab form of rows x 256 x 100, 66 the square products):
//   NN  rows x (33..48) x 64 | 128 | 256      forward of the output layer (64: the GAT models' dense head)
//   NT  rows x 64 | 128 | 256 x (33..48)      input gradient of the output layer
//   NN  rows x 128 x 100                      forward of the first layer (K = 16 G + 4); rows x 256 x 100 as two column slabs
//   NN / NT  rows x 128 x 128, rows x 256 x 256   the hidden layers' products that are not fused into an aggregation
//   TN  (68..128 | 196..256, % 4) x (33..48) x rows   weight gradient of the output layer
int gaib_sgemm_skinny_try(gaib_ctx* ctx, int transA, int transB, int64_t M, int64_t N, int64_t K, const float* d_A,
                          const float* d_B, int flags, float* d_C, int* handled) {
  *handled = 0;
  const int sv = ctx->sgemm_variant;
  if (sv != 0 && sv != 28 && sv != 29 && sv != 39 && sv != 60 && sv != 62 && sv != 63 && sv != 64 && sv != 66 && sv != 67) return GAIB_OK;  // (28 / 29 / 39 / 67: switches of the weight gradients only)
  if ((((uintptr_t)d_A | (uintptr_t)d_B) & 3) != 0 || (((uintptr_t)d_C) & 15) != 0) return GAIB_OK;
  SkinnyArgs a;
  a.A = d_A;
  a.B = d_B;
  a.C = d_C;
  a.accum = (flags & GAIB_ACCUMULATE) ? 1 : 0;
  a.relu = (flags & GAIB_RELU) ? 1 : 0;
  a.slabs = nullptr;
  const double bytes = 4.0 * ((double)M * K + (double)K * N + (double)M * N * (a.accum ? 2.0 : 1.0));
  const double flops = 2.0 * (double)M * (double)N * (double)K;
  if (!transA && M >= 65536 && sv != 63) {
    a.M = M;
    a.N = (int)N;
    a.ldn = (int)N;
    a.K = (int)K;
    const int ct = (int)cdiv64(N, 16);
#define GAIB_ROWS(G_, E_, CT_, RT_, BT_, WPS_, LO_, NB_)                           \
  do {                                                                             \
    GAIB_HIP(hipSetDevice(ctx->device));                                           \
    ProfScope ps(ctx, "sgemm", bytes, flops, 0, SkinnyTag(M, N, K).s);             \
    *handled = 1;                                                                  \
    return launch_rows<G_, E_, CT_, RT_, BT_, WPS_, LO_, NB_>(ctx, a);             \
  } while (0)
    // Schedules as measured at 2.45 M rows (scripts/gemm_narrow.py; ring depth NBUF, waves per SIMD WPS, row tiles per step RT;
    // profiles/r06/gemm_skinny_sweep.jsonl): a deeper ring pays where a wave alone on its SIMD has the registers for it;
    // non-temporal loads / stores changed nothing (0.34-0.36 either way) and are not built.
    if (!transB && ct == 3 && K == 64) GAIB_ROWS(4, 0, 3, 1, false, 2, true, 3);     // 47 x 64 (the GAT models' dense head)
    if (!transB && ct == 3 && K == 128) GAIB_ROWS(8, 0, 3, 1, false, 1, true, 5);    // 0.349 (WPS 2, NBUF 2) -> 0.32 ms (NBUF 4: 0.324, 6: 0.317 with 52-68 registers spilled)
    if (!transB && ct == 3 && K == 256) GAIB_ROWS(16, 0, 3, 1, false, 1, true, 2);   // 0.600 (NBUF 3: 0.605)
    // (the same three products with B stored [K x N] -- no layer of the reference has a 47-wide INPUT, but the form costs nothing)
    if (!transB && N == 128 && K > 32 && K <= 48) GAIB_ROWS(3, 0, 8, 2, false, 1, false, 3);
    if (!transB && N == 256 && K > 32 && K <= 48) GAIB_ROWS(3, 0, 16, 1, false, 1, false, 4);
    if (transB && N == 64 && K > 32 && K <= 48) GAIB_ROWS(3, 0, 4, 2, true, 2, false, 3);
    if (transB && N == 128 && K > 32 && K <= 48) GAIB_ROWS(3, 0, 8, 2, true, 1, false, 3);   // 0.445 (WPS 2, NBUF 2) -> 0.36-0.40
    if (transB && N == 256 && K > 32 && K <= 48) GAIB_ROWS(3, 0, 16, 1, true, 1, false, 4);  // 0.712 (RT 2, NBUF 2) -> 0.69 (RT 2, NBUF 3: its C += form spills 46 registers)
    if (!transB && N == 128 && K == 100) GAIB_ROWS(6, 1, 8, 1, false, 1, false, 2);  // 0.571 (NBUF 3: 0.569)
    // The square hidden-layer products (MFMA-bound: 32 flop / B): the same kernel, no LDS in the loop -- 128 x 128 with the whole
    // matrix in registers, 256 x 256 as four 64-column slabs (each slab's 256 x 64 piece of op(B) = 256 registers; the slabs of a
    // row tile run side by side on the same XCD: traffic stays at the algorithmic bytes).  2.45 M x 256 x 256: 2.75 -> 2.45 ms =
    // 0.83 of the fp32 MFMA peak (the LDS-slab streaming kernel 0.74, rocBLAS 2.51); sgemm_variant 66 keeps the streaming kernel.
    if (sv != 66 && N == 128 && K == 128) {
      if (transB) GAIB_ROWS(8, 0, 8, 1, true, 1, false, 2);
      GAIB_ROWS(8, 0, 8, 1, false, 1, false, 2);
    }
    if (sv != 66 && N == 256 && K == 256) {
      a.N = 64;
      if (transB) GAIB_ROWS(16, 0, 4, 1, true, 1, false, 2);
      GAIB_ROWS(16, 0, 4, 1, false, 1, false, 2);
    }
    // two column slabs of 128 (the matrix in registers is 100 x 128): 1.45 -> 1.11-1.14 ms; four slabs of 64 at two waves per SIMD: 1.32
    if (!transB && N == 256 && K == 100 && sv != 64) {
      a.N = 128;
      GAIB_ROWS(6, 1, 8, 1, false, 1, false, 2);
    }
#undef GAIB_ROWS
    return GAIB_OK;
  }
  if (transA && !transB && K >= 65536 && sv != 62 && M % 4 == 0 && (((uintptr_t)d_A & 15) == 0) && N > 32 && N <= 48) {
    // C[M x N] = A[K x M]^T . B[K x N]: the kernel's names: rows KR = K, output MW = M
    a.M = K;
    a.K = (int)M;
    a.N = (int)N;
    a.ldn = (int)N;
#define GAIB_TNS(NH_, S_, WPS_, NB_)                                               \
  do {                                                                             \
    GAIB_HIP(hipSetDevice(ctx->device));                                           \
    ProfScope ps(ctx, "sgemm", bytes, flops, 0, SkinnyTag(M, N, K).s);             \
    *handled = 1;                                                                  \
    return launch_tn<NH_, S_, WPS_, NB_>(ctx, a);                                  \
  } while (0)
    // (one wave per SIMD with rings of 4-6 sets measured the same 0.34-0.36 / 0.60-0.62 ms: not latency-bound any more)
    if (M <= 128 && M > 64) GAIB_TNS(1, 4, 2, 2);
    if (M <= 256 && M > 192) GAIB_TNS(2, 4, 2, 2);
#undef GAIB_TNS
  }
  return GAIB_OK;
}

// The weight gradients whose input side is 100 or 128 wide and whose output side is 128 or 256 (wide_tn_kernel), plain (mask ==
// NULL) or with the d_relu mask folded in (G is rewritten in place).  Called by gaib_sgemm_ex / gaib_sgemm_drelu before their own
// rules; *handled = 1: launched here.  sgemm_variant 67 turns the form off.
int gaib_sgemm_wide_tn_try(gaib_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* d_A, float* d_G, const float* d_mask,
                           int flags, float* d_C, int* handled) {
  *handled = 0;
  const int sv = ctx->sgemm_variant;
  if (sv != 0 && sv != 28 && sv != 29 && !(sv >= 60 && sv <= 66)) return GAIB_OK;  // (39: the masked form without non-temporal hints = sgemm.hip's kernel)
  // (M <= 112: seven 16-row tiles instead of 128 rows of 32 x 32 tiles.  At 113 .. 128 the two kernels measured the same -- 0.72 / 0.72
  // plain, 0.886-0.893 / 0.889-0.899 masked inside the GCN layer step -- and sgemm.hip's stays)
  if (K < 65536 || (N != 128 && N != 256) || M < 68 || M > 112) return GAIB_OK;
  if ((((uintptr_t)d_A) & 3) != 0 || ((((uintptr_t)d_G | (uintptr_t)d_mask | (uintptr_t)d_C)) & 15) != 0) return GAIB_OK;
  SkinnyArgs a;
  a.A = d_A;
  a.B = d_G;
  a.C = d_C;
  a.M = K;
  a.K = (int)M;
  a.N = (int)N;
  a.ldn = (int)N;
  a.accum = (flags & GAIB_ACCUMULATE) ? 1 : 0;
  a.relu = (flags & GAIB_RELU) ? 1 : 0;
  a.slabs = nullptr;
  const double bytes = 4.0 * ((double)M * K + (double)K * N + (double)M * N * (a.accum ? 2.0 : 1.0) + (d_mask ? 2.0 * (double)K * N : 0.0));
  const double flops = 2.0 * (double)M * (double)N * (double)K;
  GAIB_HIP(hipSetDevice(ctx->device));
  ProfScope ps(ctx, "sgemm", bytes, flops, 0, SkinnyTag(M, N, K).s);
  *handled = 1;
  // (S steps of 4 rows per register set, NBUF sets: the deepest ring without register spills whose loads and stores in
  // flight stay below the 63 that vmcnt counts -- masked, 7 tiles: 4 x 13; masked, 8 tiles: 3 x 14; plain: 3 x 18 / 20)
#define GAIB_WTN(MT_, NH_)                                                                         \
  do {                                                                                             \
    if (d_mask) return launch_wide_tn<MT_, NH_, true, 1, (MT_ == 7 ? 5 : 4)>(ctx, a, d_mask, d_G); \
    return launch_wide_tn<MT_, NH_, false, 2, 4>(ctx, a, nullptr, nullptr);                        \
  } while (0)
  if (N == 128) GAIB_WTN(7, 1);
  GAIB_WTN(7, 2);
#undef GAIB_WTN
}
