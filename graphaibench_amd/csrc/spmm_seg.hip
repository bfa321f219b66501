// spmm_seg.hip -- the fused aggregation + dense product for a numbering WITH LOCALITY, swept in COLUMN SEGMENTS so that what an
// XCD gathers at any moment fits its 4 MB L2.
//
// The case (bench.py's planted-locality leg): vertices come in blocks of consecutive ids (communities; after
// METIS / a community order) and most edges stay inside the block.  The XCD-affine tile supply of spmm_gemm_kernel already sends
// one block at a time to one XCD -- but a block of 16 384 rows x 512 B is 8 MB, twice the L2, and a wave walks a row's edges from
// the block's first column to its last: at any moment the XCD's 512 waves gather from ALL of it (31.9 GB of fabric traffic for a
// 3.0 GB compulsory set).  Running the segments as separate passes over all rows (scripts/locality_colseg.py) loses more to the
// partial sums' round trip through HBM than the hits return.
//
// Here the partial sums never leave the CU.  Every wave OWNS a strip of 8 rows for a whole round and keeps their running sums in
// its LDS strip (the one the dense product later reads its operand from); the block's columns are cut into K segments, and the
// XCD's waves walk them together: phase s of a round = every wave adds its 8 rows' edges whose column lies in segment s.  A row's
// edges are sorted by column and the segments are ascending column ranges, so the edge order of every row -- and every sum, bit
// for bit -- is that of the one-pass kernel.  An XCD's round covers 32 CUs x 16 waves x 8 rows = 4 096 rows; after two rounds a
// wave has the 16 rows of one matrix-core tile and multiplies them with op(W) from LDS.
//   * the edges are stored a second time in (strip, segment, row) order (gaib_seg: column ids + the original edge id for the
//     weights), so a phase is ONE contiguous edge stream per wave -- the software-pipelined stream of the FLAT / RING form
//     (spmm_kernels.h), 8 gathers in flight across row boundaries;
//   * the waves of an XCD are kept in step by a counter per XCD (who has finished phase p): a wave starts phase p + 1 when all
//     but `slack` waves have finished phase p.  The counter is a PACING HINT only -- no result depends on it, a wave that has
//     polled `spin_cap` times goes on alone -- so a launch whose workgroups are not all resident still terminates;
//   * blocks are dealt to the XCDs round robin (workgroup -> XCD = blockIdx & 7, as for the affine tile supply).
// No reference counterpart: update_all (src/gnn/gconv/gcn_aggregator.cpp:48-77) is one OpenMP loop over the rows.
#include <hipcub/hipcub.hpp>
#include "spmm_kernels.h"

struct gaib_seg {
  int64_t* vrowptr;   // [n_strips * K * 8 + 1]: virtual row (strip t, segment s, row r) = edges of row 8 t + r in segment s
  uint32_t* vcol;     // [ne] column ids in virtual-row order
  uint32_t* eperm;    // [ne] the edge of the graph each entry is (per-edge weights are read through it)
  int K, block_rows, heavy_thr;
  int64_t n_strips, nv, ne;
};

void gaib_seg_free(void* p) {
  gaib_seg* s = static_cast<gaib_seg*>(p);
  if (!s) return;
  if (s->vrowptr) (void)hipFree(s->vrowptr);
  if (s->vcol) (void)hipFree(s->vcol);
  if (s->eperm) (void)hipFree(s->eperm);
  delete s;
}

namespace {

struct SegArgs {
  const int64_t* vrowptr;
  const uint32_t* vcol;
  const uint32_t* eperm;
  int* progress;  // [8] per XCD: (wave, phase) pairs finished; zeroed before the launch
  int K;
  int block_rows;
  int sync;      // 0 = waves run free, 1 = in step at the start of every round, 2 = at every phase
  int slack;     // waves of the XCD that may still be in the previous phase
  int spin_cap;  // polls after which a wave stops waiting for good
};

// thread per row: the K - 1 cut positions of the row (first edge whose column is >= the cut) and the K virtual-row lengths.
// Rows above the heavy threshold are aggregated by spmm_heavy_kernel: their virtual rows stay empty.
__global__ void seg_count_kernel(int64_t nv, const int64_t* rowptr, const uint32_t* col, int B, int K, int heavy_thr,
                                 int64_t* vcount, uint32_t* cutpos) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nv) return;
  const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
  const int64_t deg = e1 - e0;
  const int64_t vb = ((i >> 3) * K) * 8 + (i & 7);
  if (deg > heavy_thr) {
    for (int s = 0; s < K; ++s) vcount[vb + (int64_t)s * 8] = 0;
    for (int j = 0; j + 1 < K; ++j) cutpos[i * (K - 1) + j] = 0;
    return;
  }
  const int64_t b0 = (i / B) * B;
  int64_t prev = 0;
  for (int j = 1; j < K; ++j) {
    const int64_t cut = b0 + ((int64_t)j * B) / K;
    int64_t lo = prev, hi = deg;  // cuts ascend: the search continues from the previous one
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)col[e0 + mid] < cut) lo = mid + 1;
      else hi = mid;
    }
    cutpos[i * (K - 1) + j - 1] = (uint32_t)lo;
    vcount[vb + (int64_t)(j - 1) * 8] = lo - prev;
    prev = lo;
  }
  vcount[vb + (int64_t)(K - 1) * 8] = deg - prev;
}

// rows past nv of the last strip: empty virtual rows; entry n_virt closes the scan
__global__ void seg_tail_kernel(int64_t nv, int64_t n_strips, int K, int64_t* vcount) {
  const int64_t t = n_strips - 1;
  const int idx = threadIdx.x;  // (s, r)
  if (idx < K * 8) {
    const int r = idx & 7;
    if (t * 8 + r >= nv) vcount[(t * K) * 8 + idx] = 0;
  }
  if (idx == 0) vcount[n_strips * K * 8] = 0;
}

// one wave per row: every edge to its place in virtual-row order
__global__ __launch_bounds__(256) void seg_fill_kernel(int64_t nv, const int64_t* rowptr, const uint32_t* col, int K, int heavy_thr,
                                                       const int64_t* vrowptr, const uint32_t* cutpos, uint32_t* vcol,
                                                       uint32_t* eperm) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= nv) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[i], e1 = rowptr[i + 1];
  if (e1 - e0 > heavy_thr) return;
  const int64_t vb = ((i >> 3) * K) * 8 + (i & 7);
  for (int64_t k = lane; k < e1 - e0; k += 64) {
    int s = 0;
    uint32_t start = 0;
    for (int j = 0; j + 1 < K; ++j) {
      const uint32_t cp = cutpos[i * (K - 1) + j];
      if ((uint32_t)k >= cp) s = j + 1, start = cp;
    }
    const int64_t dst = vrowptr[vb + (int64_t)s * 8] + (k - start);
    vcol[dst] = col[e0 + k];
    eperm[dst] = (uint32_t)(e0 + k);
  }
}

int seg_build(gaib_ctx* ctx, gaib_graph* g, int B, int K, gaib_seg** out) {
  GAIB_NOT_WHILE_CAPTURING(ctx, "building the graph's column-segment edge order");
  gaib_seg* s = new gaib_seg();
  memset(s, 0, sizeof(*s));
  struct Guard {
    gaib_seg* s;
    void* tmp[2];
    ~Guard() {
      gaib_seg_free(s);
      for (void* p : tmp)
        if (p) (void)hipFree(p);
    }
  } guard{s, {nullptr, nullptr}};
  s->K = K;
  s->block_rows = B;
  s->heavy_thr = g->heavy_thr;
  s->nv = g->nv;
  s->ne = g->ne;
  s->n_strips = cdiv64(g->nv, 8);
  const int64_t n_virt = s->n_strips * K * 8;
  GAIB_HIP(hipMalloc(&s->vrowptr, sizeof(int64_t) * (size_t)(n_virt + 1)));
  GAIB_HIP(hipMalloc(&s->vcol, sizeof(uint32_t) * (size_t)(g->ne > 0 ? g->ne : 1)));
  GAIB_HIP(hipMalloc(&s->eperm, sizeof(uint32_t) * (size_t)(g->ne > 0 ? g->ne : 1)));
  uint32_t* cutpos = nullptr;
  GAIB_HIP(hipMalloc(&cutpos, sizeof(uint32_t) * (size_t)g->nv * (size_t)(K - 1) + 4));
  guard.tmp[0] = cutpos;
  seg_count_kernel<<<(unsigned)cdiv64(g->nv, 256), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->colidx, B, K, g->heavy_thr,
                                                                          s->vrowptr, cutpos);
  GAIB_LAUNCH_CHECK();
  seg_tail_kernel<<<1, 256, 0, ctx->stream>>>(g->nv, s->n_strips, K, s->vrowptr);
  GAIB_LAUNCH_CHECK();
  size_t tmp_bytes = 0;
  GAIB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, s->vrowptr, s->vrowptr, (int)(n_virt + 1), ctx->stream));
  void* tmp = nullptr;
  GAIB_HIP(hipMalloc(&tmp, tmp_bytes + 16));
  guard.tmp[1] = tmp;
  GAIB_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, s->vrowptr, s->vrowptr, (int)(n_virt + 1), ctx->stream));
  seg_fill_kernel<<<(unsigned)cdiv64(g->nv, 4), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->colidx, K, g->heavy_thr, s->vrowptr,
                                                                      cutpos, s->vcol, s->eperm);
  GAIB_LAUNCH_CHECK();
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  guard.s = nullptr;
  *out = s;
  return GAIB_OK;
}

template <int VEC, int WMODE, int GM>
__global__ __launch_bounds__(FUSE_WAVES * 64) void spmm_seg_kernel(SpmmArgs a, FuseArgs f, SegArgs sg) {
  typedef typename VecT<VEC>::type vec_t;
  constexpr int U = 8;
  constexpr int K = 64 * VEC;
  constexpr int KQ = K / 4;
  constexpr int LDT = K + 4;
  constexpr int HALF = 8;
  extern __shared__ __attribute__((aligned(16))) float fuse_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_pad = (f.n_out + 15) & ~15;
  float* wl = fuse_lds;                                 // [n_pad][LDT], zero padded
  float* tile = fuse_lds + n_pad * LDT + wave * (HALF * LDT);  // this wave's strip [HALF][LDT]: the running sums of its 8 rows
  for (int t = threadIdx.x; t < n_pad * K; t += FUSE_WAVES * 64) {
    const int n = t / K, k = t % K;
    wl[n * LDT + k] = (n < f.n_out && k < a.ncols) ? f.wt[(int64_t)n * f.ldw + k] : 0.f;
  }
  __syncthreads();  // the only workgroup barrier
  const int i = lane & 15, kq = lane >> 4;
  const bool colok = lane * VEC < a.ncols;
  const uint32_t voff = colok ? (uint32_t)(lane * VEC * 4) : 0u;
  const RowGather<VEC, GM> gather(a);
  float* trow_w = tile + lane * VEC;  // this lane's columns of strip row 0
  const int x = blockIdx.x & 7;                                // the XCD this workgroup sits on
  const int nwx = (int)(gridDim.x >> 3) * FUSE_WAVES;          // waves of one XCD
  const int wx = (int)(blockIdx.x >> 3) * FUSE_WAVES + wave;   // this wave among them
  const int64_t R = (int64_t)nwx * HALF;                       // rows of one round
  const int64_t B = sg.block_rows;
  const int rounds = 2 * (int)((B + 2 * R - 1) / (2 * R));     // rounds per block (even: two strips make a matrix-core tile)
  const int64_t nblk = (a.n_rows + B - 1) / B;
  const int KS = sg.K;
  int phase = 0;
  bool alone = sg.sync == 0;  // no pacing (any more)
  float af[KQ];
#pragma unroll
  for (int s4 = 0; s4 < KQ; ++s4) af[s4] = 0.f;
  int rid = -1;  // lane q < 16: the row of the caller's matrices that row q of the pending matrix-core tile is (-1: none)
  for (int64_t blk = x; blk < nblk; blk += 8) {
    const int64_t blk_end = (blk + 1) * B < (int64_t)a.n_rows ? (blk + 1) * B : (int64_t)a.n_rows;
    for (int rr = 0; rr < rounds; ++rr) {
      const int64_t row0 = blk * B + (int64_t)rr * R + (int64_t)wx * HALF;
      const bool live = row0 < blk_end;  // wave-uniform
      const int hs = rr & 1;
      unsigned heavy_mask = 0;
      float rwv = 0.f;
      if (live) {
        const int64_t row = row0 + (lane < HALF ? lane : 0);
        int64_t d = 0;
        if (row < a.n_rows) d = a.rowptr[row + 1] - a.rowptr[row];
        heavy_mask = (unsigned)(__ballot(lane < HALF && d > (int64_t)a.heavy_thr) & 0xffull);
        if constexpr (WMODE == 0) rwv = row < a.n_rows ? a.rw[row] : 0.f;
      }
      if (hs == 0) rid = -1;
      if ((lane >> 3) == hs && lane < 16) rid = (live && row0 + (lane & 7) < a.n_rows) ? (int)(row0 + (lane & 7)) : -1;
      for (int s = 0; s < KS; ++s) {
        // the 9 boundaries of this phase's virtual strip (requested before the pacing wait)
        int64_t vp = 0;
        if (live) vp = sg.vrowptr[((row0 >> 3) * KS + s) * 8 + (lane < HALF ? lane : HALF)];
        if (!alone && (sg.sync == 2 || s == 0)) {
          const int need = phase * nwx - sg.slack;
          int spins = 0;
          while (need > 0) {
            int v = 0;
            if (lane == 0) v = __hip_atomic_fetch_add(sg.progress + x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = __builtin_amdgcn_readfirstlane(v);
            if (v >= need) break;
            if (++spins > sg.spin_cap) {
              alone = true;
              break;
            }
            __builtin_amdgcn_s_sleep(4);
          }
        }
        if (live) {
          const int vp_lo = (int)(uint32_t)(vp & 0xffffffffll), vp_hi = (int)(vp >> 32);
          auto rp_at = [&](int q) -> int64_t {
            return ((int64_t)__builtin_amdgcn_readlane(vp_hi, q) << 32) | (uint32_t)__builtin_amdgcn_readlane(vp_lo, q);
          };
          const int64_t e_lo = rp_at(0), e_hi = rp_at(HALF);
          const int total = (int)(e_hi - e_lo);
          const bool last = s == KS - 1;
          if (total > 0 || s == 0 || last) {  // (a middle phase without edges leaves the strip as it is)
            uint32_t c_cur = 0, c_nxt = 0;
            float w_cur = 0.f, w_nxt = 0.f;
            auto load_ids = [&](int q, uint32_t& c, float& w) {
              const int64_t e = e_lo + 64 * (int64_t)q + lane;
              c = 0;
              w = 0.f;
              if (e < e_hi) {
                c = sg.vcol[e];
                if constexpr (WMODE == 1) w = a.ew[sg.eperm[e]];
              }
            };
            load_ids(0, c_cur, w_cur);
            if (total > 64) load_ids(1, c_nxt, w_nxt);
            int r = 0;
            int64_t row_end = rp_at(1);
            vec_t acc = s == 0 ? vzero<VEC>() : *reinterpret_cast<const vec_t*>(trow_w);
            float roww = (WMODE == 0) ? readlane_f(rwv, 0) : 0.f;
            auto flush = [&]() {  // the part of row r in this segment is summed: park it (last segment: the row is complete)
              if (last) {
                const int64_t row = row0 + r;
                if ((heavy_mask >> r) & 1u) {  // a heavy row: its aggregate comes from spmm_heavy_kernel's scratch
                  int lo = 0, hi = f.n_heavy - 1;
                  while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (f.heavy_rows[mid] < (uint32_t)row) lo = mid + 1;
                    else hi = mid;
                  }
                  acc = colok ? *reinterpret_cast<const vec_t*>(f.heavy_agg + (int64_t)lo * a.ldo + lane * VEC) : vzero<VEC>();
                }
                if (row < a.n_rows && a.out && colok)
                  __builtin_nontemporal_store(acc, reinterpret_cast<vec_t*>(a.out + row * a.ldo + lane * VEC));
              }
              *reinterpret_cast<vec_t*>(trow_w + r * LDT) = colok ? acc : vzero<VEC>();
              ++r;
              if (r < HALF) {
                row_end = rp_at(r + 1);
                acc = s == 0 ? vzero<VEC>() : *reinterpret_cast<const vec_t*>(trow_w + r * LDT);
                if constexpr (WMODE == 0) roww = readlane_f(rwv, r);
              }
            };
            vec_t xg[U];
            if (total > 0) {
#pragma unroll
              for (int u = 0; u < U; ++u)  // (entries past the phase's end read row 0 of the table: never consumed)
                xg[u] = gather.load((uint32_t)__builtin_amdgcn_readlane((int)c_cur, u), voff);
            }
            for (int k = 0; k < total; k += U) {
              const int kc = k & 63;
              if (kc == 0 && k > 0) {  // entering the next 64 edges: their ids are here, request the ones after them
                c_cur = c_nxt;
                w_cur = w_nxt;
                if (k + 64 < total) load_ids((k >> 6) + 1, c_nxt, w_nxt);
              }
              const bool wrap = kc == 64 - U;  // the refills of this batch belong to the next 64 edges
              const uint32_t c_src = wrap ? c_nxt : c_cur;
              if (k + 2 * U <= total) {  // a full batch with a full batch behind it: straight-line
#pragma unroll
                for (int u = 0; u < U; ++u) {
                  while (e_lo + k + u == row_end) flush();
                  vacc<VEC>(acc, (WMODE == 0) ? roww : readlane_f(w_cur, kc + u), xg[u]);
                  xg[u] = gather.load((uint32_t)__builtin_amdgcn_readlane((int)c_src, (kc + u + U) & 63), voff);
                }
              } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                  if (k + u < total) {
                    while (e_lo + k + u == row_end) flush();
                    vacc<VEC>(acc, (WMODE == 0) ? roww : readlane_f(w_cur, kc + u), xg[u]);
                  }
                  if (k + u + U < total)
                    xg[u] = gather.load((uint32_t)__builtin_amdgcn_readlane((int)c_src, (kc + u + U) & 63), voff);
                }
              }
            }
            while (r < HALF) flush();  // the row in progress and the rows without edges in this segment behind it
          }
        }
        // (a wave that stopped waiting still reports: the others are not held up by it)
        if (sg.sync != 0 && lane == 0) __hip_atomic_fetch_add(sg.progress + x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++phase;
      }
      // the strip's 8 complete rows into matrix-core operand order (lane l: A[i = l & 15][k = (l >> 4) * K / 4 + s4])
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      {
        const bool mine = (i / HALF) == hs;
        const float* trow = tile + (i % HALF) * LDT + kq * KQ;
#pragma unroll
        for (int s4 = 0; s4 < KQ / 4; ++s4) {
          f32x4_t tv = *reinterpret_cast<const f32x4_t*>(trow + 4 * s4);
          if (!live) tv = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) af[4 * s4 + e] = mine ? tv[e] : af[4 * s4 + e];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (hs == 1 && __ballot(rid >= 0) != 0) {
        const float* wbase = wl + i * LDT + kq * KQ;
        int yrow[4];  // C/D layout: row = 4 * (lane >> 4) + reg, col = lane & 15
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) yrow[reg] = __shfl(rid, 4 * kq + reg, 64);
        for (int n0 = 0; n0 < n_pad; n0 += 16) {
          f32x4_t c = {0.f, 0.f, 0.f, 0.f};
          const float* wr = wbase + n0 * LDT;
#pragma unroll
          for (int s4 = 0; s4 < KQ / 4; ++s4) {
            const f32x4_t b = *reinterpret_cast<const f32x4_t*>(wr + 4 * s4);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * s4 + 0], b[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * s4 + 1], b[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * s4 + 2], b[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * s4 + 3], b[3], c, 0, 0, 0);
          }
          if (n0 + i < f.n_out) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              if (yrow[reg] >= 0) {
                float v = c[reg];
                if (f.relu) v = v > 0.f ? v : 0.f;
                f.y[(int64_t)yrow[reg] * f.ldy + n0 + i] = v;
              }
            }
          }
        }
      }
    }
  }
}

template <int VEC, int WMODE>
int launch_seg(gaib_ctx* ctx, gaib_graph* g, const SpmmArgs& a, const FuseArgs& f, const SegArgs& sg, float* heavy_scratch) {
  constexpr int U = 16;
  constexpr int K = 64 * VEC;
  const bool buf = a.in_bytes != 0 && ctx->spmm_addr_mode != 2;
  if (g->n_heavy > 0) {
    SpmmArgs h = a;
    h.row_list = g->heavy_rows;
    h.row_order = g->heavy_rows + g->n_heavy;
    h.out = heavy_scratch;
    h.compact = 1;
    h.relu = 0;
    h.accumulate = 0;
    const size_t lds = sizeof(float) * HEAVY_WAVES * 64 * VEC;
    ProfScope ps(ctx, "spmm_heavy");
    if (buf) spmm_heavy_kernel<VEC, 1, WMODE, U, 1><<<dim3((unsigned)g->n_heavy), HEAVY_WAVES * 64, lds, ctx->stream>>>(h);
    else spmm_heavy_kernel<VEC, 1, WMODE, U, 0><<<dim3((unsigned)g->n_heavy), HEAVY_WAVES * 64, lds, ctx->stream>>>(h);
    GAIB_LAUNCH_CHECK();
  }
  const size_t lds = fuse_lds_bytes(K, f.n_out, false, 8);
  int cus = ctx->spmm_fuse_cus > 0 ? ctx->spmm_fuse_cus : ctx->num_cus;
  if (f.overlaps_transfer && ctx->comm_reserve_cus > 0) cus = std::max(cus - ctx->comm_reserve_cus, std::min(cus, 64));
  const unsigned grid = (unsigned)std::max(8, cus & ~7);  // the same number of workgroups on every XCD
  GAIB_HIP(hipMemsetAsync(f.tile_counter, 0, 8 * sizeof(int), ctx->stream));
  ProfScope ps(ctx, "spmm_gemm_seg");
  if (buf) {
    GAIB_HIP(hipFuncSetAttribute((const void*)spmm_seg_kernel<VEC, WMODE, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    spmm_seg_kernel<VEC, WMODE, 1><<<dim3(grid), FUSE_WAVES * 64, lds, ctx->stream>>>(a, f, sg);
  } else {
    GAIB_HIP(hipFuncSetAttribute((const void*)spmm_seg_kernel<VEC, WMODE, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    spmm_seg_kernel<VEC, WMODE, 0><<<dim3(grid), FUSE_WAVES * 64, lds, ctx->stream>>>(a, f, sg);
  }
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

}  // namespace

// GAIB_ERR_UNSUPPORTED: not a shape for this form -- the caller goes on with the one-pass kernel
int gaib_spmm_seg_fused(gaib_ctx* ctx, gaib_graph* g, const void* spmm_args, const void* fuse_args, float* heavy_scratch, int vec,
                        int wmode, int segments) {
  const SpmmArgs& a = *static_cast<const SpmmArgs*>(spmm_args);
  const FuseArgs& f = *static_cast<const FuseArgs*>(fuse_args);
  const int B = ctx->spmm_seg_block > 0 ? ctx->spmm_seg_block & ~7 : 16384;
  if (segments < 2 || segments > 32 || B < 8 || g->nc != g->nv || g->rows_unsorted || g->row_map || a.in2 || f.agg_in || f.wt2 ||
      f.y_accum || wmode > 1 || g->ne >= ((int64_t)1 << 32) || cdiv64(g->nv, 8) * segments * 8 >= ((int64_t)1 << 31) - 1 || fuse_strip_rows(64 * vec, f.n_out, false) != 8 || ctx->capturing)
    return GAIB_ERR_UNSUPPORTED;
  gaib_seg* s = static_cast<gaib_seg*>(g->seg);
  if (!s || s->K != segments || s->block_rows != B || s->heavy_thr != g->heavy_thr) {
    gaib_seg_free(g->seg);
    g->seg = nullptr;
    GAIB_TRY(seg_build(ctx, g, B, segments, &s));
    g->seg = s;
  }
  SegArgs sg;
  sg.vrowptr = s->vrowptr;
  sg.vcol = s->vcol;
  sg.eperm = s->eperm;
  sg.progress = f.tile_counter;
  sg.K = segments;
  sg.block_rows = B;
  sg.sync = ctx->spmm_seg_sync;
  sg.slack = ctx->spmm_seg_slack;
  sg.spin_cap = 256;
  if (vec == 1) return wmode == 0 ? launch_seg<1, 0>(ctx, g, a, f, sg, heavy_scratch) : launch_seg<1, 1>(ctx, g, a, f, sg, heavy_scratch);
  return wmode == 0 ? launch_seg<2, 0>(ctx, g, a, f, sg, heavy_scratch) : launch_seg<2, 1>(ctx, g, a, f, sg, heavy_scratch);
}
