// libpylattice_hip.so — host side of the C ABI declared in include/pylattice_hip.h (gfx950 / MI355X).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <functional>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/pylattice_hip.h"
#include "pl_kernels.h"
#include "pl_parallel.h"
#include "pl_tile.h"
#include "pl_dist.h"
#include "pl_coarse.h"
#include "pl_cg1.h"
#include "pl_palette.h"
#include "pl_ddm.h"
#include "pl_lzone.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define PL_HIP(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess)                                                                              \
      return fail(PL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                     \
  } while (0)

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  hipError_t alloc(size_t count) {
    release();
    n = count;
    if (count == 0) return hipSuccess;
    return hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
  }
};

inline unsigned grid_for(int64_t n, int block = pl::kBlock) { return (unsigned)((n + block - 1) / block); }
inline unsigned grid_stream(int64_t n) {
  // memory-bound grid-stride kernels: cap at 256 CUs x 8 blocks
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + pl::kBlock - 1) / pl::kBlock, 2048));
}

}  // namespace

struct pl_context {
  pl_opts_t opt{};
  pl::Material mat{};
  int64_t N = 0, B = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // pl_assemble overlaps the latency-bound dense factorisation chain (main stream) with the bandwidth-bound fills
  // (palette, Jacobi diagonal, tile blocks, explicit BSR) on a second stream
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_chol = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  hipStream_t side2 = nullptr;   // tile blocks of the 12-mode dense level beside its strain rows
  bool assembled = false, have_bc = false, have_bsr = false;
  int pal_fallback_flags[2] = {1, 0};
  int *pal_host_flags = pal_fallback_flags;   // pinned once the palette is in use: a D2H copy into pageable memory blocks the host
  bool want_bsr = false;   // pl_assemble_bsr was called once: pl_assemble keeps the explicit matrix current
  int bsr_with_bc = 0;

  // caller numbering <-> device numbering (perm[dev] = caller node)
  std::vector<int32_t> perm, iperm;
  bool reordered = false;
  // caller strut order <-> device strut order (bperm[dev] = caller strut); struts are numbered by home tile
  std::vector<int32_t> bperm;

  // geometry / topology (device numbering)
  DevBuf<double> xyz, radius, seg_len;
  DevBuf<int32_t> conn, seg_nsub;
  DevBuf<pl::Record> rec;
  DevBuf<double> rec5;   // compact 5-scalar copy of the records for the streaming K*p (tile kernel, no palette)
  // node -> strut incidence, sliced ELL
  DevBuf<int64_t> slice_ptr;
  DevBuf<int2> ent;
  int64_t n_slices = 0, n_ent = 0;
  int lpn = pl::kDefaultLPN;   // lanes per node of the gather kernels (1, 2, 4, 8 or 16)
  // BSR
  DevBuf<int64_t> bsr_rowptr;
  DevBuf<int32_t> bsr_col, ent_slot, diag_slot;
  DevBuf<double> bsr_vals;
  int64_t nblk = 0;
  std::vector<int64_t> h_rowptr;
  std::vector<int32_t> h_col;
  // boundary data
  DevBuf<uint8_t> fixed;       // [6N] 0/1
  DevBuf<uint8_t> fixedbits;   // [N] 6 bits
  DevBuf<double> ubar, f;
  // solver state
  DevBuf<double> diag, dinv, x, r, z, p, Ap, tmp, tmp2, scal, hist;
  DevBuf<double> cg1;   // single-reduction PCG: two reduction blocks, r_c.y_c slots, (gamma, alpha) pairs, Z^T s
  int hist_cap = 0;
  // LDS-tile operator
  pl::TilePlan tile;
  // DDM operator (pl_ddm.h): opkind = 1 replaces the strut operator by sum_c B^T S B
  int opkind = 0;
  int64_t ddm_cells = 0;
  int ddm_nb = 0;
  DevBuf<int32_t> ddm_cell_nodes, ddm_cell_S;
  DevBuf<double> ddm_St;
  DevBuf<int64_t> ddm_node_ptr;       // node -> (cell * nb + slot) entries: the atomic-free scatter of k_ddm_node_gather
  DevBuf<int32_t> ddm_node_ent;
  DevBuf<double> ddm_stage;           // [cells][6 nb] local products
  DevBuf<int32_t> ddm_order;          // cells sorted by matrix id (k_ddm_cell_product_lds)
  // assembled-Schur preconditioner of the DDM operator (opt.precond = 2): optional palette of its own + dense factor
  DevBuf<int32_t> ddm_cell_P;
  DevBuf<double> ddm_Pt;
  bool ddm_have_P = false;
  DevBuf<double> dd_A, dd_Lf, dd_W, dd_Wt, dd_Dinv, dd_tv;
  DevBuf<int> dd_info;
  int dd_n = 0;          // padded order of the dense matrix (0: not allocated)
  int dd_bw = 0;         // its block bandwidth in the caller's node numbering (from the cells' node spans)
  bool dd_ready = false;
  // record palette (pl_palette.h)
  DevBuf<unsigned long long> pal_keys;
  DevBuf<int> pal_owner, pal_flags;
  DevBuf<uint16_t> pal_id;
  DevBuf<pl::Record> palette;
  bool pal_ready = false;
  int pal_entries = 0;
  // two-level preconditioner (rigid-body coarse space)
  pl::Coarse coarse;
  int coarse_info = 0;
  // precond = 4: a second, rank-LOCAL dense level (aggregates of this handle only, nodes shared with other ranks left
  // out, never communicated) under the global one, so that the aggregate size can stay fixed under weak scaling
  // while the all-reduced global level coarsens
  pl::Coarse coarseL;
  DevBuf<uint8_t> sharedbits, maskL;
  // exact elimination of an independent node set inside the PCG (opts.condense, pl_coarse.h)
  std::vector<uint8_t> h_cand;        // candidates (an independent set of the node graph, chosen at pl_create; device numbering)
  std::vector<uint8_t> h_shared;      // multi-GPU: nodes that also live on another rank (never condensed)
  DevBuf<uint8_t> cend;           // strut -> bits: end A / B is a condensed node (pl_tile.h CondSolve)
  DevBuf<int32_t> cnodes, cbase;  // condensed nodes; node -> offset of its K_cc^-1 block (class table or per node), -1
  int cbase_state = -1;           // what cbase was built for: -1 stale, 0 per-node blocks, 1 class table
  DevBuf<double> kcc_inv;
  DevBuf<uint8_t> maskC, cflag;       // Dirichlet bits | 0x3f on condensed nodes; 1 on condensed nodes
  // classes of eliminated nodes with the same K_cc^-1 (pl_coarse.h k_cls_*): only with a record palette
  DevBuf<unsigned long long> cls_key, cls_keys;
  DevBuf<int> cls_owner, cls_flags;
  DevBuf<uint16_t> cls_id;
  DevBuf<double> cls_table;
  int *cls_host_flag = nullptr;       // pinned
  bool cls_ready = false;
  int last_iterations = 0;   // of the previous converged pcg_solve on this handle (hint for the first convergence check)
  int64_t n_cond = 0;
  bool cond_ready = false;   // K_cc^-1 valid for the current records and mask
  bool cond_use = false;     // the running solve eliminates them (fp64 PCG and precision = 1)
  // multi-GPU
  pl::Dist dist;

  pl_stats_t last{};
  double ms_assembly = 0.0;

  ~pl_context() {
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (ev_chol) (void)hipEventDestroy(ev_chol);
    if (ev_t0) (void)hipEventDestroy(ev_t0);
    if (ev_t1) (void)hipEventDestroy(ev_t1);
    if (side2) (void)hipStreamDestroy(side2);
    if (side) (void)hipStreamDestroy(side);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

// ----------------------------------------------------------------------------------------------------------
// host <-> device vector transfer in caller numbering
// ----------------------------------------------------------------------------------------------------------
int upload6(pl_context *c, const double *host, double *dev, std::vector<double> &stage) {
  const size_t n6 = (size_t)c->N * 6;
  if (!c->reordered) {
    PL_HIP(hipMemcpyAsync(dev, host, n6 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    PL_HIP(hipStreamSynchronize(c->stream));
    return PL_OK;
  }
  stage.resize(n6);
  for (int64_t i = 0; i < c->N; ++i) std::memcpy(&stage[6 * i], host + 6 * (size_t)c->perm[i], 6 * sizeof(double));
  PL_HIP(hipMemcpyAsync(dev, stage.data(), n6 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  PL_HIP(hipStreamSynchronize(c->stream));
  return PL_OK;
}

int download6(pl_context *c, const double *dev, double *host) {
  const size_t n6 = (size_t)c->N * 6;
  if (!c->reordered) {
    PL_HIP(hipMemcpyAsync(host, dev, n6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PL_HIP(hipStreamSynchronize(c->stream));
    return PL_OK;
  }
  std::vector<double> stage(n6);
  PL_HIP(hipMemcpyAsync(stage.data(), dev, n6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  PL_HIP(hipStreamSynchronize(c->stream));
  for (int64_t i = 0; i < c->N; ++i) std::memcpy(host + 6 * (size_t)c->perm[i], &stage[6 * i], 6 * sizeof(double));
  return PL_OK;
}

// ----------------------------------------------------------------------------------------------------------
// operator launches (device vectors, device numbering)
// ----------------------------------------------------------------------------------------------------------
// auto: the LDS-tile kernel when the nodes are brick-ordered (its tiles are then compact), else the per-node gather
int choose_kernel(const pl_context *c) {
  if (c->opt.spmv_kernel != 0) return c->opt.spmv_kernel;
  return (c->reordered && c->tile.ready) ? 3 : 2;
}

template <int LPN>
void launch_gather_lpn(pl_context *c, const double *x, double *y, bool masked, double *dot_dev) {
  const unsigned g = grid_for(c->n_slices, pl::kBlock / pl::kWave);   // one wave per ELL slice
#define PL_G(M, D)                                                                                               \
  hipLaunchKernelGGL((pl::k_spmv_gather<LPN, M, D>), dim3(g), dim3(pl::kBlock), 0, c->stream, c->N, c->slice_ptr.p, \
                     c->ent.p, c->rec.p, c->fixedbits.p, x, y, dot_dev)
  if (masked && dot_dev) PL_G(true, true);
  else if (masked) PL_G(true, false);
  else if (dot_dev) PL_G(false, true);
  else PL_G(false, false);
#undef PL_G
}

int dispatch_gather(pl_context *c, const double *x, double *y, bool masked, double *dot_dev) {
  switch (c->lpn) {
    case 1: launch_gather_lpn<1>(c, x, y, masked, dot_dev); break;
    case 2: launch_gather_lpn<2>(c, x, y, masked, dot_dev); break;
    case 4: launch_gather_lpn<4>(c, x, y, masked, dot_dev); break;
    case 8: launch_gather_lpn<8>(c, x, y, masked, dot_dev); break;
    case 16: launch_gather_lpn<16>(c, x, y, masked, dot_dev); break;
    default: return fail(PL_ERR_ARG, "lanes per node must be 1, 2, 4, 8 or 16");
  }
  return PL_OK;
}

// y = K x (masked -> y = P K x, x assumed zero on fixed dofs); optional dot(x, y) accumulated into *dot_dev.
// K_cc^-1 for the fused first pass of the condensed operator (kEndsCondensedSolve)
__global__ __launch_bounds__(pl::kBlock) void k_cond_ends(int64_t B, const int2 *__restrict__ conn2,
                                                         const uint8_t *__restrict__ cflag, uint8_t *__restrict__ cend) {
  const int64_t b = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x;
  if (b >= B) return;
  const int2 c = conn2[b];
  cend[b] = (uint8_t)((cflag[c.x] ? 1 : 0) | (cflag[c.y] ? 2 : 0));
}
__global__ __launch_bounds__(pl::kBlock) void k_cond_base(int64_t nc, const int32_t *__restrict__ cnodes,
                                                         const uint16_t *__restrict__ cls /* may be null */,
                                                         int32_t *__restrict__ base) {
  const int64_t q = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x;
  if (q < nc) base[cnodes[q]] = 36 * (cls ? (int32_t)cls[q] : (int32_t)q);
}
inline pl::CondSolve cond_solve(pl_context *c, int ends) {
  pl::CondSolve cs;
  if (ends != pl::kEndsAll) cs.cend = c->cend.p;
  if (ends == pl::kEndsCondensedSolve) {
    const int want = c->cls_ready ? 1 : 0;
    if (c->cbase_state != want) {   // (after pl_set_bc / when the class table comes or goes: once per assembly at most)
      (void)hipMemsetAsync(c->cbase.p, 0xFF, (size_t)c->N * sizeof(int32_t), c->stream);
      hipLaunchKernelGGL(k_cond_base, dim3(grid_for(c->n_cond)), dim3(pl::kBlock), 0, c->stream, c->n_cond, c->cnodes.p,
                         c->cls_ready ? (const uint16_t *)c->cls_id.p : (const uint16_t *)nullptr, c->cbase.p);
      c->cbase_state = want;
    }
    cs.inv = c->cls_ready ? (const double *)c->cls_table.p : (const double *)c->kcc_inv.p;
    cs.base = c->cbase.p;
  }
  return cs;
}

int launch_spmv(pl_context *c, const double *x, double *y, bool masked, double *dot_dev,
                const uint8_t *maskbits = nullptr, int ends = pl::kEndsAll, bool reduce_dot = true) {
  const int kind = choose_kernel(c);
  if ((maskbits || ends != pl::kEndsAll) && kind == 3 && c->tile.ready && c->opkind == 0) {
    // tile kernel with a caller-chosen row mask and / or only one kind of strut ends (node elimination)
    const uint8_t *cf = ends != pl::kEndsAll ? (const uint8_t *)c->cflag.p : (const uint8_t *)nullptr;
    const pl::CondSolve cs = cond_solve(c, ends);
    if (c->pal_ready)
      pl::launch_tile_spmv(c->tile, c->conn.p, c->palette.p, c->pal_id.p, maskbits, x, y, dot_dev, c->stream,
                           (const double *)nullptr, ends, cf, cs);
    else if (c->rec5.p)
      pl::launch_tile_spmv(c->tile, c->conn.p, reinterpret_cast<const pl::Record *>(c->rec5.p), nullptr, maskbits, x, y,
                           dot_dev, c->stream, c->xyz.p, ends, cf, cs);
    else
      pl::launch_tile_spmv(c->tile, c->conn.p, c->rec.p, nullptr, maskbits, x, y, dot_dev, c->stream,
                           (const double *)nullptr, ends, cf, cs);
    PL_HIP(hipGetLastError());
    return PL_OK;
  }
  const int64_t n6 = c->N * 6;
  if (c->opkind == 1) {
    const int m = 6 * c->ddm_nb;
    const size_t lds = ((size_t)m * m + (size_t)(pl::kBlock / pl::kWave) * m) * sizeof(double);
    const unsigned gw = grid_for((c->ddm_cells + pl::kDdmWaveChunk - 1) / pl::kDdmWaveChunk, pl::kBlock / pl::kWave);
    if (m <= 48)
      hipLaunchKernelGGL(pl::k_ddm_cell_product_reg<48>, dim3(gw), dim3(pl::kBlock), 0, c->stream, c->ddm_cells,
                         c->ddm_nb, c->ddm_order.p, c->ddm_cell_nodes.p, c->ddm_cell_S.p, c->ddm_St.p, x, c->ddm_stage.p);
    else if (m <= 64)
      hipLaunchKernelGGL(pl::k_ddm_cell_product_reg<64>, dim3(gw), dim3(pl::kBlock), 0, c->stream, c->ddm_cells,
                         c->ddm_nb, c->ddm_order.p, c->ddm_cell_nodes.p, c->ddm_cell_S.p, c->ddm_St.p, x, c->ddm_stage.p);
    else if (lds <= 64 * 1024)
      hipLaunchKernelGGL(pl::k_ddm_cell_product_lds, dim3(grid_for(c->ddm_cells, pl::kDdmChunk)), dim3(pl::kBlock), lds,
                         c->stream, c->ddm_cells, c->ddm_nb, c->ddm_order.p, c->ddm_cell_nodes.p, c->ddm_cell_S.p,
                         c->ddm_St.p, x, c->ddm_stage.p);
    else
      hipLaunchKernelGGL(pl::k_ddm_cell_product, dim3(grid_for(c->ddm_cells, pl::kBlock / pl::kWave)), dim3(pl::kBlock),
                         0, c->stream, c->ddm_cells, c->ddm_nb, c->ddm_cell_nodes.p, c->ddm_cell_S.p, c->ddm_St.p, x,
                         c->ddm_stage.p);
    hipLaunchKernelGGL(pl::k_ddm_node_gather, dim3(grid_for(n6)), dim3(pl::kBlock), 0, c->stream, c->N,
                       c->ddm_node_ptr.p, c->ddm_node_ent.p, (const double *)c->ddm_stage.p, y);
    if (masked || dot_dev)
      hipLaunchKernelGGL(pl::k_mask_dot, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6,
                         masked ? c->fixed.p : (const uint8_t *)nullptr, x, y, dot_dev);
    PL_HIP(hipGetLastError());
    return PL_OK;
  }
  if (kind == 1) {
    PL_HIP(hipMemsetAsync(y, 0, n6 * sizeof(double), c->stream));
    hipLaunchKernelGGL(pl::k_spmv_atomic, dim3(grid_for(c->B)), dim3(pl::kBlock), 0, c->stream, c->B, c->conn.p,
                       c->rec.p, x, y);
    if (masked || dot_dev)
      hipLaunchKernelGGL(pl::k_mask_dot, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6,
                         masked ? c->fixed.p : (const uint8_t *)nullptr, x, y, dot_dev);
  } else if (kind == 3 && c->tile.ready) {
    if (c->pal_ready)
      pl::launch_tile_spmv(c->tile, c->conn.p, c->palette.p, c->pal_id.p, masked ? c->fixedbits.p : nullptr, x, y,
                           dot_dev, c->stream);
    else if (c->rec5.p)
      pl::launch_tile_spmv(c->tile, c->conn.p, reinterpret_cast<const pl::Record *>(c->rec5.p), nullptr,
                           masked ? c->fixedbits.p : nullptr, x, y, dot_dev, c->stream, c->xyz.p);
    else
      pl::launch_tile_spmv(c->tile, c->conn.p, c->rec.p, nullptr, masked ? c->fixedbits.p : nullptr, x, y, dot_dev,
                           c->stream);
  } else {
    int rc = dispatch_gather(c, x, y, masked, dot_dev);
    if (rc) return rc;
  }
  if (c->dist.active) {
    // Every rank now holds the product of ITS struts.  The Dirichlet mask commutes with the sum over ranks, and
    // x.(K x) = sum_r x_r.(K_r x_r) with the LOCAL partial products and NO multiplicity weights, so the kernels above
    // ran exactly as on one GPU; the interface rows and the 32 slots of the partial dot travel in one all-reduce.
    // (reduce_dot = false: the caller sums the dot slots in a collective of its own - single-reduction PCG)
    const bool with_dot = dot_dev && reduce_dot;
    int rc = pl::dist_sum_shared(c->dist, y, c->stream, with_dot ? dot_dev : nullptr, with_dot ? pl::kSlots : 0);
    if (rc) return fail(PL_ERR_HIP, "RCCL all-reduce of interface forces failed");
  }
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// K*p of the fp32 solver modes: tile kernel only, fp32-stored x / y, fp64 arithmetic (pl_tile.h)
int launch_spmv_f32(pl_context *c, const float *x, float *y, bool masked, double *dot_dev,
                    const uint8_t *maskbits = nullptr, int ends = pl::kEndsAll) {
  const uint8_t *mk = maskbits ? maskbits : (masked ? (const uint8_t *)c->fixedbits.p : (const uint8_t *)nullptr);
  const uint8_t *cf = ends != pl::kEndsAll ? (const uint8_t *)c->cflag.p : (const uint8_t *)nullptr;
  const pl::CondSolve cs = cond_solve(c, ends);
  if (c->pal_ready)
    pl::launch_tile_spmv<float>(c->tile, c->conn.p, c->palette.p, c->pal_id.p, mk, x, y, dot_dev, c->stream,
                                (const double *)nullptr, ends, cf, cs);
  else if (c->rec5.p)
    pl::launch_tile_spmv<float>(c->tile, c->conn.p, reinterpret_cast<const pl::Record *>(c->rec5.p), nullptr, mk, x, y,
                                dot_dev, c->stream, c->xyz.p, ends, cf, cs);
  else
    pl::launch_tile_spmv<float>(c->tile, c->conn.p, c->rec.p, nullptr, mk, x, y, dot_dev, c->stream,
                                (const double *)nullptr, ends, cf, cs);
  if (c->dist.active) {
    int rc = pl::dist_sum_shared<float>(c->dist, y, c->stream, dot_dev, dot_dev ? pl::kSlots : 0);
    if (rc) return fail(PL_ERR_HIP, "RCCL all-reduce of interface forces failed");
  }
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// residual history + (reference-CG mode) direction norm, solution norm and step length of every iteration
int ensure_hist(pl_context *c, int cap) {
  if (cap <= c->hist_cap) return PL_OK;
  PL_HIP(c->hist.alloc((size_t)cap * 4));
  c->hist_cap = cap;
  return PL_OK;
}
// the reference's CG extras are on when the caller asked for any of them (pl_opts_t.mintol / restart_every)
inline bool ref_cg(const pl_context *c) { return c->opt.mintol > 0.0 || c->opt.restart_every > 0; }

int launch_records(pl_context *c) {
  hipLaunchKernelGGL(pl::k_build_records, dim3(grid_for(c->B)), dim3(pl::kBlock), 0, c->stream, c->B, c->xyz.p,
                     c->conn.p, c->radius.p, c->seg_len.p, c->seg_nsub.p, c->mat, c->rec.p, c->rec5.p);
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// Try to replace the per-strut records by palette ids (periodic lattices); leaves pal_ready = false otherwise.
// launch_palette queues the kernels and the flag read-back on `st`; finish_palette (after a sync) reads the verdict.
int launch_palette(pl_context *c, hipStream_t st) {
  c->pal_ready = false;
  c->pal_host_flags[0] = 1;
  c->pal_host_flags[1] = 0;
  if (!c->opt.palette) return PL_OK;
  if (!c->pal_keys.p) {
    PL_HIP(c->pal_keys.alloc(pl::kPalSize));
    PL_HIP(c->pal_owner.alloc(pl::kPalSize));
    PL_HIP(c->pal_flags.alloc(2));
    PL_HIP(c->pal_id.alloc(c->B));
    PL_HIP(c->palette.alloc(pl::kPalSize));
    void *pinned = nullptr;
    PL_HIP(hipHostMalloc(&pinned, 2 * sizeof(int), hipHostMallocDefault));
    c->pal_host_flags = static_cast<int *>(pinned);
    c->pal_host_flags[0] = 1;
    c->pal_host_flags[1] = 0;
  }
  PL_HIP(hipMemsetAsync(c->pal_keys.p, 0xFF, pl::kPalSize * sizeof(unsigned long long), st));
  PL_HIP(hipMemsetAsync(c->pal_owner.p, 0x7F, pl::kPalSize * sizeof(int), st));
  PL_HIP(hipMemsetAsync(c->pal_flags.p, 0, 2 * sizeof(int), st));
  PL_HIP(hipMemsetAsync(c->palette.p, 0, pl::kPalSize * sizeof(pl::Record), st));
  const dim3 g(grid_for(c->B)), blk(pl::kBlock);
  hipLaunchKernelGGL(pl::k_pal_insert, g, blk, 0, st, c->B, c->rec.p, c->pal_keys.p, c->pal_owner.p, c->pal_id.p,
                     c->pal_flags.p);
  hipLaunchKernelGGL(pl::k_pal_publish, g, blk, 0, st, c->B, c->rec.p, c->pal_owner.p, c->pal_id.p, c->palette.p);
  hipLaunchKernelGGL(pl::k_pal_verify, g, blk, 0, st, c->B, c->rec.p, c->pal_id.p, c->palette.p, c->pal_owner.p,
                     c->pal_flags.p);
  PL_HIP(hipGetLastError());
  PL_HIP(hipMemcpyAsync(c->pal_host_flags, c->pal_flags.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  return PL_OK;
}
void finish_palette(pl_context *c) {
  if (!c->opt.palette) return;
  c->pal_entries = c->pal_host_flags[1];
  c->pal_ready = (c->pal_host_flags[0] == 0);
}
int build_palette(pl_context *c) {
  int rc = launch_palette(c, c->stream);
  if (rc) return rc;
  PL_HIP(hipStreamSynchronize(c->stream));
  finish_palette(c);
  return PL_OK;
}

int launch_diag(pl_context *c, hipStream_t st) {
  const unsigned g = grid_for(c->n_slices, pl::kBlock / pl::kWave);
  const uint8_t *fb = c->have_bc ? c->fixedbits.p : (const uint8_t *)nullptr;
#define PL_D(L)                                                                                                  \
  hipLaunchKernelGGL((pl::k_diag_gather<L>), dim3(g), dim3(pl::kBlock), 0, st, c->N, c->slice_ptr.p, c->ent.p, \
                     c->rec.p, fb, c->diag.p, c->dinv.p)
  switch (c->lpn) { case 1: PL_D(1); break; case 2: PL_D(2); break; case 4: PL_D(4); break; case 8: PL_D(8); break;
                    default: PL_D(16); }
#undef PL_D
  PL_HIP(hipGetLastError());
  return PL_OK;
}
// multi-GPU: the diagonal of shared nodes is the sum over ranks; then invert again (main stream: RCCL)
int finish_diag_dist(pl_context *c) {
  if (!c->dist.active) return PL_OK;
  int rc = pl::dist_sum_shared(c->dist, c->diag.p, c->stream);
  if (rc) return fail(PL_ERR_HIP, "RCCL all-reduce of the Jacobi diagonal failed");
  pl::launch_invert_diag(c->N * 6, c->diag.p, c->have_bc ? c->fixed.p : nullptr, c->dinv.p, c->stream);
  return PL_OK;
}

// Dirichlet mask OR "shared with another rank" (all six dofs): the mask of the rank-local levels
__global__ void k_local_mask(int64_t N, const uint8_t *__restrict__ fixedbits, const uint8_t *__restrict__ shared,
                             uint8_t *__restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) mask[i] = (uint8_t)(fixedbits[i] | (shared[i] ? 0x3f : 0));
}

// fp32 copy of the Jacobi inverse for the multi-level PCG kernels (main stream, after the diagonal is final)
int launch_dinv32(pl_context *c) {
  if (!c->coarse.enabled) return PL_OK;
  hipLaunchKernelGGL(pl::k_to_float, dim3(grid_for(c->N * 6)), dim3(pl::kBlock), 0, c->stream, c->N * 6, c->dinv.p,
                     c->coarse.dinv32);
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// The rank-local levels (tile blocks on several GPUs, local dense level) see shared nodes as constrained.
int launch_local_mask(pl_context *c) {
  if (!c->coarse.enabled || !(c->dist.active || c->coarseL.enabled)) return PL_OK;
  hipLaunchKernelGGL(k_local_mask, dim3(grid_for(c->N)), dim3(pl::kBlock), 0, c->stream, c->N, c->fixedbits.p,
                     c->sharedbits.p, c->maskL.p);
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// modes of the tile level in use (not with the rank-local level of precond = 4); decided the same way when the blocks are
// built and when they are applied.  With a communicator the strain modes, like the rigid ones, leave out the nodes shared
// with other ranks.
inline int tile_modes_now(const pl_context *c) {
  return (c->coarse.tile_level && c->coarse.tile_modes == 12 && !c->coarseL.enabled) ? 12 : 6;
}

int launch_tile_blocks(pl_context *c, hipStream_t st) {
  pl::Coarse &cs = c->coarse;
  if (!cs.enabled || !c->have_bc || !cs.tile_level) return PL_OK;
  if (cs.cm == 12) return PL_OK;        // built inside build_coarse_level, which needs them first
  const uint8_t *fb = c->dist.active ? c->maskL.p : c->fixedbits.p;
  const bool twelve = tile_modes_now(c) == 12;
  hipLaunchKernelGGL(pl::k_tile_blocks, dim3((unsigned)cs.n_tiles), dim3(pl::kBlock), 0, st, c->tile.tile_start.p,
                     c->tile.home_ptr.p, c->tile.foreign_ptr.p, c->tile.foreign_idx.p,
                     reinterpret_cast<const int2 *>(c->conn.p), c->rec.p, cs.agg_of_tile.p, cs.cen.p, c->xyz.p,
                     fb, cs.Bt_inv, twelve ? cs.Bt_raw : (double *)nullptr);
  if (twelve) {
    hipLaunchKernelGGL(pl::k_tile_blocks_strain, dim3((unsigned)cs.n_tiles), dim3(pl::kBlock), 0, st,
                       c->tile.tile_start.p, c->tile.home_ptr.p, c->tile.foreign_ptr.p, c->tile.foreign_idx.p,
                       reinterpret_cast<const int2 *>(c->conn.p), c->rec.p, cs.agg_of_tile.p, cs.cen.p, c->xyz.p, fb,
                       cs.Bt_raw);
    hipLaunchKernelGGL(pl::k_tile_invert12, dim3(grid_for(cs.n_tiles, pl::kInv12Block)), dim3(pl::kInv12Block), 0, st, cs.n_tiles,
                       (const double *)cs.Bt_raw, cs.Bt_inv);
  }
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// A_c = Z^T P K P Z on the device, then its Cholesky factor and W = L^-1 (pl_dense.h).
int build_coarse_level(pl_context *c, pl::Coarse &cs, const uint8_t *mask, bool reduce,
                       const std::function<void()> &after_chol = nullptr) {
  cs.ready = false;
  if (!cs.enabled || !c->have_bc) {
    if (after_chol) after_chol();
    return PL_OK;
  }
  bool tile_invert_pending = false;
  const int n = cs.ncp;
  if (!cs.ac_clean) PL_HIP(hipMemsetAsync(cs.Ac, 0, (size_t)n * n * sizeof(double), c->stream));
  cs.ac_clean = false;
  PL_HIP(hipMemsetAsync(cs.info, 0, 2 * sizeof(int), c->stream));
  if (cs.cm == 12) {
    // 12 modes per aggregate: A_c from the 12 x 12 tile blocks (built HERE, ahead of the factorisation that needs them,
    // instead of beside it on the side stream) plus the cross-tile struts
    const dim3 gt((unsigned)cs.n_tiles), blk(pl::kBlock);
    const int2 *conn2 = reinterpret_cast<const int2 *>(c->conn.p);
    // (the rigid x rigid part on a second stream beside the strain rows and the cross-tile struts; the inversion of the
    // tile blocks, which only the solve needs, beside the factorisation)
    // Several GPUs: A_c is the sum over ranks of what each rank's struts give on ALL nodes (Dirichlet mask only), while the
    // tile LEVEL lives on this rank's own nodes (mask | shared): two sets of tile blocks, the second one for the level.
    const bool two_sets = c->dist.active;
    if (two_sets) {
      if (!cs.Bt_rawA && hipMalloc((void **)&cs.Bt_rawA, (size_t)cs.n_tiles * 144 * sizeof(double)) != hipSuccess)
        return fail(PL_ERR_HIP, "pl_assemble: out of device memory for the tile blocks");
      hipLaunchKernelGGL(pl::k_tile_blocks, gt, blk, 0, c->stream, c->tile.tile_start.p, c->tile.home_ptr.p,
                         c->tile.foreign_ptr.p, c->tile.foreign_idx.p, conn2, c->rec.p, cs.agg_of_tile.p, cs.cen.p,
                         c->xyz.p, (const uint8_t *)c->maskL.p, cs.Bt_inv, cs.Bt_raw);
      hipLaunchKernelGGL(pl::k_tile_blocks_strain, gt, blk, 0, c->stream, c->tile.tile_start.p, c->tile.home_ptr.p,
                         c->tile.foreign_ptr.p, c->tile.foreign_idx.p, conn2, c->rec.p, cs.agg_of_tile.p, cs.cen.p,
                         c->xyz.p, (const uint8_t *)c->maskL.p, cs.Bt_raw);
    }
    double *rawA = two_sets ? cs.Bt_rawA : cs.Bt_raw;
    PL_HIP(hipEventRecord(c->ev_t0, c->stream));
    PL_HIP(hipStreamWaitEvent(c->side2, c->ev_t0, 0));
    hipLaunchKernelGGL(pl::k_tile_blocks, gt, blk, 0, c->side2, c->tile.tile_start.p, c->tile.home_ptr.p,
                       c->tile.foreign_ptr.p, c->tile.foreign_idx.p, conn2, c->rec.p, cs.agg_of_tile.p, cs.cen.p, c->xyz.p,
                       mask, cs.Bt_inv, rawA);
    PL_HIP(hipEventRecord(c->ev_t1, c->side2));
    hipLaunchKernelGGL(pl::k_tile_blocks_strain, gt, blk, 0, c->stream, c->tile.tile_start.p, c->tile.home_ptr.p,
                       c->tile.foreign_ptr.p, c->tile.foreign_idx.p, conn2, c->rec.p, cs.agg_of_tile.p, cs.cen.p, c->xyz.p,
                       mask, rawA);
    if (cs.n_cross > 0)
      hipLaunchKernelGGL(pl::k_coarse_cross12, dim3(grid_for(cs.n_cross)), blk, 0, c->stream, cs.n_cross,
                         cs.cross_idx.p, c->conn.p, c->rec.p, cs.agg_of_node.p, cs.cen.p, c->xyz.p, mask, n, cs.Ac);
    PL_HIP(hipStreamWaitEvent(c->stream, c->ev_t1, 0));
    hipLaunchKernelGGL(pl::k_agg_add_tiles, dim3(grid_for(cs.n_tiles * 144)), blk, 0, c->stream, cs.n_tiles,
                       cs.agg_of_tile.p, (const double *)rawA, n, cs.Ac);
    PL_HIP(hipEventRecord(c->ev_t0, c->stream));
    PL_HIP(hipStreamWaitEvent(c->side2, c->ev_t0, 0));
    hipLaunchKernelGGL(pl::k_tile_invert12, dim3(grid_for(cs.n_tiles, pl::kInv12Block)), dim3(pl::kInv12Block), 0,
                       c->side2, cs.n_tiles, (const double *)cs.Bt_raw, cs.Bt_inv);
    PL_HIP(hipEventRecord(c->ev_t1, c->side2));
    tile_invert_pending = true;
  } else {
  if (cs.n_fix < 0) {   // Dirichlet set changed: list the in-aggregate struts that touch it, grouped by aggregate
    if (!cs.fix_count) PL_HIP(hipMalloc((void **)&cs.fix_count, sizeof(int)));
    int cnt = 0;
    DevBuf<int64_t> keys;
    for (int pass = 0; pass < 2; ++pass) {
      PL_HIP(hipMemsetAsync(cs.fix_count, 0, sizeof(int), c->stream));
      hipLaunchKernelGGL(pl::k_list_fixed_struts, dim3(grid_for(c->B)), dim3(pl::kBlock), 0, c->stream, c->B,
                         c->conn.p, cs.agg_of_node.p, mask, pass ? keys.p : (int64_t *)nullptr, cs.fix_count);
      if (pass == 0) {
        PL_HIP(hipMemcpyAsync(&cnt, cs.fix_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        PL_HIP(hipStreamSynchronize(c->stream));
        if (cnt == 0) break;
        PL_HIP(keys.alloc((size_t)cnt));
      }
    }
    std::vector<int32_t> list;
    if (cnt > 0) {
      std::vector<int64_t> hk((size_t)cnt);
      PL_HIP(hipMemcpyAsync(hk.data(), keys.p, hk.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
      PL_HIP(hipStreamSynchronize(c->stream));
      std::sort(hk.begin(), hk.end());
      list.reserve(hk.size() + hk.size() / 4 + pl::kWave);
      for (size_t q = 0; q < hk.size(); ++q) {
        if (q > 0 && (hk[q] >> 32) != (hk[q - 1] >> 32))
          while (list.size() % pl::kWave) list.push_back(-1);
        list.push_back((int32_t)(hk[q] & 0xffffffffLL));
      }
      while (list.size() % pl::kWave) list.push_back(-1);
      PL_HIP(cs.fix_list.upload(list));
    }
    cs.n_fix = (int64_t)list.size();
  }
  if (cs.n_fix > 0)
    hipLaunchKernelGGL(pl::k_coarse_assemble, dim3(grid_for(cs.n_fix)), dim3(pl::kBlock), 0, c->stream, cs.n_fix,
                       cs.fix_list.p, c->conn.p, c->rec.p, cs.agg_of_node.p, cs.cen.p, c->xyz.p, mask, n,
                       cs.Ac);
  if (cs.n_cross > 0)
    hipLaunchKernelGGL(pl::k_coarse_assemble_cross, dim3(grid_for(cs.n_cross)), dim3(pl::kBlock), 0, c->stream,
                       cs.n_cross, cs.cross_idx.p, c->conn.p, c->rec.p, cs.agg_of_node.p, cs.cen.p, c->xyz.p,
                       mask, n, cs.Ac);
  }
  if (reduce && c->dist.active) {   // every rank holds the contribution of ITS struts; all ranks then factor the same matrix
    const int nb = n / pl::kNB;
    if (cs.bw_blocks > 0 && cs.bw_blocks + 1 < nb) {   // only the block band of the lower triangle travels (L_f is free)
      const int64_t cnt = (int64_t)n * (cs.bw_blocks + 1) * pl::kNB;
      hipLaunchKernelGGL(pl::k_band_pack, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, n, n,
                         cs.bw_blocks, cs.Ac, cs.Lf);
      if (pl::dist_sum_scalars(c->dist, cs.Lf, (int)cnt, c->stream))
        return fail(PL_ERR_HIP, "RCCL all-reduce of the coarse operator failed");
      hipLaunchKernelGGL(pl::k_band_unpack, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, n, n,
                         cs.bw_blocks, cs.Lf, cs.Ac);
    } else if (pl::dist_sum_scalars(c->dist, cs.Ac, n * n, c->stream)) {
      return fail(PL_ERR_HIP, "RCCL all-reduce of the coarse operator failed");
    }
  }
  hipLaunchKernelGGL(pl::k_coarse_regularize, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, cs.Ac);
  pl::dense_factor_inverse(cs.Ac, cs.Lf, cs.W, cs.Wt, cs.Dinv, n, n, cs.info, cs.bw_blocks, c->stream, after_chol,
                           c->opt.chol_persistent ? cs.bar : (unsigned *)nullptr);
  if (tile_invert_pending) PL_HIP(hipStreamWaitEvent(c->stream, c->ev_t1, 0));
  PL_HIP(hipGetLastError());
  int info[2] = {0, 0};
  PL_HIP(hipMemcpyAsync(info, cs.info, sizeof(info), hipMemcpyDeviceToHost, c->stream));
  PL_HIP(hipStreamSynchronize(c->stream));
  if (&cs == &c->coarse) c->coarse_info = info[0];
  cs.ready = (info[0] == 0);   // not SPD -> fall back to Jacobi
  return PL_OK;
}

int build_coarse(pl_context *c, const std::function<void()> &after_chol = nullptr) {
  int rc = build_coarse_level(c, c->coarse, c->fixedbits.p, true, after_chol);
  if (rc || !c->coarseL.enabled) return rc;
  // (mask = Dirichlet | shared, launch_local_mask: its modes live on this rank's own nodes only)
  return build_coarse_level(c, c->coarseL, c->maskL.p, false);
}

inline const double *cinv(const pl_context *c) { return c->cls_ready ? c->cls_table.p : c->kcc_inv.p; }
inline const uint16_t *ccls(const pl_context *c) { return c->cls_ready ? c->cls_id.p : (const uint16_t *)nullptr; }

// The nodes eliminated in this solve: the candidates of pl_create (an independent set of the node graph, so K_cc is block
// diagonal) that carry no Dirichlet dof and are not shared with another rank.
int select_condensed(pl_context *c, const std::vector<uint8_t> &bits) {
  c->cond_ready = false;
  c->n_cond = 0;
  // (single-GPU handles only for now: the two passes of the condensed operator would each need the interface exchange)
  const bool wanted = c->opt.condense >= 0 && c->opkind == 0 && c->coarse.enabled && c->opt.precision != 2 &&
                      !c->h_cand.empty() && !c->dist.active;
  if (!wanted) return PL_OK;
  const int64_t N = c->N;
  std::vector<int32_t> picked;
  for (int64_t i = 0; i < N; ++i)
    if (c->h_cand[i] && !bits[i] && (c->h_shared.empty() || !c->h_shared[i])) picked.push_back((int32_t)i);
  if (picked.empty()) return PL_OK;
  std::vector<uint8_t> flag((size_t)N, 0), mask(bits);
  for (int32_t i : picked) {
    flag[i] = 1;
    mask[i] = 0x3f;
  }
  PL_HIP(c->cnodes.alloc(picked.size()));
  PL_HIP(c->kcc_inv.alloc(picked.size() * 36));
  if (!c->maskC.p) {
    PL_HIP(c->maskC.alloc(N));
    PL_HIP(c->cflag.alloc(N));
  }
  PL_HIP(hipMemcpy(c->cnodes.p, picked.data(), picked.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  if (!c->cbase.p) PL_HIP(c->cbase.alloc(N));
  c->cbase_state = -1;
  PL_HIP(hipMemcpy(c->maskC.p, mask.data(), N, hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(c->cflag.p, flag.data(), N, hipMemcpyHostToDevice));
  // per strut: which ends are condensed (read by every pass of the condensed operator)
  if (!c->cend.p) PL_HIP(c->cend.alloc(c->B));
  hipLaunchKernelGGL(k_cond_ends, dim3(grid_for(c->B)), dim3(pl::kBlock), 0, c->stream, c->B,
                     reinterpret_cast<const int2 *>(c->conn.p), (const uint8_t *)c->cflag.p, c->cend.p);
  PL_HIP(hipGetLastError());
  c->n_cond = (int64_t)picked.size();
  return PL_OK;
}
// K_cc^-1 of the condensed nodes from the current records (after launch_records, same stream)
int launch_condensed_blocks(pl_context *c, hipStream_t st) {
  c->cond_ready = false;
  if (c->n_cond <= 0 || !c->have_bc) return PL_OK;
  hipLaunchKernelGGL(pl::k_node_block_inverse, dim3(grid_for(c->n_cond)), dim3(pl::kBlock), 0, st, c->n_cond,
                     c->cnodes.p, pl::kWave / c->lpn, c->slice_ptr.p, c->ent.p, c->rec.p, c->kcc_inv.p);
  PL_HIP(hipGetLastError());
  c->cond_ready = true;
  c->cls_ready = false;
  c->cbase_state = -1;       // class ids are assigned anew below
  if (c->opt.palette && c->pal_id.p) {
    // Classes by the record-palette ids of the incident struts (queued behind launch_palette on the same stream; if the
    // record palette turns out not to hold, finish_condensed_classes drops the classes as well)
    if (!c->cls_table.p) {
      PL_HIP(c->cls_keys.alloc(65536));
      PL_HIP(c->cls_owner.alloc(65536));
      PL_HIP(c->cls_flags.alloc(1));
      PL_HIP(c->cls_table.alloc((size_t)65536 * 36));
      void *pinned = nullptr;
      PL_HIP(hipHostMalloc(&pinned, sizeof(int), hipHostMallocDefault));
      c->cls_host_flag = static_cast<int *>(pinned);
    }
    if (c->cls_key.n < (size_t)c->n_cond) {
      PL_HIP(c->cls_key.alloc((size_t)c->n_cond));
      PL_HIP(c->cls_id.alloc((size_t)c->n_cond));
    }
    *c->cls_host_flag = 1;
    PL_HIP(hipMemsetAsync(c->cls_keys.p, 0xFF, 65536 * sizeof(unsigned long long), st));
    PL_HIP(hipMemsetAsync(c->cls_owner.p, 0x7F, 65536 * sizeof(int), st));
    PL_HIP(hipMemsetAsync(c->cls_flags.p, 0, sizeof(int), st));
    const dim3 g(grid_for(c->n_cond)), blk(pl::kBlock);
    hipLaunchKernelGGL(pl::k_cls_hash, g, blk, 0, st, c->n_cond, c->cnodes.p, pl::kWave / c->lpn, c->slice_ptr.p,
                       c->ent.p, c->pal_id.p, c->cls_key.p);
    hipLaunchKernelGGL(pl::k_cls_insert, g, blk, 0, st, c->n_cond, c->cls_key.p, c->cls_keys.p, c->cls_owner.p,
                       c->cls_id.p, c->cls_flags.p);
    hipLaunchKernelGGL(pl::k_cls_publish, dim3(grid_for(c->n_cond * 36)), blk, 0, st, c->n_cond, c->kcc_inv.p,
                       c->cls_owner.p, c->cls_id.p, c->cls_table.p);
    hipLaunchKernelGGL(pl::k_cls_verify, g, blk, 0, st, c->n_cond, c->kcc_inv.p, c->cls_id.p, c->cls_table.p,
                       c->cls_flags.p);
    PL_HIP(hipGetLastError());
    PL_HIP(hipMemcpyAsync(c->cls_host_flag, c->cls_flags.p, sizeof(int), hipMemcpyDeviceToHost, st));
  }
  return PL_OK;
}
// after the stream has drained and finish_palette() has run
void finish_condensed_classes(pl_context *c) {
  c->cls_ready = c->cond_ready && c->cls_host_flag && *c->cls_host_flag == 0 && c->pal_ready && c->opt.palette;
  if (c->cls_host_flag) *c->cls_host_flag = 1;
}

int launch_bsr_fill(pl_context *c, int with_bc, hipStream_t st) {
  const uint8_t *fb = c->have_bc ? c->fixedbits.p : (const uint8_t *)nullptr;
  const size_t lds = (size_t)(pl::kBsrBlock / pl::kWave) * 64 * pl::kBsrPitch * sizeof(double);   // 38 KB
  const unsigned gb = grid_for(c->n_slices, pl::kBsrBlock / pl::kWave);
#define PL_B(L)                                                                                                  \
  hipLaunchKernelGGL((pl::k_bsr_fill<L>), dim3(gb), dim3(pl::kBsrBlock), lds, st, c->N, c->slice_ptr.p, c->ent.p,  \
                     c->rec.p, c->bsr_rowptr.p, c->ent_slot.p, c->diag_slot.p, fb, with_bc, c->bsr_vals.p)
  switch (c->lpn) { case 1: PL_B(1); break; case 2: PL_B(2); break; case 4: PL_B(4); break; case 8: PL_B(8); break;
                    default: PL_B(16); }
#undef PL_B
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// Everything of a two-level PCG iteration after K*p: update + restriction, coarse solve, new direction.
template <typename PT, typename RT>
int pcg_tail_coarse_t(pl_context *c, double *cur, double *nxt, int hist_slot, PT *p, const PT *Ap, RT *x, RT *r) {
  pl::Coarse &cs = c->coarse, &cl = c->coarseL;
  const bool useL = cl.ready;
#define PL_UPD(TM, MULTI, LOCAL)                                                                                    \
  hipLaunchKernelGGL((pl::k_pcg_update_tile<PT, RT, TM, MULTI, LOCAL>), dim3((unsigned)cs.n_tiles), dim3(cs.vblock), \
                     0, c->stream,                                                                                   \
                     c->tile.tile_start.p, cs.agg_of_tile.p, cs.cen.p, c->xyz.p, Ap, cs.dinv32,                       \
                     c->dist.active ? (const double *)c->dist.weight.p : (const double *)nullptr, r, cur,             \
                     cs.rc, cs.tile_level ? (const double *)cs.Bt_inv : (const double *)nullptr, cs.yt,               \
                     useL ? (const int32_t *)cl.agg_of_tile.p : (const int32_t *)nullptr, cl.cen.p,                   \
                     (c->dist.active || useL) ? (const uint8_t *)c->sharedbits.p : (const uint8_t *)nullptr, cl.rc,   \
                     cs.ncp, c->cond_use ? (const uint8_t *)c->cflag.p : (const uint8_t *)nullptr, cs.cm)
  if (useL) {
    if (tile_modes_now(c) == 12) PL_UPD(12, true, true);
    else PL_UPD(6, true, true);
  } else if (c->dist.active) {
    if (tile_modes_now(c) == 12) PL_UPD(12, true, false);
    else PL_UPD(6, true, false);
  } else if (tile_modes_now(c) == 12) PL_UPD(12, false, false);
  else PL_UPD(6, false, false);
#undef PL_UPD
  if (useL)   // rank-local level: y_L is never communicated, but its share of r.z, r_L . A_L^-1 r_L, is a per-rank
              // partial sum: it joins the r.D^-1 r slots BEFORE they travel in the collective below
    pl::dense_apply(cl.W, cl.Wt, cl.ncp, cl.ncp, cl.rc, cl.tv, cl.yc, cs.rc + cs.ncp + pl::kSlots,
                    (const double *)nullptr, c->stream);
  if (c->dist.active) {   // one collective: [Z^T r | r.r slots | r.D^-1 r slots]; the coarse solve is then redundant per rank
    if (pl::dist_sum_scalars(c->dist, cs.rc, cs.ncp + 2 * pl::kSlots, c->stream))
      return fail(PL_ERR_HIP, "RCCL all-reduce of the coarse residual failed");
  }
  pl::dense_apply(cs.W, cs.Wt, cs.ncp, cs.ncp, cs.rc, cs.tv, cs.yc, cur + pl::S_RZ_NEW * pl::kSlots,
                  cs.rc + cs.ncp + pl::kSlots, c->stream);
#define PL_DIR(TM, MULTI, LOCAL)                                                                                         \
  hipLaunchKernelGGL((pl::k_pcg_direction_coarse<PT, RT, TM, MULTI, LOCAL>), dim3((unsigned)cs.n_tiles), dim3(cs.vblock), \
                     0, c->stream,                                                                                        \
                     c->tile.tile_start.p, (const RT *)r, cs.dinv32, c->xyz.p, cs.agg_of_tile.p, cs.cen.p, cs.yc,         \
                     cs.tile_level ? (const double *)cs.yt : (const double *)nullptr, c->fixedbits.p, p, x, cur, nxt,     \
                     c->hist.p, hist_slot, cs.rc, cs.ncp,                                                                \
                     useL ? (const int32_t *)cl.agg_of_tile.p : (const int32_t *)nullptr, cl.cen.p, cl.yc,                \
                     (c->dist.active || useL) ? (const uint8_t *)c->sharedbits.p : (const uint8_t *)nullptr, cl.rc,       \
                     cl.ncp, c->cond_use ? (const uint8_t *)c->cflag.p : (const uint8_t *)nullptr, cs.cm)
#define PL_DIRF(TM)                                                                                                      \
  hipLaunchKernelGGL((pl::k_pcg_direction_flat<PT, RT, TM>), dim3((unsigned)((3 * c->N + pl::kBlock - 1) / pl::kBlock)),   \
                     dim3(pl::kBlock), 0, c->stream, c->N, cs.tile_of_node.p, (const RT *)r, cs.dinv32, c->xyz.p,         \
                     cs.agg_of_tile.p, cs.cen.p, cs.yc, cs.tile_level ? (const double *)cs.yt : (const double *)nullptr,  \
                     c->fixedbits.p, p, x, cur, nxt, c->hist.p, hist_slot, cs.rc, cs.ncp,                                \
                     c->cond_use ? (const uint8_t *)c->cflag.p : (const uint8_t *)nullptr, cs.cm,                      \
                     c->dist.active ? (const uint8_t *)c->sharedbits.p : (const uint8_t *)nullptr)
  // flat mapping, contiguous per wave: fp64 p without a rank-local level (measured on one box, 50^3 Octet: 26.2 -> 24.5 us;
  // fp32 p / fp64 r the same either way, fp32 p / fp32 r 19.0 -> 22.0 us - 8-byte loads per lane are too few in flight)
  if (!useL && sizeof(PT) == 8) {
    if (tile_modes_now(c) == 12) PL_DIRF(12);
    else PL_DIRF(6);
  } else if (useL) {
    if (tile_modes_now(c) == 12) PL_DIR(12, true, true);
    else PL_DIR(6, true, true);
  } else if (c->dist.active) {
    if (tile_modes_now(c) == 12) PL_DIR(12, true, false);
    else PL_DIR(6, true, false);
  } else if (tile_modes_now(c) == 12) PL_DIR(12, false, false);
  else PL_DIR(6, false, false);
#undef PL_DIRF
#undef PL_DIR
  PL_HIP(hipGetLastError());
  return PL_OK;
}

int pcg_tail_coarse(pl_context *c, double *cur, double *nxt, int hist_slot) {
  return pcg_tail_coarse_t<double, double>(c, cur, nxt, hist_slot, c->p.p, (const double *)c->Ap.p, c->x.p, c->r.p);
}

// One PCG iteration (k = iteration index: selects the scalar set by parity and the residual-history slot).
int pcg_iteration(pl_context *c, int k) {
  const int64_t n6 = c->N * 6;
  const int set = pl::S_COUNT * pl::kSlots;
  double *cur = c->scal.p + (k & 1) * set, *nxt = c->scal.p + ((k + 1) & 1) * set;
  if (c->cond_use) {
    // S p: the condensed nodes take their equilibrium position under p (first pass, their rows of p are 0 on entry),
    // then the ordinary product with their rows masked like Dirichlet rows (second pass, with p.Ap)
    // (first pass fused with the 6 x 6 solves: every tile writes -K_cc^-1 (K p_v)_c into the p rows of its condensed nodes)
    int rc = launch_spmv(c, c->p.p, c->p.p, false, nullptr, nullptr, pl::kEndsCondensedSolve);
    if (rc) return rc;
    rc = launch_spmv(c, c->p.p, c->Ap.p, true, cur + pl::S_PAP * pl::kSlots, c->maskC.p, pl::kEndsOthers);
    if (rc) return rc;
    return pcg_tail_coarse(c, cur, nxt, k);
  }
  int rc = launch_spmv(c, c->p.p, c->Ap.p, true, cur + pl::S_PAP * pl::kSlots);
  if (rc) return rc;
  if (c->coarse.ready) return pcg_tail_coarse(c, cur, nxt, k);
  // reference-CG mode (conjugate_gradient_solver.py:79-109): every restart_every-th iteration the direction is rebuilt
  // on the PREVIOUS z (with a preconditioner; kept in tmp) or on the updated residual (without one: z aliases r there)
  const bool ref = ref_cg(c) && !c->dist.active;
  const bool restart = ref && c->opt.restart_every > 0 && k > 0 && (k % c->opt.restart_every) == 0;
  const bool has_M = c->dd_ready || c->opt.precond >= 1;
  const double *pn = c->p.p, *psrc = nullptr;
  if (restart) {
    if (has_M) {
      PL_HIP(hipMemcpyAsync(c->tmp.p, c->z.p, n6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      pn = psrc = c->tmp.p;
    } else {
      pn = nullptr;        // ||r_new||
      psrc = c->z.p;       // = r_new once the update kernel has run (dinv = 1 on free dofs)
    }
  }
  const int hcap = ref ? c->hist_cap : 0;
  if (c->dd_ready) {   // DDM with the factorised assembled matrix: update leaves z = 0, r.z = 0; then z = G^-1 r
    if (ref)
      hipLaunchKernelGGL(pl::k_pcg_update<true>, dim3(grid_stream(n6 / 2)), dim3(pl::kBlock), 0, c->stream, n6, c->p.p,
                         c->Ap.p, c->dinv.p, c->x.p, c->r.p, c->z.p, cur, c->opt.alpha_max, pn);
    else
      hipLaunchKernelGGL(pl::k_pcg_update<false>, dim3(grid_stream(n6 / 2)), dim3(pl::kBlock), 0, c->stream, n6, c->p.p,
                         c->Ap.p, c->dinv.p, c->x.p, c->r.p, c->z.p, cur, c->opt.alpha_max, (const double *)nullptr);
    pl::dense_apply(c->dd_W.p, c->dd_Wt.p, c->dd_n, c->dd_n, c->r.p, c->dd_tv.p, c->z.p,
                    cur + pl::S_RZ_NEW * pl::kSlots, (const double *)nullptr, c->stream);
    hipLaunchKernelGGL(pl::k_pcg_direction, dim3(grid_stream(n6 / 2)), dim3(pl::kBlock), 0, c->stream, n6, c->z.p,
                       c->p.p, cur, nxt, c->hist.p, k, psrc, hcap);
    PL_HIP(hipGetLastError());
    return PL_OK;
  }
  if (c->dist.active) {
    pl::launch_pcg_update_weighted(n6, c->p.p, c->Ap.p, c->dinv.p, c->dist.weight.p, c->x.p, c->r.p, c->z.p, cur,
                                   c->stream);
    if (pl::dist_sum_scalars(c->dist, cur + pl::S_RZ_NEW * pl::kSlots, 2 * pl::kSlots, c->stream))
      return fail(PL_ERR_HIP, "RCCL all-reduce of the PCG scalars failed");
  } else if (ref) {
    hipLaunchKernelGGL(pl::k_pcg_update<true>, dim3(grid_stream(n6 / 2)), dim3(pl::kBlock), 0, c->stream, n6, c->p.p,
                       c->Ap.p, c->dinv.p, c->x.p, c->r.p, c->z.p, cur, c->opt.alpha_max, pn);
  } else {
    hipLaunchKernelGGL(pl::k_pcg_update<false>, dim3(grid_stream(n6 / 2)), dim3(pl::kBlock), 0, c->stream, n6, c->p.p,
                       c->Ap.p, c->dinv.p, c->x.p, c->r.p, c->z.p, cur, c->opt.alpha_max, (const double *)nullptr);
  }
  hipLaunchKernelGGL(pl::k_pcg_direction, dim3(grid_stream(n6 / 2)), dim3(pl::kBlock), 0, c->stream, n6, c->z.p,
                     c->p.p, cur, nxt, c->hist.p, k, psrc, hcap);
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// Solve P K P x = rhs (device rhs already masked), x0 = 0.  Result in c->x.  Returns iterations through stats.
int pcg_solve(pl_context *c, const double *f_dev, const double *Kubar_dev, double rtol, int max_iter,
              pl_stats_t *st) {
  const int64_t n6 = c->N * 6;
  int rc = ensure_hist(c, max_iter + 1);
  if (rc) return rc;
  PL_HIP(hipMemsetAsync(c->scal.p, 0, 2 * pl::S_COUNT * pl::kSlots * sizeof(double), c->stream));
  if (c->dist.active)
    pl::launch_pcg_init_weighted(n6, f_dev, Kubar_dev, c->fixed.p, c->dinv.p, c->dist.weight.p, c->x.p, c->r.p,
                                 c->z.p, c->p.p, c->scal.p, c->stream);
  else
    hipLaunchKernelGGL(pl::k_pcg_init, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6, f_dev, Kubar_dev,
                       c->fixed.p, c->dinv.p, c->x.p, c->r.p, c->z.p, c->p.p, c->scal.p);
  PL_HIP(hipGetLastError());
  if (c->dist.active) {
    if (pl::dist_sum_scalars(c->dist, c->scal.p + pl::S_RZ_OLD * pl::kSlots, pl::kSlots, c->stream) ||
        pl::dist_sum_scalars(c->dist, c->scal.p + pl::S_BB * pl::kSlots, pl::kSlots, c->stream))
      return fail(PL_ERR_HIP, "RCCL all-reduce of the initial PCG scalars failed");
  }
  if (c->dd_ready) {   // z0 = p0 = G^-1 r0, rz_old = r0.z0 (k_pcg_init ran with dinv = 0)
    pl::dense_apply(c->dd_W.p, c->dd_Wt.p, c->dd_n, c->dd_n, c->r.p, c->dd_tv.p, c->z.p,
                    c->scal.p + pl::S_RZ_OLD * pl::kSlots, (const double *)nullptr, c->stream);
    PL_HIP(hipMemcpyAsync(c->p.p, c->z.p, n6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  if (c->cond_ready && c->coarse.ready) {
    // start from the iterate whose condensed nodes are in equilibrium: x_c = K_cc^-1 r_c, r <- r - K x (rows of the
    // condensed nodes become exactly 0 and stay 0: every later step keeps them in equilibrium)
    // t_c = K_cc^-1 b_c (rows of z, zero elsewhere), r_v -= (K t)_v: the load the eliminated nodes pass on.  Their own
    // rows of r keep b_c, their rows of x stay 0 until the back-substitution after the loop.
    PL_HIP(hipMemsetAsync(c->z.p, 0, n6 * sizeof(double), c->stream));
    hipLaunchKernelGGL(pl::k_condense_solve<double>, dim3(grid_for(c->n_cond * 6)), dim3(pl::kBlock), 0, c->stream,
                       c->n_cond, c->cnodes.p, cinv(c), ccls(c), (const double *)c->r.p, c->z.p, 1.0);
    rc = launch_spmv(c, c->z.p, c->tmp2.p, true, nullptr, c->maskC.p, pl::kEndsOthers);
    if (rc) return rc;
    hipLaunchKernelGGL(pl::k_condense_subtract<double>, dim3(grid_for(n6)), dim3(pl::kBlock), 0, c->stream, c->N, c->cflag.p,
                       (const double *)c->tmp2.p, c->r.p);
    PL_HIP(hipGetLastError());
  }
  if (c->coarse.ready) {
    // z0 = M^-1 r0 needs the coarse solve: run the tail of an iteration "-1" with p = 0, alpha = 0 (p.Ap = 0) on
    // scalar set 1; its direction kernel leaves p = z0 and rz_old = r0.z0 in set 0, where iteration 0 starts.
    const int set = pl::S_COUNT * pl::kSlots;
    PL_HIP(hipMemsetAsync(c->p.p, 0, n6 * sizeof(double), c->stream));
    PL_HIP(hipMemsetAsync(c->Ap.p, 0, n6 * sizeof(double), c->stream));
    rc = pcg_tail_coarse(c, c->scal.p + set, c->scal.p, max_iter);
    if (rc) return rc;
  }
  double h_scal[pl::kSlots];
  PL_HIP(hipMemcpyAsync(h_scal, c->scal.p + pl::S_BB * pl::kSlots, sizeof(h_scal), hipMemcpyDeviceToHost,
                        c->stream));
  PL_HIP(hipStreamSynchronize(c->stream));
  double bb = 0.0;
  for (int k = 0; k < pl::kSlots; ++k) bb += h_scal[k];
  st->b_norm = std::sqrt(bb);
  st->iterations = 0;
  st->converged = 0;
  st->rel_residual = 0.0;
  if (!(bb > 0.0)) {   // zero right-hand side -> zero solution
    st->converged = 1;
    return std::isnan(bb) ? fail(PL_ERR_NAN, "NaN in the right-hand side") : PL_OK;
  }
  const double thresh = rtol * rtol * bb;
  // The host looks at the residual history every `chunk` iterations.  With the default (check_every = 0) the chunk
  // adapts: 32 while far from the threshold, then what the observed decay rate predicts is still needed - a fixed
  // chunk overshoots by 16 iterations on average, 8 % of a 200-iteration solve.  (Every rank of a multi-GPU run sees
  // the same all-reduced history, hence takes the same decisions.)
  const bool adaptive = c->opt.check_every <= 0;
  const int chunk = adaptive ? 32 : c->opt.check_every;
  const bool ref = ref_cg(c) && !c->dist.active && !c->coarse.ready;
  const int hcap = c->hist_cap;
  // A design loop solves a slowly changing system over and over: the iteration count of the previous converged solve on
  // this handle (identical on every rank) is where the first look at the history is worth taking - three iterations
  // before it - instead of every 32 iterations on the way there (each look drains the stream: 30-50 us).
  const int first = (adaptive && !ref && c->last_iterations > 40) ? std::min(c->last_iterations - 3, max_iter) : chunk;
  const int hbuf = std::max(chunk, first);
  std::vector<double> h_hist(hbuf), h_pp(ref ? hbuf : 0), h_xx(ref ? hbuf : 0), h_al(ref ? hbuf : 0);
  st->info = 1.0;
  int k = 0, next = first;
  double rr_prev = bb;
  int k_prev = 0;
  while (k < max_iter) {
    const int todo = std::min(next, max_iter - k);
    for (int j = 0; j < todo; ++j) {
      rc = pcg_iteration(c, k + j);
      if (rc) return rc;
    }
    PL_HIP(hipMemcpyAsync(h_hist.data(), c->hist.p + k, todo * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (ref) {
      PL_HIP(hipMemcpyAsync(h_pp.data(), c->hist.p + hcap + k, todo * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      PL_HIP(hipMemcpyAsync(h_xx.data(), c->hist.p + 2 * (size_t)hcap + k, todo * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      PL_HIP(hipMemcpyAsync(h_al.data(), c->hist.p + 3 * (size_t)hcap + k, todo * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    PL_HIP(hipStreamSynchronize(c->stream));
    for (int j = 0; j < todo; ++j) {
      const double rr = h_hist[j];
      if (std::isnan(rr) || std::isinf(rr)) return fail(PL_ERR_NAN, "NaN/Inf in the PCG residual");
      if (!st->converged) st->rel_residual = std::sqrt(rr / bb);
      if (rr <= thresh && !st->converged) {
        st->converged = 1;
        st->iterations = k + j + 1;
        st->info = 0.0;
        st->stop_reason = 0.0;
      }
      if (ref && !st->converged) {
        // conjugate_gradient_solver.py:102-109, in its order: the direction-norm stop, then the "tiny step" flag
        if (c->opt.mintol > 0.0 && std::sqrt(h_pp[j]) < c->opt.mintol * (std::sqrt(h_xx[j]) + 1e-12)) {
          st->converged = 1;
          st->iterations = k + j + 1;
          st->info = 0.0;
          st->stop_reason = 1.0;
        } else if (h_al[j] < 1e-6) {
          st->info = 2.0;
        }
      }
    }
    k += todo;
    if (st->converged) break;
    if (adaptive) {
      const double rr_end = h_hist[todo - 1];
      next = chunk;
      if (rr_end < rr_prev && rr_end > thresh) {
        const double per_it = std::log(rr_end / rr_prev) / (double)(k - k_prev);      // < 0
        const double need = std::log(thresh / rr_end) / per_it;
        if (need < 2.0 * chunk) next = std::max(2, std::min(chunk, (int)std::ceil(0.75 * need)));
      }
      rr_prev = rr_end;
      k_prev = k;
    }
  }
  if (!st->converged) st->iterations = k;
  c->last_iterations = st->converged ? st->iterations : 0;
  if (c->cond_use) {   // eliminated nodes: x_c = K_cc^-1 (b_c - (K [x_v ; 0])_c)
    rc = launch_spmv(c, c->x.p, c->tmp2.p, false, nullptr, nullptr, pl::kEndsCondensed);
    if (rc) return rc;
    hipLaunchKernelGGL(pl::k_condense_backsubst<double>, dim3(grid_for(c->n_cond * 6)), dim3(pl::kBlock), 0, c->stream,
                       c->n_cond, c->cnodes.p, cinv(c), ccls(c), (const double *)c->r.p, (const double *)c->tmp2.p, c->x.p);
    PL_HIP(hipGetLastError());
  }
  return PL_OK;
}

// ----------------------------------------------------------------------------------------------------------
// Single-reduction PCG (opts.cg_form = 1; pl_cg1.h): u -> z, w -> Ap, s -> tmp2.
// ----------------------------------------------------------------------------------------------------------
inline bool cg1_applies(const pl_context *c) {
  return c->opt.cg_form == 1 && c->coarse.ready && !c->coarseL.ready && !c->cond_use && c->opt.precision == 0 &&
         c->opkind == 0 && c->tile.ready && choose_kernel(c) == 3;
}

int pcg_solve_cg1(pl_context *c, const double *f_dev, const double *Kubar_dev, double rtol, int max_iter,
                  pl_stats_t *st) {
  const int64_t n6 = c->N * 6;
  pl::Coarse &cs = c->coarse;
  const int ncp = cs.ncp, bs = pl::cg1_block_size(ncp);
  int rc = ensure_hist(c, max_iter + 2);
  if (rc) return rc;
  const size_t need = 2 * (size_t)bs + 2 * pl::kSlots + 4 + (size_t)ncp;
  if (!c->cg1.p || c->cg1.n < need) PL_HIP(c->cg1.alloc(need));
  double *blk[2] = {c->cg1.p, c->cg1.p + bs};
  double *gc[2] = {c->cg1.p + 2 * bs, c->cg1.p + 2 * bs + pl::kSlots};
  double *stt[2] = {c->cg1.p + 2 * bs + 2 * pl::kSlots, c->cg1.p + 2 * bs + 2 * pl::kSlots + 2};
  double *sc = c->cg1.p + 2 * bs + 2 * pl::kSlots + 4;
  double *u = c->z.p, *w = c->Ap.p, *s = c->tmp2.p;
  const double *wt = c->dist.active ? (const double *)c->dist.weight.p : (const double *)nullptr;
  const uint8_t *shared = c->dist.active ? (const uint8_t *)c->sharedbits.p : (const uint8_t *)nullptr;
  const double *Bt = cs.tile_level ? (const double *)cs.Bt_inv : (const double *)nullptr;
  const double *yt = cs.tile_level ? (const double *)cs.yt : (const double *)nullptr;
  const dim3 gt((unsigned)cs.n_tiles), blkdim(cs.vblock);

  PL_HIP(hipMemsetAsync(c->scal.p, 0, 2 * pl::S_COUNT * pl::kSlots * sizeof(double), c->stream));
  PL_HIP(hipMemsetAsync(c->cg1.p, 0, need * sizeof(double), c->stream));
  // r0 = P (f - K ubar), x = 0, ||b||^2 (the z / p the init kernel also writes are overwritten below)
  if (c->dist.active)
    pl::launch_pcg_init_weighted(n6, f_dev, Kubar_dev, c->fixed.p, c->dinv.p, c->dist.weight.p, c->x.p, c->r.p,
                                 c->z.p, c->p.p, c->scal.p, c->stream);
  else
    hipLaunchKernelGGL(pl::k_pcg_init, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6, f_dev, Kubar_dev,
                       c->fixed.p, c->dinv.p, c->x.p, c->r.p, c->z.p, c->p.p, c->scal.p);
  PL_HIP(hipGetLastError());
  if (c->dist.active && pl::dist_sum_scalars(c->dist, c->scal.p + pl::S_BB * pl::kSlots, pl::kSlots, c->stream))
    return fail(PL_ERR_HIP, "RCCL all-reduce of the initial PCG scalars failed");
  PL_HIP(hipMemsetAsync(c->p.p, 0, n6 * sizeof(double), c->stream));
  PL_HIP(hipMemsetAsync(s, 0, n6 * sizeof(double), c->stream));
  PL_HIP(hipMemsetAsync(cs.rc, 0, (size_t)ncp * sizeof(double), c->stream));

  // u = M^-1 r, w = K u and the reduction block of iteration k (k = -1: the pass that prepares iteration 0)
  auto second_half = [&](int k) -> int {
    const int cur = k & 1, nxt = (k + 1) & 1;
    pl::dense_apply(cs.W, cs.Wt, ncp, ncp, cs.rc, cs.tv, cs.yc, gc[nxt], (const double *)nullptr, c->stream);
    hipLaunchKernelGGL(pl::k_cg1_precond, gt, blkdim, 0, c->stream, c->tile.tile_start.p, (const double *)c->r.p,
                       cs.dinv32, c->xyz.p, cs.agg_of_tile.p, cs.cen.p, cs.yc, yt, c->fixedbits.p, shared, u,
                       k >= 0 ? blk[cur] : (double *)nullptr, bs, k >= 0 ? gc[cur] : (double *)nullptr);
    int r2 = launch_spmv(c, u, w, true, blk[nxt] + ncp, nullptr, pl::kEndsAll, false);
    if (r2) return r2;
    hipLaunchKernelGGL(pl::k_cg1_restrict, gt, blkdim, 0, c->stream, c->tile.tile_start.p, cs.agg_of_tile.p, cs.cen.p,
                       c->xyz.p, (const double *)w, wt, blk[nxt]);
    if (c->dist.active && pl::dist_sum_scalars(c->dist, blk[nxt], bs, c->stream))
      return fail(PL_ERR_HIP, "RCCL all-reduce of the single-reduction PCG failed");
    PL_HIP(hipGetLastError());
    return PL_OK;
  };
  // Z^T r0 (summed over ranks once), tile level and partial sums of r0 into block 0, then u0, w0
  hipLaunchKernelGGL(pl::k_cg1_restrict, gt, blkdim, 0, c->stream, c->tile.tile_start.p, cs.agg_of_tile.p, cs.cen.p,
                     c->xyz.p, (const double *)c->r.p, wt, cs.rc);
  if (c->dist.active && pl::dist_sum_scalars(c->dist, cs.rc, ncp, c->stream))
    return fail(PL_ERR_HIP, "RCCL all-reduce of the initial coarse residual failed");
  hipLaunchKernelGGL(pl::k_cg1_update<true>, gt, blkdim, 0, c->stream, c->tile.tile_start.p, cs.agg_of_tile.p, cs.cen.p,
                     c->xyz.p, (const double *)u, (const double *)w, cs.dinv32, wt, c->p.p, s, c->x.p, c->r.p,
                     (const double *)blk[1], (const double *)gc[1], (const double *)stt[1], stt[0], blk[0], Bt, cs.yt,
                     shared, cs.rc, sc, ncp, c->hist.p, -1);
  rc = second_half(-1);
  if (rc) return rc;

  double h_scal[pl::kSlots];
  PL_HIP(hipMemcpyAsync(h_scal, c->scal.p + pl::S_BB * pl::kSlots, sizeof(h_scal), hipMemcpyDeviceToHost, c->stream));
  PL_HIP(hipStreamSynchronize(c->stream));
  double bb = 0.0;
  for (int k = 0; k < pl::kSlots; ++k) bb += h_scal[k];
  st->b_norm = std::sqrt(bb);
  st->iterations = 0;
  st->converged = 0;
  st->rel_residual = 0.0;
  st->info = 1.0;
  if (!(bb > 0.0)) {
    st->converged = 1;
    return std::isnan(bb) ? fail(PL_ERR_NAN, "NaN in the right-hand side") : PL_OK;
  }
  const double thresh = rtol * rtol * bb;
  const bool adaptive = c->opt.check_every <= 0;
  const int chunk = adaptive ? 32 : c->opt.check_every;
  std::vector<double> h_hist(chunk);
  int k = 0, next = chunk, k_prev = 0;
  double rr_prev = bb;
  // hist[k] = ||r_k||^2, the residual BEFORE update k (it is reduced together with that iteration's other sums)
  while (k < max_iter + 1) {
    const int todo = std::min(next, max_iter + 1 - k);
    for (int j = 0; j < todo; ++j) {
      const int it = k + j, cur = it & 1, nxt = (it + 1) & 1;
      hipLaunchKernelGGL(pl::k_cg1_update<false>, gt, blkdim, 0, c->stream, c->tile.tile_start.p, cs.agg_of_tile.p,
                         cs.cen.p, c->xyz.p, (const double *)u, (const double *)w, cs.dinv32, wt, c->p.p, s, c->x.p,
                         c->r.p, (const double *)blk[cur], (const double *)gc[cur], (const double *)stt[cur], stt[nxt],
                         blk[nxt], Bt, cs.yt, shared, cs.rc, sc, ncp, c->hist.p, it);
      rc = second_half(it);
      if (rc) return rc;
    }
    PL_HIP(hipMemcpyAsync(h_hist.data(), c->hist.p + k, todo * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PL_HIP(hipStreamSynchronize(c->stream));
    for (int j = 0; j < todo; ++j) {
      const double rr = h_hist[j];
      if (std::isnan(rr) || std::isinf(rr)) return fail(PL_ERR_NAN, "NaN/Inf in the PCG residual");
      if (!st->converged) st->rel_residual = std::sqrt(rr / bb);
      if (rr <= thresh && !st->converged) {
        st->converged = 1;
        st->iterations = k + j;      // updates applied when this residual was reached (x has had a few more since)
        st->info = 0.0;
        st->stop_reason = 0.0;
      }
    }
    k += todo;
    if (st->converged) break;
    if (adaptive) {
      const double rr_end = h_hist[todo - 1];
      next = chunk;
      if (rr_end < rr_prev && rr_end > thresh) {
        const double per_it = std::log(rr_end / rr_prev) / (double)(k - k_prev);
        const double need_it = std::log(thresh / rr_end) / per_it;
        if (need_it < 2.0 * chunk) next = std::max(2, std::min(chunk, (int)std::ceil(0.75 * need_it) + 1));
      }
      rr_prev = rr_end;
      k_prev = k;
    }
  }
  if (!st->converged) st->iterations = std::min(k, max_iter);
  return PL_OK;
}

// ----------------------------------------------------------------------------------------------------------
// fp32 solver modes (opts.precision; multi-level PCG on the tile kernel only):
//   1  inner PCG on fp32-stored x, r, p, Ap; the fp64 solution accumulates the inner corrections and every restart
//      begins from the TRUE fp64 residual P(f - K(ubar + x)) (classical iterative refinement);
//   2  only the search direction p and K*p are stored in fp32, x and the residual recurrence stay fp64: no restart
//      is needed to reach fp64 accuracy, the true residual is verified once the recurrence says "converged" (the
//      fp32 rounding of K*p lets the two drift apart by ~6e-8 of the accumulated steps).
// In both modes every product and sum is evaluated in fp64 (pl_tile.h, pl_coarse.h): fp32 only halves the bytes.
// ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(pl::kBlock) void k_mp_true_residual(int64_t n6, const double *__restrict__ f,
                                                                const double *__restrict__ Kubar,
                                                                const double *__restrict__ Kx /* may be null */,
                                                                const uint8_t *__restrict__ fixed,
                                                                const double *__restrict__ w /* may be null */,
                                                                double *__restrict__ r, double *__restrict__ rr_slots) {
  __shared__ double red[pl::kBlock / pl::kWave];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x; i < n6; i += (int64_t)gridDim.x * pl::kBlock) {
    const double v = fixed[i] ? 0.0 : f[i] - Kubar[i] - (Kx ? Kx[i] : 0.0);
    r[i] = v;
    acc += (w ? w[i] : 1.0) * v * v;
  }
  const double t = pl::block_sum(acc, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(rr_slots + (blockIdx.x & (pl::kSlots - 1)), t);
}
// start of an fp32 inner solve: r32 = r, x32 = 0
__global__ __launch_bounds__(pl::kBlock) void k_mp_restart(int64_t n6, const double *__restrict__ r,
                                                          float *__restrict__ r32, float *__restrict__ x32) {
  for (int64_t i = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x; i < n6; i += (int64_t)gridDim.x * pl::kBlock) {
    r32[i] = (float)r[i];
    x32[i] = 0.f;
  }
}
__global__ __launch_bounds__(pl::kBlock) void k_mp_accumulate(int64_t n6, const float *__restrict__ x32,
                                                             double *__restrict__ x) {
  for (int64_t i = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x; i < n6; i += (int64_t)gridDim.x * pl::kBlock)
    x[i] += (double)x32[i];
}

int read_slots(pl_context *c, const double *dev, double *sum) {
  double h[pl::kSlots];
  PL_HIP(hipMemcpyAsync(h, dev, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  PL_HIP(hipStreamSynchronize(c->stream));
  *sum = 0.0;
  for (int k = 0; k < pl::kSlots; ++k) *sum += h[k];
  return PL_OK;
}

template <typename RT>
int pcg_solve_mp_t(pl_context *c, const double *f_dev, const double *Kubar_dev, double rtol, int max_iter,
                   pl_stats_t *st) {
  constexpr bool kAll32 = sizeof(RT) == 4;
  const int64_t n6 = c->N * 6;
  const int set = pl::S_COUNT * pl::kSlots;
  int rc = ensure_hist(c, max_iter + 2);
  if (rc) return rc;
  float *p32 = reinterpret_cast<float *>(c->p.p), *Ap32 = reinterpret_cast<float *>(c->Ap.p);
  // mode 1: the inner iterate / residual live in the (otherwise unused) z buffer
  RT *xi = kAll32 ? reinterpret_cast<RT *>(c->z.p) : reinterpret_cast<RT *>(c->x.p);
  RT *ri = kAll32 ? reinterpret_cast<RT *>(c->z.p) + n6 : reinterpret_cast<RT *>(c->r.p);
  const double *w = c->dist.active ? (const double *)c->dist.weight.p : (const double *)nullptr;
  double *aux = c->scal.p + pl::S_AUX * pl::kSlots;   // slots outside the two per-parity sets' live entries
  PL_HIP(hipMemsetAsync(c->scal.p, 0, 2 * set * sizeof(double), c->stream));
  PL_HIP(hipMemsetAsync(c->x.p, 0, n6 * sizeof(double), c->stream));
  hipLaunchKernelGGL(k_mp_true_residual, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6, f_dev, Kubar_dev,
                     (const double *)nullptr, c->fixed.p, w, c->r.p, aux);
  PL_HIP(hipGetLastError());
  if (c->dist.active && pl::dist_sum_scalars(c->dist, aux, pl::kSlots, c->stream))
    return fail(PL_ERR_HIP, "RCCL all-reduce of ||b||^2 failed");
  double bb = 0.0;
  rc = read_slots(c, aux, &bb);
  if (rc) return rc;
  st->b_norm = std::sqrt(bb);
  st->iterations = 0;
  st->converged = 0;
  st->rel_residual = 0.0;
  if (!(bb > 0.0)) {
    st->converged = 1;
    return std::isnan(bb) ? fail(PL_ERR_NAN, "NaN in the right-hand side") : PL_OK;
  }
  const double thresh = rtol * rtol * bb;
  // an fp32 residual recurrence is trustworthy over ~4 decades: restart from the true residual after that
  const double inner_drop = kAll32 ? 1e-8 : 0.0;      // on ||r||^2
  double rr_true = bb;
  int k = 0;                      // iterations over all inner solves
  std::vector<double> h_hist(32);
  for (int outer = 0; outer < 40 && k < max_iter; ++outer) {
    // ---- (re)start: p = M^-1 r through the tail of an iteration "-1" (alpha = 0) on scalar set 1
    PL_HIP(hipMemsetAsync(c->scal.p, 0, 2 * set * sizeof(double), c->stream));
    if (kAll32)
      hipLaunchKernelGGL(k_mp_restart, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6, c->r.p,
                         reinterpret_cast<float *>(ri), reinterpret_cast<float *>(xi));
    PL_HIP(hipMemsetAsync(p32, 0, n6 * sizeof(float), c->stream));
    if (kAll32 && c->cond_use) {
      // node elimination inside the inner solve (as in pcg_solve): t_c = K_cc^-1 b_c in the rows of p32, r_v -= (K t)_v
      float *r32 = reinterpret_cast<float *>(ri);
      hipLaunchKernelGGL(pl::k_condense_solve<float>, dim3(grid_for(c->n_cond * 6)), dim3(pl::kBlock), 0, c->stream,
                         c->n_cond, c->cnodes.p, cinv(c), ccls(c), (const float *)r32, p32, 1.0);
      rc = launch_spmv_f32(c, p32, Ap32, true, nullptr, c->maskC.p, pl::kEndsOthers);
      if (rc) return rc;
      hipLaunchKernelGGL(pl::k_condense_subtract<float>, dim3(grid_for(n6)), dim3(pl::kBlock), 0, c->stream, c->N,
                         c->cflag.p, (const float *)Ap32, r32);
      PL_HIP(hipMemsetAsync(p32, 0, n6 * sizeof(float), c->stream));
    }
    PL_HIP(hipMemsetAsync(Ap32, 0, n6 * sizeof(float), c->stream));
    rc = pcg_tail_coarse_t<float, RT>(c, c->scal.p + set, c->scal.p, max_iter + 1, p32, (const float *)Ap32, xi, ri);
    if (rc) return rc;
    const double stop = std::max(thresh, inner_drop * rr_true);
    bool inner_done = false;
    int j = 0, next = 32;
    double rr_prev = rr_true;
    int j_prev = 0;
    while (!inner_done && k < max_iter) {
      const int todo = std::min(next, max_iter - k);
      for (int q = 0; q < todo; ++q) {
        double *cur = c->scal.p + ((j + q) & 1) * set, *nxt = c->scal.p + ((j + q + 1) & 1) * set;
        if (kAll32 && c->cond_use) {
          rc = launch_spmv_f32(c, p32, p32, false, nullptr, nullptr, pl::kEndsCondensedSolve);
          if (rc) return rc;
          rc = launch_spmv_f32(c, p32, Ap32, true, cur + pl::S_PAP * pl::kSlots, c->maskC.p, pl::kEndsOthers);
        } else {
          rc = launch_spmv_f32(c, p32, Ap32, true, cur + pl::S_PAP * pl::kSlots);
        }
        if (rc) return rc;
        rc = pcg_tail_coarse_t<float, RT>(c, cur, nxt, k + q, p32, (const float *)Ap32, xi, ri);
        if (rc) return rc;
      }
      PL_HIP(hipMemcpyAsync(h_hist.data(), c->hist.p + k, todo * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      PL_HIP(hipStreamSynchronize(c->stream));
      int used = todo;
      for (int q = 0; q < todo; ++q) {
        const double rr = h_hist[q];
        if (std::isnan(rr) || std::isinf(rr)) return fail(PL_ERR_NAN, "NaN/Inf in the PCG residual");
        if (rr <= stop) { inner_done = true; used = q + 1; break; }
      }
      // (the device has run the whole chunk: x holds the iterate after `todo` iterations, which is what is kept)
      const double rr_end = h_hist[todo - 1];
      j += todo;
      k += todo;
      (void)used;
      if (!inner_done) {
        next = 32;
        if (rr_end < rr_prev && rr_end > stop) {
          const double per_it = std::log(rr_end / rr_prev) / (double)(j - j_prev);
          const double need = std::log(stop / rr_end) / per_it;
          if (need < 64.0) next = std::max(2, std::min(32, (int)std::ceil(0.75 * need)));
        }
        rr_prev = rr_end;
        j_prev = j;
      }
    }
    // ---- true residual of the accumulated solution
    if (kAll32 && c->cond_use) {   // the eliminated nodes of this inner solve: x_c = K_cc^-1 (b_c - (K [x_v ; 0])_c)
      float *x32 = reinterpret_cast<float *>(xi), *r32 = reinterpret_cast<float *>(ri);
      rc = launch_spmv_f32(c, x32, Ap32, false, nullptr, nullptr, pl::kEndsCondensed);
      if (rc) return rc;
      hipLaunchKernelGGL(pl::k_condense_backsubst<float>, dim3(grid_for(c->n_cond * 6)), dim3(pl::kBlock), 0, c->stream,
                         c->n_cond, c->cnodes.p, cinv(c), ccls(c), (const float *)r32, (const float *)Ap32, x32);
    }
    if (kAll32)
      hipLaunchKernelGGL(k_mp_accumulate, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6,
                         reinterpret_cast<const float *>(xi), c->x.p);
    rc = launch_spmv(c, c->x.p, c->tmp2.p, true, nullptr);
    if (rc) return rc;
    PL_HIP(hipMemsetAsync(aux, 0, pl::kSlots * sizeof(double), c->stream));
    hipLaunchKernelGGL(k_mp_true_residual, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6, f_dev, Kubar_dev,
                       (const double *)c->tmp2.p, c->fixed.p, w, c->r.p, aux);
    PL_HIP(hipGetLastError());
    if (c->dist.active && pl::dist_sum_scalars(c->dist, aux, pl::kSlots, c->stream))
      return fail(PL_ERR_HIP, "RCCL all-reduce of the true residual failed");
    rc = read_slots(c, aux, &rr_true);
    if (rc) return rc;
    if (std::isnan(rr_true) || std::isinf(rr_true)) return fail(PL_ERR_NAN, "NaN/Inf in the true residual");
    st->rel_residual = std::sqrt(rr_true / bb);
    st->restarts = (double)(outer + 1);     // restarts (inner solves) taken
    if (rr_true <= thresh * 1.0000001) {
      st->converged = 1;
      break;
    }
  }
  st->iterations = k;
  return PL_OK;
}

// ----------------------------------------------------------------------------------------------------------
// incidence (sliced ELL) + BSR pattern, on the host
// ----------------------------------------------------------------------------------------------------------
int build_incidence(pl_context *c, const std::vector<int32_t> &conn) {
  const int64_t N = c->N, B = c->B;
  std::vector<int32_t> deg(N, 0);
  pl::parallel_for(2 * B, [&](int64_t k0, int64_t k1, unsigned) {
    for (int64_t k = k0; k < k1; ++k) __atomic_fetch_add(&deg[conn[k]], 1, __ATOMIC_RELAXED);
  }, 1 << 16);
  std::vector<int64_t> ptr(N + 1, 0);
  for (int64_t i = 0; i < N; ++i) ptr[i + 1] = ptr[i] + deg[i];
  struct E {
    int32_t other, code;
  };
  std::vector<E> adj((size_t)ptr[N]);
  std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
  pl::parallel_for(B, [&](int64_t b0, int64_t b1, unsigned) {       // any order inside a row: the rows are sorted next
    for (int64_t b = b0; b < b1; ++b) {
      const int32_t a = conn[2 * b], d = conn[2 * b + 1];
      adj[__atomic_fetch_add(&fill[d], (int64_t)1, __ATOMIC_RELAXED)] = {a, (int32_t)b};   // d is the strut's tip (point2)
      adj[__atomic_fetch_add(&fill[a], (int64_t)1, __ATOMIC_RELAXED)] =                    // a is point1 -> reversed record
          {d, (int32_t)((uint32_t)b | 0x80000000u)};
    }
  }, 1 << 16);
  pl::parallel_for(N, [&](int64_t i0, int64_t i1, unsigned) {
    for (int64_t i = i0; i < i1; ++i)
      std::sort(adj.begin() + ptr[i], adj.begin() + ptr[i + 1],
                [](const E &l, const E &r) { return l.other < r.other || (l.other == r.other && l.code < r.code); });
  });

  // sliced ELL: kSliceNodes (16) nodes per slice, width padded to a multiple of kLPN (4) so that one slice is a
  // whole number of 64-entry wave trips
  const int SN = pl::kWave / c->lpn;
  const int64_t S = (N + SN - 1) / SN;
  std::vector<int64_t> sp(S + 1, 0);
  for (int64_t s = 0; s < S; ++s) {
    int w = 0;
    for (int64_t i = s * SN; i < std::min<int64_t>(N, s * SN + SN); ++i) w = std::max(w, deg[i]);
    w = (w + c->lpn - 1) / c->lpn * c->lpn;
    sp[s + 1] = sp[s] + (int64_t)w * SN;
  }
  std::vector<int2> ent((size_t)sp[S], int2{-1, 0});
  // BSR pattern: per row the diagonal block + one block per entry, columns ascending.  Parallel struts between the
  // same pair of nodes (possible in hybrid cells) get separate blocks with equal column index.
  c->h_rowptr.assign(N + 1, 0);
  for (int64_t i = 0; i < N; ++i) c->h_rowptr[i + 1] = c->h_rowptr[i] + deg[i] + 1;
  c->nblk = c->h_rowptr[N];
  c->h_col.assign((size_t)c->nblk, 0);
  std::vector<int32_t> ent_slot((size_t)sp[S], 0), diag_slot(N, 0);
  pl::parallel_for(N, [&](int64_t i0, int64_t i1, unsigned) {
  for (int64_t i = i0; i < i1; ++i) {
    const int64_t s = i / SN, lane = i % SN;
    int slot = 0;
    bool diag_done = false;
    for (int j = 0; j < deg[i]; ++j) {
      const E &e = adj[ptr[i] + j];
      if (!diag_done && e.other > i) {
        diag_slot[i] = slot;
        c->h_col[c->h_rowptr[i] + slot++] = (int32_t)i;
        diag_done = true;
      }
      const int64_t pos = sp[s] + (int64_t)j * SN + lane;
      ent[pos] = int2{e.other, e.code};
      ent_slot[pos] = slot;
      c->h_col[c->h_rowptr[i] + slot++] = e.other;
    }
    if (!diag_done) {
      diag_slot[i] = slot;
      c->h_col[c->h_rowptr[i] + slot++] = (int32_t)i;
    }
  }
  });
  c->n_slices = S;
  c->n_ent = sp[S];
  PL_HIP(c->slice_ptr.alloc(S + 1));
  PL_HIP(c->ent.alloc(std::max<size_t>(1, ent.size())));
  PL_HIP(c->ent_slot.alloc(std::max<size_t>(1, ent_slot.size())));
  PL_HIP(c->diag_slot.alloc(N));
  PL_HIP(hipMemcpy(c->slice_ptr.p, sp.data(), (S + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  if (!ent.empty()) {
    PL_HIP(hipMemcpy(c->ent.p, ent.data(), ent.size() * sizeof(int2), hipMemcpyHostToDevice));
    PL_HIP(hipMemcpy(c->ent_slot.p, ent_slot.data(), ent_slot.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  PL_HIP(hipMemcpy(c->diag_slot.p, diag_slot.data(), N * sizeof(int32_t), hipMemcpyHostToDevice));
  return PL_OK;
}

bool valid(pl_handle h) { return h != nullptr; }

// PL_TIMING=1 in the environment: wall clock of the host-side stages of pl_create on stderr
struct StageTimer {
  const char *what;
  bool on;
  std::chrono::steady_clock::time_point t0;
  explicit StageTimer(const char *w) : what(w), on(std::getenv("PL_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
  void mark(const char *stage) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[%s] %-28s %8.1f ms\n", what, stage, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

}  // namespace

// ==========================================================================================================
// C ABI
// ==========================================================================================================
namespace {
template <typename WT>
int debug_spd_solve_t(int device, int32_t n, const double *A, const double *b, double *x, double *quad) {
  if (n <= 0 || !A || !b || !x) return fail(PL_ERR_ARG, "pl_debug_spd_solve: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PL_ERR_NODEVICE, "no HIP device visible");
  PL_HIP(hipSetDevice(device));
  const int np = (n + pl::kNB - 1) / pl::kNB * pl::kNB;
  std::vector<double> Ap((size_t)np * np, 0.0), bp(np, 0.0);
  for (int i = 0; i < np; ++i) {
    if (i < n) {
      std::memcpy(&Ap[(size_t)i * np], A + (size_t)i * n, n * sizeof(double));
      bp[i] = b[i];
    } else {
      Ap[(size_t)i * np + i] = 1.0;
    }
  }
  DevBuf<double> dA, dL, dD, db, dt, dy, dq;
  DevBuf<WT> dW, dWt;
  DevBuf<int> dinfo;
  PL_HIP(dA.alloc(Ap.size()));
  PL_HIP(dW.alloc(Ap.size()));
  PL_HIP(dL.alloc(Ap.size()));
  PL_HIP(dWt.alloc(Ap.size()));
  PL_HIP(hipMemset(dWt.p, 0, Ap.size() * sizeof(WT)));
  PL_HIP(dD.alloc((size_t)np * pl::kNB));
  PL_HIP(db.alloc(np));
  PL_HIP(dt.alloc(np));
  PL_HIP(dy.alloc(np));
  PL_HIP(dq.alloc(pl::kSlots));
  PL_HIP(dinfo.alloc(2));
  PL_HIP(hipMemcpy(dA.p, Ap.data(), Ap.size() * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(db.p, bp.data(), np * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemset(dW.p, 0, Ap.size() * sizeof(WT)));
  PL_HIP(hipMemset(dq.p, 0, pl::kSlots * sizeof(double)));
  PL_HIP(hipMemset(dinfo.p, 0, 2 * sizeof(int)));
  pl::dense_factor_inverse(dA.p, dL.p, dW.p, dWt.p, dD.p, np, np, dinfo.p, 0, nullptr);
  pl::dense_apply(dW.p, dWt.p, np, np, db.p, dt.p, dy.p, dq.p, nullptr, nullptr);
  PL_HIP(hipGetLastError());
  PL_HIP(hipDeviceSynchronize());
  int info[2];
  PL_HIP(hipMemcpy(info, dinfo.p, sizeof(info), hipMemcpyDeviceToHost));
  if (info[0] != 0) return fail(PL_ERR_ARG, "pl_debug_spd_solve: matrix is not positive definite");
  std::vector<double> y(np), q(pl::kSlots);
  PL_HIP(hipMemcpy(y.data(), dy.p, np * sizeof(double), hipMemcpyDeviceToHost));
  PL_HIP(hipMemcpy(q.data(), dq.p, pl::kSlots * sizeof(double), hipMemcpyDeviceToHost));
  std::memcpy(x, y.data(), n * sizeof(double));
  if (quad) {
    *quad = 0.0;
    for (double v : q) *quad += v;
  }
  return PL_OK;
}

// ---- multi-GPU ------------------------------------------------------------------------------------------
}  // namespace

extern "C" {

const char *pl_last_error(void) { return g_err.c_str(); }
const char *pl_version(void) { return "pylattice_hip 0.1 (gfx950)"; }

void pl_default_opts(pl_opts_t *o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->young = 1013.0;   // VeroClear
  o->poisson = 0.3;
  o->kappa = 0.9;
  o->pen_coef = 1.5;
  o->device = 0;
  o->spmv_kernel = 0;
  o->precond = 1;
  o->reorder = 1;
  o->check_every = 0;   // adaptive
}

namespace {
inline bool multi_rank_handle(const pl_opts_t *o) { return o->grid_nodes > 0; }
// modes per aggregate of the dense level: 12 (rigid + strains) needs the 12-mode tile level, i.e. a single-GPU handle in
// the ordinary CG form with precond = 3
inline int coarse_modes_of(const pl_opts_t *o, int64_t N = -1) {
  const bool tile12 = o->precond == 3 && !(o->tile_modes == 6 || o->cg_form == 1);
  if (!tile12 || o->coarse_modes == 6) return 6;
  if (multi_rank_handle(o)) N = o->grid_nodes;       // every rank must decide alike: the node count of the whole lattice
  // automatic: from a quarter of a million nodes on (below, the longer set-up of the richer level costs what its
  // iterations save: 32^3 Octet 178 M beams/s with 6 modes, 169 M with 12)
  return (o->coarse_modes == 12 || N < 0 || N >= 250000) ? 12 : 6;
}
// dofs the dense level may have (see the comment at its set-up in pl_create)
inline int coarse_budget(const pl_opts_t *o, int64_t N) {
  const bool multi_rank = o->grid_nodes > 0;
  if (o->coarse_max_dofs > 0) return o->coarse_max_dofs;
  // 12 modes per aggregate: 5^3 aggregates (1 500 dofs) beat 7^3 x 6 (2 058) below a million nodes - 120 against 126
  // iterations at 50^3 Octet, 24 chain links instead of 33 (measured: 1 536 -> 203, 2 600 -> 200, 800 -> 193 M beams/s)
  // (the level's cost does not grow with the lattice, its benefit does: 100^3 BCC 3 072 -> 59.7, 6 144 -> 61.4 M beams/s;
  // 200 x 200 x 50 BCC + Octet 3 072 -> 104.9, 6 144 -> 136.4 M beams/s)
  // several GPUs: replicated on every rank and all-reduced per assembly, decided on the node count of the WHOLE lattice so
  // that every rank decides alike.  Emulated 4 / 8 ranks (50 x 50N x 50 Octet): 161 / 179 iterations with 6 modes and 3 072
  // dofs, 139 / 174 with 12 modes and 3 072, 120 / 139 with 12 modes and 6 144
  if (multi_rank) return (coarse_modes_of(o, N) == 12 && o->grid_nodes >= 2000000) ? 6144 : 3072;
  if (coarse_modes_of(o, N) == 12) return N >= 2000000 ? 6144 : (N >= 1000000 ? 3072 : (N >= 250000 ? 1536 : 768));
  return N >= 1000000 ? 3072 : 2100;
}
}  // namespace

int pl_create(const pl_mesh_t *m, const pl_opts_t *o, pl_handle *out) {
  if (!m || !o || !out) return fail(PL_ERR_ARG, "pl_create: null argument");
  StageTimer stage("pl_create");
  if (m->n_nodes <= 0 || m->n_beams <= 0) return fail(PL_ERR_ARG, "pl_create: empty mesh");
  if (m->n_nodes >= (1LL << 31) - 64 || m->n_beams >= (1LL << 31) - 64)
    return fail(PL_ERR_ARG, "pl_create: more than 2^31 nodes/struts per handle");
  if (!m->node_xyz || !m->beam_conn || !m->beam_radius || !m->seg_len || !m->seg_nsub)
    return fail(PL_ERR_ARG, "pl_create: null mesh array");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(PL_ERR_NODEVICE, "pl_create: no HIP device visible (libpylattice_hip has no CPU fallback)");
  if (o->device < 0 || o->device >= ndev) return fail(PL_ERR_ARG, "pl_create: bad device ordinal");
  const int64_t N = m->n_nodes, B = m->n_beams;
  {
    std::atomic<int64_t> bad_ends{-1}, bad_radius{-1}, bad_len{-1}, bad_seg{-1};
    pl::parallel_for(B, [&](int64_t b0, int64_t b1, unsigned) {
      for (int64_t b = b0; b < b1; ++b) {
        const int32_t a = m->beam_conn[2 * b], d = m->beam_conn[2 * b + 1];
        if (a < 0 || d < 0 || a >= N || d >= N || a == d) bad_ends = b;
        if (!(m->beam_radius[b] > 0.0)) bad_radius = b;
        const double L = m->seg_len[3 * b] + m->seg_len[3 * b + 1] + m->seg_len[3 * b + 2];
        if (!(L > 0.0)) bad_len = b;
        for (int k = 0; k < 3; ++k)
          if (m->seg_len[3 * b + k] < 0.0 || (m->seg_len[3 * b + k] > 0.0 && m->seg_nsub[3 * b + k] < 1)) bad_seg = b;
      }
    });
    if (bad_ends >= 0)
      return fail(PL_ERR_ARG, "pl_create: strut " + std::to_string(bad_ends.load()) + " has invalid end nodes");
    if (bad_radius >= 0) return fail(PL_ERR_ARG, "pl_create: non-positive radius");
    if (bad_len >= 0) return fail(PL_ERR_ARG, "pl_create: strut with zero length");
    if (bad_seg >= 0) return fail(PL_ERR_ARG, "pl_create: bad segment data on strut " + std::to_string(bad_seg.load()));
  }
  stage.mark("validate");
  PL_HIP(hipSetDevice(o->device));
  pl_context *c = new pl_context();
  c->opt = *o;
  c->N = N;
  c->B = B;
  c->mat = {o->young, o->young / (2.0 * (1.0 + o->poisson)), o->kappa, o->pen_coef};
  c->lpn = o->lanes_per_node > 0 ? o->lanes_per_node : pl::kDefaultLPN;
  if (c->lpn != 1 && c->lpn != 2 && c->lpn != 4 && c->lpn != 8 && c->lpn != 16) {
    delete c;
    return fail(PL_ERR_ARG, "pl_create: lanes_per_node must be 0 (default), 1, 2, 4, 8 or 16");
  }
  auto bail = [&](int rc) {
    delete c;
    return rc;
  };
#define PL_TRY(expr)              \
  do {                            \
    int _rc = (expr);             \
    if (_rc) return bail(_rc);    \
  } while (0)
#define PL_HIPC(expr)                                                                  \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess)                                                              \
      return bail(fail(PL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); \
  } while (0)
  {   // the main stream carries the latency-bound chains (dense factorisation, PCG): it goes ahead of the bulk fills
    int prio_low = 0, prio_high = 0;
    PL_HIPC(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    PL_HIPC(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_high));
    PL_HIPC(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_low));
  }
  PL_HIPC(hipEventCreate(&c->ev0));
  PL_HIPC(hipEventCreate(&c->ev1));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_chol, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_t0, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_t1, hipEventDisableTiming));
  PL_HIPC(hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking));

  // node ordering on the device
  const double global_grid[7] = {o->grid_lo[0], o->grid_lo[1], o->grid_lo[2], o->grid_hi[0], o->grid_hi[1],
                                 o->grid_hi[2], (double)o->grid_nodes};
  c->perm.resize(N);
  std::iota(c->perm.begin(), c->perm.end(), 0);
  std::vector<int32_t> tile_start, tile_of;
  std::vector<int64_t> tile_brick;
  pl::BrickGrid grid;
  if (o->reorder == 1) {
    pl::spatial_order(m->node_xyz, N, c->perm, tile_start,
                      (double)(o->tile_nodes > 0 ? std::min(o->tile_nodes, pl::kTileMaxNodes) : 256), tile_brick,
                      grid, o->grid_nodes > 0 ? global_grid : nullptr,
                      (o->precond >= 2 && o->precond <= 4) ? coarse_budget(o, N) * 6 / coarse_modes_of(o, N) : 0);
    c->reordered = true;
  } else {
    pl::chunk_tiles(N, tile_start);
  }
  stage.mark("spatial order");
  c->iperm.resize(N);
  pl::parallel_for(N, [&](int64_t i0, int64_t i1, unsigned) {
    for (int64_t i = i0; i < i1; ++i) c->iperm[c->perm[i]] = (int32_t)i;
  }, 1 << 16);
  if (o->condense >= 0 && o->precond >= 2 && o->precision != 2 && o->reorder == 1 && o->grid_nodes == 0) {
    // Candidates for exact elimination inside the PCG (opts.condense): a greedy maximal independent set of the node
    // graph (no two share a strut; at least three struts each).  Inside every tile they are numbered LAST, so that the
    // vector kernels, which skip them, skip one contiguous run of rows per tile.
    std::vector<int64_t> aptr((size_t)N + 1, 0);
    pl::parallel_for(2 * B, [&](int64_t k0, int64_t k1, unsigned) {
      for (int64_t k = k0; k < k1; ++k)
        __atomic_fetch_add(&aptr[c->iperm[m->beam_conn[k]] + 1], (int64_t)1, __ATOMIC_RELAXED);
    }, 1 << 16);
    for (int64_t i = 0; i < N; ++i) aptr[i + 1] += aptr[i];
    std::vector<int32_t> adj((size_t)2 * B);
    {   // neighbour lists in any order: the greedy passes below only ask whether a neighbour is taken
      std::vector<int64_t> fill(aptr.begin(), aptr.end() - 1);
      pl::parallel_for(B, [&](int64_t b0, int64_t b1, unsigned) {
        for (int64_t b = b0; b < b1; ++b) {
          const int32_t u = c->iperm[m->beam_conn[2 * b]], v = c->iperm[m->beam_conn[2 * b + 1]];
          adj[__atomic_fetch_add(&fill[u], (int64_t)1, __ATOMIC_RELAXED)] = v;
          adj[__atomic_fetch_add(&fill[v], (int64_t)1, __ATOMIC_RELAXED)] = u;
        }
      }, 1 << 16);
    }
    // two greedy passes: nodes strictly inside the bounding box first, then the ones on it.  Boundary conditions sit
    // on the faces (a node with a Dirichlet dof cannot be eliminated) and face nodes have fewer struts; on BCC this
    // picks the cell centres rather than the corners (measured: 702 -> 468 iterations at 50^3 against 575 the other way)
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = 0; i < N; ++i)
      for (int k = 0; k < 3; ++k) {
        lo[k] = std::min(lo[k], m->node_xyz[3 * i + k]);
        hi[k] = std::max(hi[k], m->node_xyz[3 * i + k]);
      }
    std::vector<uint8_t> state((size_t)N, 0);
    for (int pass = 0; pass < 2; ++pass)
      for (int64_t i = 0; i < N; ++i) {
        if (state[i] || aptr[i + 1] - aptr[i] < 3) continue;
        const double *q3 = m->node_xyz + 3 * (size_t)c->perm[i];
        const bool on_box = q3[0] <= lo[0] || q3[0] >= hi[0] || q3[1] <= lo[1] || q3[1] >= hi[1] || q3[2] <= lo[2] ||
                            q3[2] >= hi[2];
        if (on_box != (pass == 1)) continue;
        state[i] = 1;
        for (int64_t q = aptr[i]; q < aptr[i + 1]; ++q)
          if (!state[adj[q]]) state[adj[q]] = 2;
      }
    int64_t n_cand = 0;
    for (int64_t i = 0; i < N; ++i) n_cand += state[i] == 1;
    // automatic mode: only when close to half of the unknowns can go (bipartite node graphs such as BCC: measured
    // 1.1-1.2 x faster solves; Octet, a quarter of the nodes: 1.2 x slower - every iteration pays a second K*p)
    const bool use = o->condense > 0 || (double)n_cand >= 0.45 * (double)N;
    const int64_t T = use ? (int64_t)tile_start.size() - 1 : 0;
    std::vector<int32_t> np(c->perm.size());
    if (use) c->h_cand.assign((size_t)N, 0);
    for (int64_t t = 0; t < T; ++t) {
      int64_t w = tile_start[t];
      for (int pass = 0; pass < 2; ++pass)
        for (int64_t i = tile_start[t]; i < tile_start[t + 1]; ++i)
          if ((state[i] == 1) == (pass == 1)) {
            c->h_cand[w] = (uint8_t)pass;
            np[w++] = c->perm[i];
          }
    }
    if (use) {
      c->perm.swap(np);
      for (int64_t i = 0; i < N; ++i) c->iperm[c->perm[i]] = (int32_t)i;
    }
  }

  stage.mark("candidates");
  std::vector<double> xyz((size_t)N * 3);
  pl::parallel_for(N, [&](int64_t i0, int64_t i1, unsigned) {
    for (int64_t i = i0; i < i1; ++i)
      std::memcpy(&xyz[3 * i], m->node_xyz + 3 * (size_t)c->perm[i], 3 * sizeof(double));
  });
  std::vector<int32_t> conn((size_t)B * 2);
  {
    std::vector<int32_t> conn0((size_t)B * 2);
    pl::parallel_for(2 * B, [&](int64_t k0, int64_t k1, unsigned) {
      for (int64_t k = k0; k < k1; ++k) conn0[k] = c->iperm[m->beam_conn[k]];
    });
    pl::tile_strut_order(conn0, N, B, tile_start, tile_of, c->bperm);
    pl::parallel_for(B, [&](int64_t b0, int64_t b1, unsigned) {
      for (int64_t b = b0; b < b1; ++b) {
        conn[2 * b] = conn0[2 * (size_t)c->bperm[b]];
        conn[2 * b + 1] = conn0[2 * (size_t)c->bperm[b] + 1];
      }
    });
  }
  stage.mark("strut order");
  std::vector<double> radius(B), seg_len((size_t)B * 3);
  std::vector<int32_t> seg_nsub((size_t)B * 3);
  pl::parallel_for(B, [&](int64_t b0, int64_t b1, unsigned) {
    for (int64_t b = b0; b < b1; ++b) {
      const size_t ob = (size_t)c->bperm[b];
      radius[b] = m->beam_radius[ob];
      for (int k = 0; k < 3; ++k) {
        seg_len[3 * b + k] = m->seg_len[3 * ob + k];
        seg_nsub[3 * b + k] = m->seg_nsub[3 * ob + k];
      }
    }
  });

  PL_HIPC(c->xyz.alloc(N * 3));
  PL_HIPC(c->conn.alloc(B * 2));
  PL_HIPC(c->radius.alloc(B));
  PL_HIPC(c->seg_len.alloc(B * 3));
  PL_HIPC(c->seg_nsub.alloc(B * 3));
  PL_HIPC(c->rec.alloc(B));
  // opts.compact_records (default on): the tile K*p streams 40-byte records when no palette applies
  if (o->compact_records >= 0 && (o->spmv_kernel == 0 || o->spmv_kernel == 3)) PL_HIPC(c->rec5.alloc((size_t)B * 5));
  PL_HIPC(hipMemcpy(c->xyz.p, xyz.data(), xyz.size() * sizeof(double), hipMemcpyHostToDevice));
  PL_HIPC(hipMemcpy(c->conn.p, conn.data(), conn.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  PL_HIPC(hipMemcpy(c->radius.p, radius.data(), B * sizeof(double), hipMemcpyHostToDevice));
  PL_HIPC(hipMemcpy(c->seg_len.p, seg_len.data(), 3 * B * sizeof(double), hipMemcpyHostToDevice));
  PL_HIPC(hipMemcpy(c->seg_nsub.p, seg_nsub.data(), 3 * B * sizeof(int32_t), hipMemcpyHostToDevice));
  stage.mark("permute + upload");
  PL_TRY(build_incidence(c, conn));
  stage.mark("incidence + BSR pattern");
  {
    int rc = pl::build_tile_plan(c->tile, conn, N, B, tile_start, tile_of);
    if (rc) return bail(fail(PL_ERR_HIP, "pl_create: building the LDS tile plan failed (" + std::to_string(rc) + ")"));
  }
  stage.mark("tile plan");
  if (o->precond >= 2 && o->precond <= 4) {
    if (!c->reordered) return bail(fail(PL_ERR_ARG, "pl_create: precond = 2/3/4 (multi-level) needs reorder = 1"));
    c->coarse.tile_level = (o->precond >= 3);
    // 12-mode tile level (rigid + uniform strains): single-GPU handles in the ordinary CG form; opts.tile_modes = 6 keeps
    // the rigid-body blocks
    c->coarse.tile_modes = (o->tile_modes == 6 || o->precond == 4 || o->cg_form == 1) ? 6 : 12;
    // default size of the dense level: its factorisation is a ~45 us-per-64-dofs latency chain in every assembly, its
    // benefit grows with the cost of an iteration - up to 1 M nodes on one GPU the optimum is ~2 000 dofs (measured on
    // 50^3 Octet: 7^3 aggregates 18.9 ms per step, 8^3 20.7 ms), beyond that and in multi-rank runs (collectives in
    // every iteration) the full 3 072
    const bool multi_rank = o->grid_nodes > 0;
    const int max_dofs = coarse_budget(o, N);
    int rc = pl::coarse_setup(c->coarse, tile_start, tile_brick, grid, xyz.data(), N, max_dofs, conn, false, multi_rank,
                              coarse_modes_of(o, N));
    if (rc == 4)
      return bail(fail(PL_ERR_ARG, "pl_create: a strut spans more than neighbouring aggregates; the band-packed "
                                   "all-reduce of the coarse operator of a multi-GPU handle cannot hold it (use "
                                   "precond = 1 or a smaller coarse_max_dofs)"));
    if (rc) return bail(fail(PL_ERR_HIP, "pl_create: coarse-space setup failed (" + std::to_string(rc) + ")"));
    PL_HIPC(c->sharedbits.alloc(N));
    PL_HIPC(c->maskL.alloc(N));
    PL_HIPC(hipMemset(c->sharedbits.p, 0, N));
    if (o->precond == 4) {
      const int maxL = o->local_max_dofs > 0 ? o->local_max_dofs : 3072;
      rc = pl::coarse_setup(c->coarseL, tile_start, tile_brick, grid, xyz.data(), N, maxL, conn, true);
      if (rc) return bail(fail(PL_ERR_HIP, "pl_create: local coarse-space setup failed (" + std::to_string(rc) + ")"));
      c->coarseL.tile_level = false;
    }
  }

  stage.mark("coarse setup");
  const size_t n6 = (size_t)N * 6;
  PL_HIPC(c->fixed.alloc(n6));
  PL_HIPC(c->fixedbits.alloc(N));
  PL_HIPC(c->ubar.alloc(n6));
  PL_HIPC(c->f.alloc(n6));
  for (DevBuf<double> *v : {&c->diag, &c->dinv, &c->x, &c->r, &c->z, &c->p, &c->Ap, &c->tmp, &c->tmp2})
    PL_HIPC(v->alloc(n6));
  PL_HIPC(c->scal.alloc(2 * pl::S_COUNT * pl::kSlots));   // two sets, selected by iteration parity
  PL_HIPC(hipMemset(c->fixed.p, 0, n6));
  PL_HIPC(hipMemset(c->fixedbits.p, 0, N));
  PL_HIPC(hipMemset(c->ubar.p, 0, n6 * sizeof(double)));
  PL_HIPC(hipMemset(c->f.p, 0, n6 * sizeof(double)));
  PL_HIPC(hipDeviceSynchronize());
#undef PL_TRY
#undef PL_HIPC
  stage.mark("vector buffers");
  *out = c;
  return PL_OK;
}

int pl_create_ddm(int64_t n_nodes, int64_t n_cells, int32_t nb, const int32_t *cell_nodes, int32_t n_S,
                  const double *S, const int32_t *cell_S, const pl_opts_t *o, pl_handle *out) {
  if (!cell_nodes || !S || !cell_S || !o || !out) return fail(PL_ERR_ARG, "pl_create_ddm: null argument");
  if (n_nodes <= 0 || n_cells <= 0 || nb <= 0 || n_S <= 0) return fail(PL_ERR_ARG, "pl_create_ddm: empty problem");
  if (6 * nb > pl::kDdmMaxM) return fail(PL_ERR_ARG, "pl_create_ddm: more than 27 boundary nodes per cell");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(PL_ERR_NODEVICE, "pl_create_ddm: no HIP device visible (libpylattice_hip has no CPU fallback)");
  if (o->device < 0 || o->device >= ndev) return fail(PL_ERR_ARG, "pl_create_ddm: bad device ordinal");
  for (int64_t k = 0; k < n_cells * nb; ++k)
    if (cell_nodes[k] < 0 || cell_nodes[k] >= n_nodes) return fail(PL_ERR_ARG, "pl_create_ddm: node id out of range");
  for (int64_t c = 0; c < n_cells; ++c)
    if (cell_S[c] < 0 || cell_S[c] >= n_S) return fail(PL_ERR_ARG, "pl_create_ddm: matrix id out of range");
  PL_HIP(hipSetDevice(o->device));
  pl_context *c = new pl_context();
  c->opt = *o;
  // 0: the reference's plain CG; 1: Jacobi on the assembled Schur diagonal; 2: the reference's own preconditioner, the
  // factorised assembled Schur matrix (dense Cholesky on the device, hence the size limit)
  c->opt.precond = (o->precond == 1 || o->precond == 2) ? o->precond : 0;
  if (c->opt.precond == 2 && 6 * n_nodes > PL_DDM_DENSE_MAX) {
    delete c;
    return fail(PL_ERR_ARG, "pl_create_ddm: precond = 2 factorises a dense (6 n_nodes)^2 matrix; limit is " +
                                std::to_string(PL_DDM_DENSE_MAX) + " dofs (use precond = 1)");
  }
  c->opkind = 1;
  c->N = n_nodes;
  c->B = 0;
  c->ddm_cells = n_cells;
  c->ddm_nb = nb;
  {   // dofs coupled by a cell are at most 6 * (max - min node id) + 5 apart: the band of the assembled matrix
    int64_t span = 0;
    for (int64_t cc = 0; cc < n_cells; ++cc) {
      const int32_t *nd = cell_nodes + cc * nb;
      span = std::max<int64_t>(span, *std::max_element(nd, nd + nb) - *std::min_element(nd, nd + nb));
    }
    c->dd_bw = (int)((6 * span + 5) / pl::kNB + 1);
  }
  auto bail = [&](int rc) {
    delete c;
    return rc;
  };
#define PL_HIPC(expr)                                                                  \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess)                                                              \
      return bail(fail(PL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); \
  } while (0)
  {   // the main stream carries the latency-bound chains (dense factorisation, PCG): it goes ahead of the bulk fills
    int prio_low = 0, prio_high = 0;
    PL_HIPC(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    PL_HIPC(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_high));
    PL_HIPC(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_low));
  }
  PL_HIPC(hipEventCreate(&c->ev0));
  PL_HIPC(hipEventCreate(&c->ev1));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_chol, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_t0, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_t1, hipEventDisableTiming));
  PL_HIPC(hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking));
  c->perm.resize(n_nodes);
  std::iota(c->perm.begin(), c->perm.end(), 0);
  c->iperm = c->perm;
  const int m = 6 * nb;
  std::vector<double> St((size_t)n_S * m * m);
  for (int s = 0; s < n_S; ++s)
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) St[((size_t)s * m + j) * m + i] = S[((size_t)s * m + i) * m + j];
  PL_HIPC(c->ddm_cell_nodes.alloc((size_t)n_cells * nb));
  PL_HIPC(c->ddm_cell_S.alloc(n_cells));
  PL_HIPC(c->ddm_St.alloc(St.size()));
  PL_HIPC(hipMemcpy(c->ddm_cell_nodes.p, cell_nodes, (size_t)n_cells * nb * sizeof(int32_t), hipMemcpyHostToDevice));
  PL_HIPC(hipMemcpy(c->ddm_cell_S.p, cell_S, n_cells * sizeof(int32_t), hipMemcpyHostToDevice));
  PL_HIPC(hipMemcpy(c->ddm_St.p, St.data(), St.size() * sizeof(double), hipMemcpyHostToDevice));
  {
    if (n_cells * nb >= (1LL << 31)) return bail(fail(PL_ERR_ARG, "pl_create_ddm: more than 2^31 cell-node entries"));
    std::vector<int64_t> nptr((size_t)n_nodes + 1, 0);
    for (int64_t k = 0; k < n_cells * nb; ++k) nptr[cell_nodes[k] + 1]++;
    for (int64_t i = 0; i < n_nodes; ++i) nptr[i + 1] += nptr[i];
    std::vector<int32_t> nent((size_t)n_cells * nb);
    std::vector<int64_t> fill(nptr.begin(), nptr.end() - 1);
    for (int64_t k = 0; k < n_cells * nb; ++k) nent[fill[cell_nodes[k]]++] = (int32_t)k;   // cell-major: fixed sum order
    PL_HIPC(c->ddm_node_ptr.alloc(nptr.size()));
    PL_HIPC(c->ddm_node_ent.alloc(nent.size()));
    PL_HIPC(c->ddm_stage.alloc((size_t)n_cells * m));
    PL_HIPC(hipMemcpy(c->ddm_node_ptr.p, nptr.data(), nptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    PL_HIPC(hipMemcpy(c->ddm_node_ent.p, nent.data(), nent.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    std::vector<int32_t> order((size_t)n_cells);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t l, int32_t r) { return cell_S[l] < cell_S[r]; });
    PL_HIPC(c->ddm_order.alloc(order.size()));
    PL_HIPC(hipMemcpy(c->ddm_order.p, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  const size_t n6 = (size_t)n_nodes * 6;
  PL_HIPC(c->fixed.alloc(n6));
  PL_HIPC(c->fixedbits.alloc(n_nodes));
  PL_HIPC(c->ubar.alloc(n6));
  PL_HIPC(c->f.alloc(n6));
  // (vectors padded to the block size of the dense solver and zeroed: its GEMVs read / write whole blocks)
  for (DevBuf<double> *v : {&c->diag, &c->dinv, &c->x, &c->r, &c->z, &c->p, &c->Ap, &c->tmp, &c->tmp2}) {
    PL_HIPC(v->alloc(n6 + pl::kNB));
    PL_HIPC(hipMemset(v->p, 0, (n6 + pl::kNB) * sizeof(double)));
  }
  PL_HIPC(c->scal.alloc(2 * pl::S_COUNT * pl::kSlots));
  PL_HIPC(hipMemset(c->fixed.p, 0, n6));
  PL_HIPC(hipMemset(c->fixedbits.p, 0, n_nodes));
  PL_HIPC(hipMemset(c->ubar.p, 0, n6 * sizeof(double)));
  PL_HIPC(hipMemset(c->f.p, 0, n6 * sizeof(double)));
  PL_HIPC(hipDeviceSynchronize());
#undef PL_HIPC
  *out = c;
  return PL_OK;
}

int pl_ddm_set_preconditioner(pl_handle h, int32_t n_S, const double *S, const int32_t *cell_S) {
  if (!valid(h)) return fail(PL_ERR_ARG, "pl_ddm_set_preconditioner: null handle");
  if (h->opkind != 1) return fail(PL_ERR_STATE, "pl_ddm_set_preconditioner: not a DDM handle");
  PL_HIP(hipSetDevice(h->opt.device));
  h->assembled = false;
  if (!S) {   // back to the operator's own matrices
    h->ddm_have_P = false;
    return PL_OK;
  }
  if (n_S <= 0 || !cell_S) return fail(PL_ERR_ARG, "pl_ddm_set_preconditioner: bad argument");
  for (int64_t c = 0; c < h->ddm_cells; ++c)
    if (cell_S[c] < 0 || cell_S[c] >= n_S) return fail(PL_ERR_ARG, "pl_ddm_set_preconditioner: matrix id out of range");
  const int m = 6 * h->ddm_nb;
  std::vector<double> Pt((size_t)n_S * m * m);
  for (int s = 0; s < n_S; ++s)
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) Pt[((size_t)s * m + j) * m + i] = S[((size_t)s * m + i) * m + j];
  PL_HIP(h->ddm_Pt.alloc(Pt.size()));
  PL_HIP(h->ddm_cell_P.alloc(h->ddm_cells));
  PL_HIP(hipMemcpy(h->ddm_Pt.p, Pt.data(), Pt.size() * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(h->ddm_cell_P.p, cell_S, h->ddm_cells * sizeof(int32_t), hipMemcpyHostToDevice));
  h->ddm_have_P = true;
  return PL_OK;
}

// G = sum_c B^T Shat B on the free dofs (unit diagonal elsewhere), Cholesky + explicit inverse factor on the device.
static int ddm_factor_preconditioner(pl_context *h) {
  const int64_t n6 = h->N * 6;
  const int np = (int)((n6 + pl::kNB - 1) / pl::kNB * pl::kNB);
  if (h->dd_n != np) {
    const size_t nn = (size_t)np * np;
    PL_HIP(h->dd_A.alloc(nn));
    PL_HIP(h->dd_Lf.alloc(nn));
    PL_HIP(h->dd_W.alloc(nn));
    PL_HIP(h->dd_Wt.alloc(nn));
    PL_HIP(h->dd_Dinv.alloc((size_t)np * pl::kNB));
    PL_HIP(h->dd_tv.alloc(np));
    PL_HIP(h->dd_info.alloc(2));
    PL_HIP(hipMemsetAsync(h->dd_W.p, 0, nn * sizeof(double), h->stream));
    PL_HIP(hipMemsetAsync(h->dd_Wt.p, 0, nn * sizeof(double), h->stream));
    h->dd_n = np;
  }
  h->dd_ready = false;
  PL_HIP(hipMemsetAsync(h->dd_A.p, 0, (size_t)np * np * sizeof(double), h->stream));
  PL_HIP(hipMemsetAsync(h->dd_info.p, 0, 2 * sizeof(int), h->stream));
  const int m = 6 * h->ddm_nb;
  const int64_t ne = h->ddm_cells * m * m;
  const uint8_t *fx = h->have_bc ? h->fixed.p : (const uint8_t *)nullptr;
  hipLaunchKernelGGL(pl::k_ddm_dense_assemble, dim3(grid_for(ne)), dim3(pl::kBlock), 0, h->stream, h->ddm_cells,
                     h->ddm_nb, h->ddm_cell_nodes.p, h->ddm_have_P ? h->ddm_cell_P.p : h->ddm_cell_S.p,
                     h->ddm_have_P ? h->ddm_Pt.p : h->ddm_St.p, fx, np, h->dd_A.p);
  hipLaunchKernelGGL(pl::k_ddm_dense_unit, dim3((np + 255) / 256), dim3(256), 0, h->stream, n6, np, fx, h->dd_A.p);
  pl::dense_factor_inverse(h->dd_A.p, h->dd_Lf.p, h->dd_W.p, h->dd_Wt.p, h->dd_Dinv.p, np, np, h->dd_info.p,
                           h->dd_bw, h->stream);
  PL_HIP(hipGetLastError());
  int info[2] = {0, 0};
  PL_HIP(hipMemcpyAsync(info, h->dd_info.p, sizeof(info), hipMemcpyDeviceToHost, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  h->dd_ready = (info[0] == 0);   // not positive definite (e.g. an indefinite surrogate matrix): caller falls back
  return PL_OK;
}

void pl_destroy(pl_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->opt.device);
  (void)hipStreamSynchronize(h->stream);
  pl::dist_destroy(h->dist);
  if (h->pal_host_flags != h->pal_fallback_flags) (void)hipHostFree(h->pal_host_flags);
  if (h->cls_host_flag) (void)hipHostFree(h->cls_host_flag);
  delete h;
}

int pl_set_bc(pl_handle h, const uint8_t *fixed, const double *ubar, const double *f) {
  if (!valid(h) || !fixed) return fail(PL_ERR_ARG, "pl_set_bc: null argument");
  PL_HIP(hipSetDevice(h->opt.device));
  const int64_t N = h->N;
  const size_t n6 = (size_t)N * 6;
  std::vector<uint8_t> fx(n6), bits(N);
  std::vector<double> ub(n6, 0.0), ff(n6, 0.0);
  for (int64_t i = 0; i < N; ++i) {
    const size_t src = 6 * (size_t)h->perm[i];
    uint8_t b = 0;
    for (int k = 0; k < 6; ++k) {
      const uint8_t v = fixed[src + k] ? 1 : 0;
      fx[6 * i + k] = v;
      b |= (uint8_t)(v << k);
      if (ubar && v) ub[6 * i + k] = ubar[src + k];
      if (f) ff[6 * i + k] = f[src + k];
    }
    bits[i] = b;
  }
  PL_HIP(hipMemcpy(h->fixed.p, fx.data(), n6, hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(h->fixedbits.p, bits.data(), N, hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(h->ubar.p, ub.data(), n6 * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(h->f.p, ff.data(), n6 * sizeof(double), hipMemcpyHostToDevice));
  h->have_bc = true;
  h->coarse.n_fix = -1;
  h->coarseL.n_fix = -1;
  {
    int rcs = select_condensed(h, bits);
    if (rcs) return rcs;
  }
  if (h->assembled && h->opkind == 1) {
    if (h->opt.precond == 2) {
      // the factorised G was built for the old Dirichlet mask (and dinv must stay 0 next to it): build it again
      h->assembled = false;
      h->dd_ready = false;
    } else {
      pl::launch_invert_diag(h->N * 6, h->diag.p, h->fixed.p, h->dinv.p, h->stream);
      PL_HIP(hipStreamSynchronize(h->stream));
    }
  } else if (h->assembled) {   // the Jacobi inverse and the coarse operator depend on the mask
    int rc = launch_diag(h, h->stream);
    if (rc) return rc;
    rc = finish_diag_dist(h);
    if (rc) return rc;
    rc = launch_local_mask(h);
    if (rc) return rc;
    rc = launch_tile_blocks(h, h->stream);
    if (rc) return rc;
    rc = build_coarse(h);
    if (rc) return rc;
    rc = launch_dinv32(h);
    if (rc) return rc;
    rc = launch_condensed_blocks(h, h->stream);
    if (rc) return rc;
    PL_HIP(hipStreamSynchronize(h->stream));
    {
      const bool keep = h->pal_ready;        // (the record palette is untouched by a new mask)
      finish_condensed_classes(h);
      (void)keep;
    }
    if (h->bsr_with_bc) h->have_bsr = false;   // an explicit matrix built with the old mask is stale
  }
  return PL_OK;
}

int pl_update_radii(pl_handle h, const double *beam_radius) {
  if (!valid(h) || !beam_radius) return fail(PL_ERR_ARG, "pl_update_radii: null argument");
  if (h->opkind != 0) return fail(PL_ERR_STATE, "pl_update_radii: not available on a DDM handle");
  for (int64_t b = 0; b < h->B; ++b)
    if (!(beam_radius[b] > 0.0)) return fail(PL_ERR_ARG, "pl_update_radii: non-positive radius");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<double> radius(h->B);
  for (int64_t b = 0; b < h->B; ++b) radius[b] = beam_radius[h->bperm[b]];
  PL_HIP(hipMemcpy(h->radius.p, radius.data(), h->B * sizeof(double), hipMemcpyHostToDevice));
  h->assembled = false;
  h->have_bsr = false;
  return PL_OK;
}

int pl_update_segments(pl_handle h, const double *seg_len, const int32_t *seg_nsub) {
  if (!valid(h) || !seg_len || !seg_nsub) return fail(PL_ERR_ARG, "pl_update_segments: null argument");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<double> sl((size_t)h->B * 3);
  std::vector<int32_t> sn((size_t)h->B * 3);
  for (int64_t b = 0; b < h->B; ++b)
    for (int k = 0; k < 3; ++k) {
      sl[3 * b + k] = seg_len[3 * (size_t)h->bperm[b] + k];
      sn[3 * b + k] = seg_nsub[3 * (size_t)h->bperm[b] + k];
    }
  PL_HIP(hipMemcpy(h->seg_len.p, sl.data(), 3 * h->B * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(h->seg_nsub.p, sn.data(), 3 * h->B * sizeof(int32_t), hipMemcpyHostToDevice));
  h->assembled = false;
  h->have_bsr = false;
  return PL_OK;
}

int pl_assemble(pl_handle h) {
  if (!valid(h)) return fail(PL_ERR_ARG, "pl_assemble: null handle");
  PL_HIP(hipSetDevice(h->opt.device));
  if (h->opkind == 1) {   // DDM operator: nothing to build; plain CG as the reference, Jacobi, or the factorised matrix
    h->dd_ready = false;
    if (h->opt.precond == 2) {
      PL_HIP(hipEventRecord(h->ev0, h->stream));
      int rcp = ddm_factor_preconditioner(h);
      if (rcp) return rcp;
      if (h->dd_ready) {
        PL_HIP(hipMemsetAsync(h->dinv.p, 0, h->N * 6 * sizeof(double), h->stream));   // z comes from the dense solve
        PL_HIP(hipEventRecord(h->ev1, h->stream));
        PL_HIP(hipEventSynchronize(h->ev1));
        float ms = 0.f;
        PL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        h->ms_assembly = ms;
        h->assembled = true;
        return PL_OK;
      }
      // G is not positive definite (the reference falls back from LU to ILU when its factorisation fails,
      // lattice_sim.py:1406-1413): Jacobi on the assembled diagonal; pl_stats_t.precond_used tells the caller
    }
    if (h->opt.precond >= 1) {
      PL_HIP(hipMemsetAsync(h->diag.p, 0, h->N * 6 * sizeof(double), h->stream));
      const int64_t m = (int64_t)h->ddm_cells * h->ddm_nb * 6;
      hipLaunchKernelGGL(pl::k_ddm_diag, dim3(grid_for(m)), dim3(pl::kBlock), 0, h->stream, h->ddm_cells, h->ddm_nb,
                         h->ddm_cell_nodes.p, h->ddm_cell_S.p, h->ddm_St.p, h->diag.p);
    } else {
      pl::launch_fill(h->N * 6, 1.0, h->diag.p, h->stream);
    }
    pl::launch_invert_diag(h->N * 6, h->diag.p, h->have_bc ? h->fixed.p : nullptr, h->dinv.p, h->stream);
    PL_HIP(hipStreamSynchronize(h->stream));
    h->ms_assembly = 0.0;
    h->assembled = true;
    return PL_OK;
  }
  PL_HIP(hipEventRecord(h->ev0, h->stream));
  int rc = launch_records(h);
  if (rc) return rc;
  rc = launch_local_mask(h);
  if (rc) return rc;
  // fork: everything that only streams the records runs on the side stream while the main stream walks the
  // latency-bound chain of the coarse factorisation
  PL_HIP(hipEventRecord(h->ev_fork, h->stream));
  PL_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
  rc = launch_palette(h, h->side);
  if (rc) return rc;
  rc = launch_diag(h, h->dist.active ? h->stream : h->side);
  if (rc) return rc;
  rc = finish_diag_dist(h);
  if (rc) return rc;
  rc = launch_tile_blocks(h, h->side);
  if (rc) return rc;
  rc = launch_condensed_blocks(h, h->side);
  if (rc) return rc;
  const bool refresh_bsr = h->want_bsr && (!h->bsr_with_bc || h->have_bc);
  // The explicit K is the one bulk item of the assembly (2.5 GB of traffic): next to the Cholesky chain it slows every
  // link of that latency-bound chain (44 -> 57 us), so it starts when the chain's last link is queued and runs beside
  // the single-launch inverse factor instead.
  int rc_fill = PL_OK;
  bool fill_queued = false;
  auto queue_fill = [&]() {
    if (!refresh_bsr || fill_queued) return;
    fill_queued = true;
    if (hipEventRecord(h->ev_chol, h->stream) != hipSuccess || hipStreamWaitEvent(h->side, h->ev_chol, 0) != hipSuccess) {
      rc_fill = fail(PL_ERR_HIP, "pl_assemble: could not order the BSR fill behind the factorisation");
      return;
    }
    rc_fill = launch_bsr_fill(h, h->bsr_with_bc, h->side);
    if (hipEventRecord(h->ev_join, h->side) != hipSuccess) rc_fill = fail(PL_ERR_HIP, "pl_assemble: event record failed");
  };
  PL_HIP(hipEventRecord(h->ev_join, h->side));
  rc = build_coarse(h, h->coarse.enabled && h->have_bc ? std::function<void()>(queue_fill) : std::function<void()>());
  if (rc) return rc;
  queue_fill();                                  // (no dense level: queue it now)
  if (rc_fill) return rc_fill;
  PL_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  rc = launch_dinv32(h);
  if (rc) return rc;
  PL_HIP(hipEventRecord(h->ev1, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  finish_palette(h);
  finish_condensed_classes(h);
  // the factorisation has consumed A_c: zero it now, behind the caller's back, instead of at the head of the next
  // assembly's critical chain (34 MB; it would sit in front of the coarse assembly there)
  for (pl::Coarse *cs : {&h->coarse, &h->coarseL})
    if (cs->ready && cs->Ac) {
      PL_HIP(hipMemsetAsync(cs->Ac, 0, (size_t)cs->ncp * cs->ncp * sizeof(double), h->stream));
      cs->ac_clean = true;
    }
  float ms = 0.f;
  PL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->ms_assembly = ms;
  h->assembled = true;
  h->have_bsr = refresh_bsr;
  return PL_OK;
}

int pl_assemble_bsr(pl_handle h, int with_bc, int64_t *n_block_rows, int64_t *n_blocks) {
  if (!valid(h)) return fail(PL_ERR_ARG, "pl_assemble_bsr: null handle");
  if (h->opkind != 0) return fail(PL_ERR_STATE, "pl_assemble_bsr: not available on a DDM handle");
  if (!h->assembled) return fail(PL_ERR_STATE, "pl_assemble_bsr: call pl_assemble first");
  if (with_bc && !h->have_bc) return fail(PL_ERR_STATE, "pl_assemble_bsr: with_bc needs pl_set_bc");
  PL_HIP(hipSetDevice(h->opt.device));
  if (!h->bsr_vals.p) {
    PL_HIP(h->bsr_rowptr.alloc(h->N + 1));
    PL_HIP(h->bsr_col.alloc(h->nblk));
    PL_HIP(h->bsr_vals.alloc((size_t)h->nblk * 36));
    PL_HIP(hipMemcpy(h->bsr_rowptr.p, h->h_rowptr.data(), (h->N + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    PL_HIP(hipMemcpy(h->bsr_col.p, h->h_col.data(), h->nblk * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  const bool fresh = h->have_bsr && h->want_bsr && h->bsr_with_bc == (with_bc ? 1 : 0);
  h->want_bsr = true;
  h->bsr_with_bc = with_bc ? 1 : 0;
  if (fresh) {   // pl_assemble already rebuilt it (overlapped with the coarse factorisation)
    if (n_block_rows) *n_block_rows = h->N;
    if (n_blocks) *n_blocks = h->nblk;
    return PL_OK;
  }
  PL_HIP(hipEventRecord(h->ev0, h->stream));
  int rc = launch_bsr_fill(h, with_bc, h->stream);
  if (rc) return rc;
  PL_HIP(hipEventRecord(h->ev1, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  PL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->ms_assembly += ms;
  h->have_bsr = true;
  if (n_block_rows) *n_block_rows = h->N;
  if (n_blocks) *n_blocks = h->nblk;
  return PL_OK;
}

int pl_get_bsr(pl_handle h, int64_t *rowptr, int32_t *colidx, double *vals) {
  if (!valid(h)) return fail(PL_ERR_ARG, "pl_get_bsr: null handle");
  if (!h->have_bsr) return fail(PL_ERR_STATE, "pl_get_bsr: call pl_assemble_bsr first");
  PL_HIP(hipSetDevice(h->opt.device));
  // returned in CALLER numbering: row i of the device matrix is caller node perm[i]
  std::vector<double> v((size_t)h->nblk * 36);
  PL_HIP(hipMemcpy(v.data(), h->bsr_vals.p, v.size() * sizeof(double), hipMemcpyDeviceToHost));
  if (!h->reordered) {
    if (rowptr) std::memcpy(rowptr, h->h_rowptr.data(), (h->N + 1) * sizeof(int64_t));
    if (colidx) std::memcpy(colidx, h->h_col.data(), h->nblk * sizeof(int32_t));
    if (vals) std::memcpy(vals, v.data(), v.size() * sizeof(double));
    return PL_OK;
  }
  // permuted: rebuild rows in caller order, columns re-sorted
  std::vector<int64_t> rp(h->N + 1, 0);
  for (int64_t ci = 0; ci < h->N; ++ci) {
    const int64_t di = h->iperm[ci];
    rp[ci + 1] = rp[ci] + (h->h_rowptr[di + 1] - h->h_rowptr[di]);
  }
  if (rowptr) std::memcpy(rowptr, rp.data(), (h->N + 1) * sizeof(int64_t));
  std::vector<std::pair<int32_t, int64_t>> row;
  for (int64_t ci = 0; ci < h->N; ++ci) {
    const int64_t di = h->iperm[ci];
    row.clear();
    for (int64_t p = h->h_rowptr[di]; p < h->h_rowptr[di + 1]; ++p) row.push_back({h->perm[h->h_col[p]], p});
    std::stable_sort(row.begin(), row.end(), [](auto &l, auto &r) { return l.first < r.first; });
    for (size_t k = 0; k < row.size(); ++k) {
      if (colidx) colidx[rp[ci] + k] = row[k].first;
      if (vals) std::memcpy(vals + 36 * (rp[ci] + k), &v[36 * row[k].second], 36 * sizeof(double));
    }
  }
  return PL_OK;
}

static int spmv_common(pl_handle h, const double *x, double *y, bool masked, const char *who) {
  if (!valid(h) || !x || !y) return fail(PL_ERR_ARG, std::string(who) + ": null argument");
  if (!h->assembled) return fail(PL_ERR_STATE, std::string(who) + ": call pl_assemble first");
  if (masked && !h->have_bc) return fail(PL_ERR_STATE, std::string(who) + ": call pl_set_bc first");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<double> stage;
  int rc = upload6(h, x, h->tmp.p, stage);
  if (rc) return rc;
  if (masked) {
    // enforce the contract "x is zero on fixed dofs" for arbitrary caller input
    hipLaunchKernelGGL(pl::k_mask_dot, dim3(grid_stream(h->N * 6)), dim3(pl::kBlock), 0, h->stream, h->N * 6,
                       h->fixed.p, h->tmp.p, h->tmp.p, (double *)nullptr);
  }
  rc = launch_spmv(h, h->tmp.p, h->tmp2.p, masked, nullptr);
  if (rc) return rc;
  return download6(h, h->tmp2.p, y);
}

int pl_spmv(pl_handle h, const double *x, double *y) { return spmv_common(h, x, y, false, "pl_spmv"); }
int pl_spmv_free(pl_handle h, const double *x, double *y) { return spmv_common(h, x, y, true, "pl_spmv_free"); }

int pl_spmv_bsr(pl_handle h, const double *x, double *y) {
  if (!valid(h) || !x || !y) return fail(PL_ERR_ARG, "pl_spmv_bsr: null argument");
  if (!h->have_bsr) return fail(PL_ERR_STATE, "pl_spmv_bsr: call pl_assemble_bsr first");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<double> stage;
  int rc = upload6(h, x, h->tmp.p, stage);
  if (rc) return rc;
  hipLaunchKernelGGL(pl::k_bsr_spmv, dim3(grid_for(h->N)), dim3(pl::kBlock), 0, h->stream, h->N, h->bsr_rowptr.p,
                     h->bsr_col.p, h->bsr_vals.p, h->tmp.p, h->tmp2.p);
  PL_HIP(hipGetLastError());
  return download6(h, h->tmp2.p, y);
}

int pl_solve(pl_handle h, double rtol, int32_t max_iter, double *u, pl_stats_t *stats) {
  if (!valid(h)) return fail(PL_ERR_ARG, "pl_solve: null handle");   // u == NULL: leave the solution on the device
  if (!h->assembled) return fail(PL_ERR_STATE, "pl_solve: call pl_assemble first");
  if (!h->have_bc) return fail(PL_ERR_STATE, "pl_solve: call pl_set_bc first");
  if (!(rtol > 0.0) || max_iter <= 0) return fail(PL_ERR_ARG, "pl_solve: rtol and max_iter must be positive");
  PL_HIP(hipSetDevice(h->opt.device));
  const int64_t n6 = h->N * 6;
  pl_stats_t st{};
  PL_HIP(hipEventRecord(h->ev0, h->stream));
  // lifting: tmp = K ubar (ubar is zero on free dofs)
  int rc = launch_spmv(h, h->ubar.p, h->tmp.p, false, nullptr);
  if (rc) return rc;
  // fp32 solver modes need the multi-level preconditioner on the tile kernel; anything else runs the fp64 PCG
  const bool mp = h->opt.precision != 0 && h->opkind == 0 && h->coarse.ready && choose_kernel(h) == 3 && h->tile.ready;
  h->cond_use = h->cond_ready && h->coarse.ready && h->opkind == 0 && choose_kernel(h) == 3 && h->tile.ready &&
                (!mp || h->opt.precision == 1) && h->opt.cg_form != 1;
  const bool cg1 = !mp && cg1_applies(h);
  st.cg_form_used = cg1 ? 1.0 : 0.0;
  if (cg1) rc = pcg_solve_cg1(h, h->f.p, h->tmp.p, rtol, max_iter, &st);
  else if (mp && h->opt.precision == 1) rc = pcg_solve_mp_t<float>(h, h->f.p, h->tmp.p, rtol, max_iter, &st);
  else if (mp) rc = pcg_solve_mp_t<double>(h, h->f.p, h->tmp.p, rtol, max_iter, &st);
  else rc = pcg_solve(h, h->f.p, h->tmp.p, rtol, max_iter, &st);
  if (rc) return rc;
  st.precision_used = mp ? (double)h->opt.precision : 0.0;
  st.condensed_nodes = h->cond_use ? (double)h->n_cond : 0.0;
  if (st.converged) st.info = 0.0;
  else if (st.info != 2.0) st.info = 1.0;     // precision mode the solve ran in
  hipLaunchKernelGGL(pl::k_compose_solution, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, h->stream, n6, h->fixed.p,
                     h->ubar.p, h->x.p, h->tmp2.p);
  PL_HIP(hipEventRecord(h->ev1, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  PL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  st.ms_solve = ms;
  st.ms_assembly = h->ms_assembly;
  st.precond_used = h->opkind == 1 ? (h->dd_ready ? 2 : h->opt.precond >= 1 ? 1 : 0)
                                   : (h->coarse.ready ? h->opt.precond : 1);
  if (u) {
    rc = download6(h, h->tmp2.p, u);
    if (rc) return rc;
  }
  h->last = st;
  if (stats) *stats = st;
  if (!st.converged) return fail(PL_ERR_NOCONV, "pl_solve: PCG did not reach rtol within max_iter");
  return PL_OK;
}

int pl_reactions(pl_handle h, const double *u, double *R) { return spmv_common(h, u, R, false, "pl_reactions"); }

int pl_sens(pl_handle h, const double *u, const double *lam, double *dCdr) {
  if (!valid(h) || !u || !dCdr) return fail(PL_ERR_ARG, "pl_sens: null argument");
  if (h->opkind != 0) return fail(PL_ERR_STATE, "pl_sens: not available on a DDM handle");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<double> stage;
  int rc = upload6(h, u, h->tmp.p, stage);
  if (rc) return rc;
  const double *lam_dev = h->tmp.p;
  if (lam) {
    rc = upload6(h, lam, h->tmp2.p, stage);
    if (rc) return rc;
    lam_dev = h->tmp2.p;
  }
  DevBuf<double> out;
  PL_HIP(out.alloc(h->B));
  hipLaunchKernelGGL(pl::k_sens, dim3(grid_for(h->B)), dim3(pl::kBlock), 0, h->stream, h->B, h->xyz.p, h->conn.p,
                     h->radius.p, h->seg_len.p, h->seg_nsub.p, h->mat, h->tmp.p, lam_dev, out.p);
  PL_HIP(hipGetLastError());
  std::vector<double> tmp(h->B);
  PL_HIP(hipMemcpyAsync(tmp.data(), out.p, h->B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  for (int64_t b = 0; b < h->B; ++b) dCdr[h->bperm[b]] = tmp[b];
  return PL_OK;
}

int pl_node_mod(pl_handle h, const double *u, double *out) {
  if (!valid(h) || !u || !out) return fail(PL_ERR_ARG, "pl_node_mod: null argument");
  if (h->opkind != 0) return fail(PL_ERR_STATE, "pl_node_mod: not available on a DDM handle");
  if (!h->assembled) return fail(PL_ERR_STATE, "pl_node_mod: call pl_assemble first");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<double> stage;
  int rc = upload6(h, u, h->tmp.p, stage);
  if (rc) return rc;
  DevBuf<double> dev;
  PL_HIP(dev.alloc((size_t)h->B * 12));
  hipLaunchKernelGGL(pl::k_node_mod, dim3(grid_for(h->B)), dim3(pl::kBlock), 0, h->stream, h->B, h->xyz.p, h->conn.p,
                     h->radius.p, h->seg_len.p, h->seg_nsub.p, h->mat, h->rec.p, h->tmp.p, dev.p);
  PL_HIP(hipGetLastError());
  std::vector<double> tmp((size_t)h->B * 12);
  PL_HIP(hipMemcpyAsync(tmp.data(), dev.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  for (int64_t b = 0; b < h->B; ++b) std::memcpy(out + 12 * (size_t)h->bperm[b], &tmp[12 * (size_t)b], 12 * sizeof(double));
  return PL_OK;
}

int pl_energy(pl_handle h, const double *u, double *energy) {
  if (!valid(h) || !u || !energy) return fail(PL_ERR_ARG, "pl_energy: null argument");
  if (h->opkind != 0) return fail(PL_ERR_STATE, "pl_energy: not available on a DDM handle");
  if (!h->assembled) return fail(PL_ERR_STATE, "pl_energy: call pl_assemble first");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<double> stage;
  int rc = upload6(h, u, h->tmp.p, stage);
  if (rc) return rc;
  double *aux = h->scal.p + pl::S_AUX * pl::kSlots;
  double h_aux[pl::kSlots];
  PL_HIP(hipMemsetAsync(aux, 0, sizeof(h_aux), h->stream));
  hipLaunchKernelGGL(pl::k_energy, dim3(grid_for(h->B)), dim3(pl::kBlock), 0, h->stream, h->B, h->conn.p, h->rec.p,
                     h->tmp.p, aux);
  PL_HIP(hipGetLastError());
  PL_HIP(hipMemcpyAsync(h_aux, aux, sizeof(h_aux), hipMemcpyDeviceToHost, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  *energy = 0.0;
  for (int k = 0; k < pl::kSlots; ++k) *energy += h_aux[k];
  return PL_OK;
}

int pl_schur(pl_handle h, const int32_t *boundary_nodes, int32_t nb, double rtol, int32_t max_iter, double *S) {
  if (!valid(h) || !boundary_nodes || !S || nb <= 0) return fail(PL_ERR_ARG, "pl_schur: bad argument");
  if (!h->assembled) return fail(PL_ERR_STATE, "pl_schur: call pl_assemble first");
  PL_HIP(hipSetDevice(h->opt.device));
  const int64_t N = h->N;
  const size_t n6 = (size_t)N * 6;
  for (int i = 0; i < nb; ++i)
    if (boundary_nodes[i] < 0 || boundary_nodes[i] >= N) return fail(PL_ERR_ARG, "pl_schur: node index out of range");
  // Column j of S = reaction on the boundary dofs when boundary dof j = 1, the other boundary dofs = 0 and the
  // interior is in equilibrium: exactly S = K_BB - K_BI K_II^-1 K_IB.
  std::vector<uint8_t> fixed(n6, 0);
  for (int i = 0; i < nb; ++i)
    for (int k = 0; k < 6; ++k) fixed[6 * (size_t)boundary_nodes[i] + k] = 1;
  std::vector<double> ubar(n6, 0.0), u(n6), R(n6);
  const int m = nb * 6;
  for (int j = 0; j < m; ++j) {
    const size_t dofj = 6 * (size_t)boundary_nodes[j / 6] + (j % 6);
    ubar[dofj] = 1.0;
    int rc = pl_set_bc(h, fixed.data(), ubar.data(), nullptr);
    if (rc) return rc;
    pl_stats_t st;
    rc = pl_solve(h, rtol, max_iter, u.data(), &st);
    if (rc) return rc;
    rc = pl_reactions(h, u.data(), R.data());
    if (rc) return rc;
    for (int i = 0; i < m; ++i) S[(size_t)i * m + j] = R[6 * (size_t)boundary_nodes[i / 6] + (i % 6)];
    ubar[dofj] = 0.0;
  }
  return PL_OK;
}

int pl_get_records(pl_handle h, double *rec) {
  if (!valid(h) || !rec) return fail(PL_ERR_ARG, "pl_get_records: null argument");
  if (!h->assembled) return fail(PL_ERR_STATE, "pl_get_records: call pl_assemble first");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<pl::Record> tmp(h->B);
  PL_HIP(hipMemcpy(tmp.data(), h->rec.p, h->B * sizeof(pl::Record), hipMemcpyDeviceToHost));
  for (int64_t b = 0; b < h->B; ++b) std::memcpy(rec + 8 * (size_t)h->bperm[b], &tmp[b], sizeof(pl::Record));
  return PL_OK;
}

int pl_algorithmic_bytes(pl_handle h, double *out3) {
  if (!valid(h) || !out3) return fail(PL_ERR_ARG, "pl_algorithmic_bytes: null argument");
  const double w = 8.0, B = (double)h->B, N = (double)h->N;
  out3[0] = B * (8.0 + 8.0 * w) + N * 6.0 * w * 2.0;        // bytes_spmv
  out3[1] = out3[0] + 10.0 * (6.0 * N * w);                 //                 bytes_pcg_iter
  out3[2] = B * (8.0 + 8.0 * w) + (N + 2.0 * B) * (36.0 * w + 4.0);   //      bytes_assembly_bsr
  return PL_OK;
}

int pl_time_kernel(pl_handle h, int which, int reps, double *avg_ms) {
  if (!valid(h) || !avg_ms || reps <= 0) return fail(PL_ERR_ARG, "pl_time_kernel: bad argument");
  if (h->opkind != 0 && which != 0 && which != 3) return fail(PL_ERR_STATE, "pl_time_kernel: DDM handles time K*p / PCG only");
  if (!h->assembled) return fail(PL_ERR_STATE, "pl_time_kernel: call pl_assemble first");
  if ((which == 0 || which == 3) && !h->have_bc) return fail(PL_ERR_STATE, "pl_time_kernel: call pl_set_bc first");
  if ((which == 2 || which == 4) && !h->have_bsr) return fail(PL_ERR_STATE, "pl_time_kernel: needs pl_assemble_bsr");
  PL_HIP(hipSetDevice(h->opt.device));
  const int64_t n6 = h->N * 6;
  int rc = ensure_hist(h, reps + 1);
  if (rc) return rc;
  // a well-defined operand: p = dinv (free dofs) -> nonzero everywhere that matters
  PL_HIP(hipMemcpyAsync(h->p.p, h->dinv.p, n6 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (which >= 7 && which <= 9) {   // the same operand, fp32-stored, in the z buffer; zeroed fp32 x / r in tmp
    hipLaunchKernelGGL(pl::k_to_float, dim3(grid_for(n6)), dim3(pl::kBlock), 0, h->stream, n6, h->dinv.p,
                       reinterpret_cast<float *>(h->z.p));
    PL_HIP(hipMemsetAsync(h->tmp.p, 0, n6 * sizeof(double), h->stream));
  }
  auto one = [&](int k) -> int {
    switch (which) {
      case 0: return launch_spmv(h, h->p.p, h->Ap.p, true, h->scal.p + pl::S_PAP * pl::kSlots);
      case 1: {
        int r1 = launch_records(h);
        return r1 ? r1 : build_palette(h);
      }
      case 2: return launch_bsr_fill(h, 0, h->stream);
      case 3: return pcg_iteration(h, k);
      case 4:
        hipLaunchKernelGGL(pl::k_bsr_spmv, dim3(grid_for(h->N)), dim3(pl::kBlock), 0, h->stream, h->N,
                           h->bsr_rowptr.p, h->bsr_col.p, h->bsr_vals.p, h->p.p, h->Ap.p);
        return PL_OK;
      case 5:   // multi-GPU: the interface all-reduce of a K*p alone (pack, RCCL, unpack) - collective call
        if (!h->dist.active) return fail(PL_ERR_STATE, "pl_time_kernel: 5/6 need pl_dist_init");
        return pl::dist_sum_shared(h->dist, h->Ap.p, h->stream, h->scal.p + pl::S_PAP * pl::kSlots, pl::kSlots)
                   ? fail(PL_ERR_HIP, "RCCL all-reduce failed") : PL_OK;
      case 6:   // multi-GPU: the coarse-residual all-reduce alone - collective call
        if (!h->dist.active || !h->coarse.ready) return fail(PL_ERR_STATE, "pl_time_kernel: 6 needs a coarse level");
        return pl::dist_sum_scalars(h->dist, h->coarse.tv, h->coarse.ncp, h->stream)
                   ? fail(PL_ERR_HIP, "RCCL all-reduce failed") : PL_OK;
      case 7:   // K*p on fp32-stored vectors (the operator of opts.precision = 1 / 2)
      case 8:   // one whole iteration of the fp32 inner PCG (precision = 1)
      case 9: { // one whole iteration of the mixed PCG (precision = 2: p, K*p fp32; x, r fp64)
        if (!(h->coarse.ready && h->tile.ready && choose_kernel(h) == 3))
          return fail(PL_ERR_STATE, "pl_time_kernel: 7/8/9 need the multi-level PCG on the tile kernel");
        float *p32 = reinterpret_cast<float *>(h->z.p), *Ap32 = p32 + n6;
        double *cur = h->scal.p + (k & 1) * pl::S_COUNT * pl::kSlots, *nxt = h->scal.p + ((k + 1) & 1) * pl::S_COUNT * pl::kSlots;
        int r7 = launch_spmv_f32(h, p32, Ap32, true, cur + pl::S_PAP * pl::kSlots);
        if (r7 || which == 7) return r7;
        if (which == 8) {
          float *x32 = reinterpret_cast<float *>(h->tmp.p), *r32 = x32 + n6;
          return pcg_tail_coarse_t<float, float>(h, cur, nxt, k, p32, (const float *)Ap32, x32, r32);
        }
        return pcg_tail_coarse_t<float, double>(h, cur, nxt, k, p32, (const float *)Ap32, h->x.p, h->r.p);
      }
      default: return fail(PL_ERR_ARG, "pl_time_kernel: unknown kernel id");
    }
  };
  for (int k = 0; k < 3; ++k) {   // warm-up
    rc = one(0);
    if (rc) return rc;
  }
  PL_HIP(hipEventRecord(h->ev0, h->stream));
  for (int k = 0; k < reps; ++k) {
    rc = one(k);
    if (rc) return rc;
  }
  PL_HIP(hipEventRecord(h->ev1, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  PL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *avg_ms = (double)ms / reps;
  return PL_OK;
}

int pl_debug_spd_solve(int device, int32_t n, const double *A, const double *b, double *x, double *quad,
                       int32_t fp32_factor) {
  return fp32_factor ? debug_spd_solve_t<float>(device, n, A, b, x, quad)
                     : debug_spd_solve_t<double>(device, n, A, b, x, quad);
}

int pl_lzone(int device, int64_t n_nodes, int64_t n_beams, const double *node_xyz, const int32_t *beam_conn,
             const double *beam_radius, double *lzone) {
  if (n_nodes <= 0 || n_beams <= 0 || !node_xyz || !beam_conn || !beam_radius || !lzone)
    return fail(PL_ERR_ARG, "pl_lzone: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(PL_ERR_NODEVICE, "pl_lzone: no HIP device visible (libpylattice_hip has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(PL_ERR_ARG, "pl_lzone: bad device ordinal");
  const int64_t nh = 2 * n_beams;
  std::vector<int64_t> ptr((size_t)n_nodes + 1, 0);
  for (int64_t h = 0; h < nh; ++h) {
    const int32_t v = beam_conn[h];
    if (v < 0 || v >= n_nodes) return fail(PL_ERR_ARG, "pl_lzone: node id out of range");
    ptr[(size_t)v + 1]++;
  }
  for (int64_t i = 0; i < n_nodes; ++i) ptr[i + 1] += ptr[i];
  std::vector<int32_t> half((size_t)nh);
  {
    std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
    for (int64_t h = 0; h < nh; ++h) half[(size_t)fill[beam_conn[h]]++] = (int32_t)h;
  }
  PL_HIP(hipSetDevice(device));
  DevBuf<double> dx, dr, dl;
  DevBuf<int32_t> dc, dh;
  DevBuf<int64_t> dp;
  PL_HIP(dx.alloc((size_t)n_nodes * 3));
  PL_HIP(dr.alloc(n_beams));
  PL_HIP(dl.alloc(nh));
  PL_HIP(dc.alloc(nh));
  PL_HIP(dh.alloc(nh));
  PL_HIP(dp.alloc(n_nodes + 1));
  PL_HIP(hipMemcpy(dx.p, node_xyz, (size_t)n_nodes * 3 * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(dr.p, beam_radius, n_beams * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(dc.p, beam_conn, nh * sizeof(int32_t), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(dh.p, half.data(), nh * sizeof(int32_t), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(dp.p, ptr.data(), (n_nodes + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(pl::k_lzone, dim3(grid_for(nh)), dim3(pl::kBlock), 0, nullptr, nh, dx.p, dc.p, dr.p, dp.p, dh.p,
                     dl.p);
  PL_HIP(hipGetLastError());
  PL_HIP(hipMemcpy(lzone, dl.p, nh * sizeof(double), hipMemcpyDeviceToHost));
  return PL_OK;
}

int pl_dist_unique_id_bytes(void) { return pl::dist_unique_id_bytes(); }
int pl_dist_unique_id(void *id_out) {
  if (!id_out) return fail(PL_ERR_ARG, "pl_dist_unique_id: null argument");
  return pl::dist_unique_id(id_out) ? fail(PL_ERR_HIP, "ncclGetUniqueId failed") : PL_OK;
}
int pl_dist_init(pl_handle h, int rank, int world, const void *unique_id, const int32_t *shared_local,
                 const int32_t *shared_global, int32_t n_shared, int32_t n_shared_global) {
  if (!valid(h) || !unique_id || world < 1 || rank < 0 || rank >= world || n_shared < 0)
    return fail(PL_ERR_ARG, "pl_dist_init: bad argument");
  if (n_shared > 0 && (!shared_local || !shared_global)) return fail(PL_ERR_ARG, "pl_dist_init: null index array");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<int32_t> loc(n_shared);
  for (int i = 0; i < n_shared; ++i) {
    if (shared_local[i] < 0 || shared_local[i] >= h->N || shared_global[i] < 0 || shared_global[i] >= n_shared_global)
      return fail(PL_ERR_ARG, "pl_dist_init: shared node index out of range");
    loc[i] = h->iperm[shared_local[i]];
  }
  int rc = pl::dist_init(h->dist, rank, world, unique_id, loc.data(), shared_global, n_shared, n_shared_global, h->N,
                         h->stream);
  if (rc) return fail(PL_ERR_HIP, "pl_dist_init: RCCL communicator setup failed (" + std::to_string(rc) + ")");
  if (h->coarse.enabled) {   // the tile level and the rank-local dense level leave shared nodes out
    std::vector<uint8_t> sh((size_t)h->N, 0);
    for (int i = 0; i < n_shared; ++i) sh[loc[i]] = 1;
    h->h_shared = sh;
    h->cond_ready = false;   // (re-selected at the next pl_set_bc; until then nothing is condensed)
    h->n_cond = 0;
    PL_HIP(hipMemcpy(h->sharedbits.p, sh.data(), sh.size(), hipMemcpyHostToDevice));
    h->coarseL.n_fix = -1;
  }
  h->assembled = false;
  return PL_OK;
}

int pl_dist_set_peers(pl_handle h, const int32_t *shared_peer) {
  if (!valid(h) || !shared_peer) return fail(PL_ERR_ARG, "pl_dist_set_peers: null argument");
  if (!h->dist.active) return fail(PL_ERR_STATE, "pl_dist_set_peers: call pl_dist_init first");
  PL_HIP(hipSetDevice(h->opt.device));
  int rc = pl::dist_set_peers(h->dist, shared_peer);
  if (rc) return fail(rc == 2 ? PL_ERR_ARG : PL_ERR_HIP, "pl_dist_set_peers: setup failed (" + std::to_string(rc) + ")");
  h->assembled = false;
  return PL_OK;
}

}  // extern "C"
