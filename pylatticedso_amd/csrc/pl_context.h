// Handle state of libpylattice_hip (struct pl_context) and the helpers every part of the host side uses.
// Included once, by pl_api.hip (single translation unit; the C ABI itself lives there).
#pragma once
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
#include "pl_rows.h"
#include "pl_dist.h"
#include "pl_coarse.h"
#include "pl_cg1.h"
#include "pl_small.h"
#include "pl_persist.h"
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
  hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr;   // row ranges of the inverse factor behind the factorisation chain
  hipEvent_t ev_fill1 = nullptr;                 // the part of the BSR fill that runs beside the chain
  hipStream_t side2 = nullptr;   // tile blocks of the 12-mode dense level beside its strain rows
  hipStream_t side_cu = nullptr; // bulk fills beside the factorisation chain: a stream that leaves some CUs of every XCD alone
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
  DevBuf<double> mult;   // per-strut stiffness multiplicity (pl_set_multiplicity); unallocated = 1 everywhere
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
  std::vector<uint8_t> h_fixedbits;   // host copy (device numbering): a pl_set_bc with the SAME Dirichlet set only uploads ubar / f
  DevBuf<double> ubar, f;
  // solver state
  DevBuf<double> diag, dinv, x, r, z, p, Ap, tmp, tmp2, scal, hist;
  DevBuf<double> xprev;               // previous converged solution (opts.warm_start)
  bool xprev_valid = false;
  // record palette: an attempt that fails (a graded lattice: too many distinct records) is not repeated at once
  int pal_fail_streak = 0, pal_skip_left = 0;
  bool pal_skipped = false;
  DevBuf<double> xprev2;              // the one before (opts.warm_start = 2: linear extrapolation of the design path)
  bool xprev2_valid = false;
  DevBuf<double> xprev3;              // (warm_start = 3, experiment: quadratic extrapolation)
  bool xprev3_valid = false;
  // warm_start = 4: the Galerkin start takes the best combination of the last kWarmMax solutions - xprev and a ring of older ones
  static constexpr int kWarmMax = 8;
  DevBuf<double> gh[kWarmMax - 1];    // ring of the solutions before xprev (gh_head = the newest of them)
  int gh_head = 0, gh_count = 0;
  DevBuf<double> gw[kWarmMax];        // masked copies (operands of the operator)
  DevBuf<double> gw_dots;             // [kWarmMax][kWarmMax + 1]: v_i . S v_j, then v_j . r
  DevBuf<double> cg1;   // single-reduction PCG: two reduction blocks, r_c.y_c slots, (gamma, alpha) pairs, Z^T s
  int hist_cap = 0;
  // LDS-tile operator
  pl::TilePlan tile;
  // DDM operator (pl_ddm.h): opkind = 1 replaces the strut operator by sum_c B^T S B
  int opkind = 0;
  int64_t ddm_cells = 0;
  int ddm_nb = 0;
  int ddm_n_S = 0;                       // matrices of the palette
  std::vector<int32_t> h_ddm_cell_nodes, h_ddm_cell_S;   // host copies (pl_ddm_update_matrices re-cuts the class tiles)
  DevBuf<int32_t> ddm_cell_nodes, ddm_cell_S;
  DevBuf<double> ddm_St;
  DevBuf<double> ddm_Sraw;            // upload buffer of pl_ddm_update_matrices (transposed on the device)
  DevBuf<int64_t> ddm_node_ptr;       // node -> (cell * nb + slot) entries: the atomic-free scatter of k_ddm_node_gather
  DevBuf<int32_t> ddm_node_ent;
  DevBuf<double> ddm_stage;           // [cells][6 nb] local products
  DevBuf<int32_t> ddm_order;          // cells sorted by matrix id (k_ddm_cell_product_lds)
  DevBuf<int32_t> ddm_tiles, ddm_tile_S;   // tiles of 16 cells of one matrix class (-1 = padding), their class (k_ddm_cell_product_mfma)
  int64_t ddm_n_tiles = 0;
  DevBuf<int32_t> ddm_tile_gidx;     // [tiles][ceil(m / 4)][64]: position in x of every A-operand entry (-1: zero)
  // assembled-Schur preconditioner of the DDM operator (opt.precond = 2): optional palette of its own + dense factor
  DevBuf<int32_t> ddm_cell_P;
  DevBuf<double> ddm_Pt;
  bool ddm_have_P = false;
  DevBuf<double> dd_A, dd_Lf, dd_W, dd_Wt, dd_Dinv, dd_tv;
  DevBuf<int> dd_info;
  int dd_n = 0;          // padded order of the dense matrix (0: not allocated)
  int dd_bw = 0;         // its block bandwidth in the caller's node numbering (from the cells' node spans)
  bool dd_ready = false;
  DevBuf<double> dd_B;   // node-block Jacobi of the DDM operator (opt.precond = 3): inverted 6 x 6 blocks
  bool dd_blocks = false;
  // two-level preconditioner of a DDM handle (opt.precond = 4, pl_ddm.h): the node blocks + a dense level of 12 modes per
  // aggregate of nodes; geometry and aggregates from pl_ddm_set_geometry
  DevBuf<double> dd2_xyz, dd2_cen;
  DevBuf<int32_t> dd2_agg, dd2_ptr, dd2_nodes;   // aggregate of every node; the nodes aggregate by aggregate (CSR)
  pl::Coarse dd2;                                // its matrices (A_c, factors, r_c, y_c), ncp, bw_blocks, w16
  int dd2_n_agg = 0;
  bool dd2_plan = false, dd2_ready = false;
  // device-side stop of the CG of a DDM handle (k_pcg_direction): the flag, and the numbers of the solve in flight
  DevBuf<int> stop_flag;
  bool stop_use = false;
  double stop_thresh = 0.0;
  // record palette (pl_palette.h)
  DevBuf<unsigned long long> pal_keys;
  DevBuf<int> pal_owner, pal_flags;
  DevBuf<uint16_t> pal_id;
  DevBuf<int> pal_dense_of_slot;      // hash slot -> dense palette id (0 .. pal_entries-1)
  DevBuf<pl::Record> pal_dense;       // the first kPalDenseMax entries, densely numbered (LDS table of k_spmv_tile_lds)
  DevBuf<uint32_t> vword;             // per strut visit: local rows | dense palette id | condensed-end bits (k_visit_words)
  DevBuf<uint32_t> vword_dir;         // the same with the direction-palette entry (streaming form of the LDS-resident K*p)
  bool vword_dir_fresh = false;
  bool pal_lds = false;               // the LDS-resident K*p applies (palette holds, <= kPalDenseMax entries, visit plan)
  // K*p by rows (pl_rows.h): plan, per-half-visit words (static bits | dense palette id), palette in both orientations
  pl::RowPlan rows;
  DevBuf<uint32_t> rword;
  DevBuf<pl::Record> pal_dense2;
  bool pal_rows = false;              // the row kernel applies (same conditions as pal_lds, plus the row plan)
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
  DevBuf<int32_t> ckeep;          // the other nodes, ascending (k_pcg_direction_flat maps its lanes onto these)
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
  // short form of the iteration on small lattices (pl_small.h): second p buffer, ring of four scalar sets, two r_c buffers
  DevBuf<double> p2, small_scal, small_rc;
  bool small_use = false;    // the running solve takes the short form (solver_plan)
  // the whole PCG loop as one persistent launch (pl_persist.h, opts.short_iteration = 2)
  bool persist_use = false;
  DevBuf<double> ps_Ug, ps_red;
  DevBuf<unsigned> ps_flags;          // [2 n_tiles + 4]: flagU | flagR | err | iterations, converged
  DevBuf<unsigned long long> ps_dbg;  // [8] per-phase clock ticks of workgroup 0 (PL_PERSIST_DEBUG=1 prints them)
  DevBuf<int32_t> ps_agg_ptr, ps_agg_idx;
  int ps_n_agg = 0;
  int last_iterations = 0;   // of the previous converged pcg_solve on this handle (hint for the first convergence check)
  int64_t n_cond = 0;
  bool cond_ready = false;   // K_cc^-1 valid for the current records and mask
  bool cond_use = false;     // the running solve eliminates them (fp64 PCG and precision = 1)
  int cond_agree = -1;       // several GPUs: every rank has nodes to eliminate (1) / some rank has none, so nobody does (0) /
                             // not yet agreed since the last pl_set_bc (-1): the solver's collectives must match on all ranks
  // multi-GPU
  pl::Dist dist;
  // exchange / compute overlap of K*p: the tiles that own interface rows run first, their rows travel on the
  // communication stream while the interior tiles run on the main one
  std::vector<int32_t> h_tile_start;
  DevBuf<int32_t> ov_iface, ov_inner;      // ascending tile lists
  int64_t n_ov_iface = 0, n_ov_inner = 0;
  bool ov_ready = false;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_ov_a = nullptr, ev_ov_x = nullptr;

  pl_stats_t last{};
  double ms_assembly = 0.0;
  double *pin6 = nullptr;      // pinned staging buffer of the vector transfers (pl_context.h: staging)
  size_t pin6_n = 0;
  // periodic constraints (pl_set_periodic; one-cell homogenisation): groups of nodes that share their six dofs.  The Jacobi
  // PCG then runs on Q K Q with Q = the orthogonal projector "average over every group" (pl_solver.h)
  DevBuf<int32_t> per_ptr, per_nodes;
  int64_t n_per_groups = 0;
  double *pinB = nullptr;      // pinned staging buffer of the per-strut transfers (radii, sensitivities)
  DevBuf<double> sens_out;     // [B] per-strut sensitivities of pl_sens (kept: a design loop calls it every iteration)
  DevBuf<double> usol;         // composed solution of the last pl_solve (pl_sens with u = NULL reads it)
  bool usol_valid = false;

  ~pl_context() {
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (ev_chol) (void)hipEventDestroy(ev_chol);
    if (ev_fill1) (void)hipEventDestroy(ev_fill1);
    if (ev_p0) (void)hipEventDestroy(ev_p0);
    if (ev_p1) (void)hipEventDestroy(ev_p1);
    if (ev_t0) (void)hipEventDestroy(ev_t0);
    if (ev_t1) (void)hipEventDestroy(ev_t1);
    if (pin6) (void)hipHostFree(pin6);
    if (pinB) (void)hipHostFree(pinB);
    if (ev_ov_a) (void)hipEventDestroy(ev_ov_a);
    if (ev_ov_x) (void)hipEventDestroy(ev_ov_x);
    if (comm_stream) (void)hipStreamDestroy(comm_stream);
    if (side2) (void)hipStreamDestroy(side2);
    if (side_cu) (void)hipStreamDestroy(side_cu);
    if (side) (void)hipStreamDestroy(side);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

// ----------------------------------------------------------------------------------------------------------
// host <-> device vector transfer in caller numbering
// ----------------------------------------------------------------------------------------------------------
// Pinned staging buffer of the handle (6N doubles, allocated at the first transfer): a copy into pageable memory runs at
// a few GB/s and blocks the host until the stream drains; through pinned memory it is one DMA at PCIe rate, and the
// permutation between caller and device numbering runs on all host cores beside nothing else.
int staging(pl_context *c, double **out) {
  const size_t n6 = (size_t)c->N * 6;
  if (!c->pin6 || c->pin6_n < n6) {
    if (c->pin6) (void)hipHostFree(c->pin6);
    c->pin6 = nullptr;
    void *p = nullptr;
    PL_HIP(hipHostMalloc(&p, n6 * sizeof(double), hipHostMallocDefault));
    c->pin6 = static_cast<double *>(p);
    c->pin6_n = n6;
  }
  *out = c->pin6;
  return PL_OK;
}

int stagingB(pl_context *c, double **out) {
  if (!c->pinB) {
    void *p = nullptr;
    PL_HIP(hipHostMalloc(&p, (size_t)std::max<int64_t>(c->B, 1) * sizeof(double), hipHostMallocDefault));
    c->pinB = static_cast<double *>(p);
  }
  *out = c->pinB;
  return PL_OK;
}

int upload6(pl_context *c, const double *host, double *dev, std::vector<double> &) {
  const size_t n6 = (size_t)c->N * 6;
  double *st = nullptr;
  int rc = staging(c, &st);
  if (rc) return rc;
  if (!c->reordered) {
    pl::parallel_for((int64_t)n6, [&](int64_t b, int64_t e, unsigned) { std::memcpy(st + b, host + b, (e - b) * sizeof(double)); },
                     1 << 18);
  } else {
    pl::parallel_for(c->N, [&](int64_t b, int64_t e, unsigned) {
      for (int64_t i = b; i < e; ++i) std::memcpy(st + 6 * i, host + 6 * (size_t)c->perm[i], 6 * sizeof(double));
    }, 1 << 15);
  }
  PL_HIP(hipMemcpyAsync(dev, st, n6 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  PL_HIP(hipStreamSynchronize(c->stream));
  return PL_OK;
}

int download6(pl_context *c, const double *dev, double *host) {
  const size_t n6 = (size_t)c->N * 6;
  double *st = nullptr;
  int rc = staging(c, &st);
  if (rc) return rc;
  PL_HIP(hipMemcpyAsync(st, dev, n6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  PL_HIP(hipStreamSynchronize(c->stream));
  if (!c->reordered) {
    pl::parallel_for((int64_t)n6, [&](int64_t b, int64_t e, unsigned) { std::memcpy(host + b, st + b, (e - b) * sizeof(double)); },
                     1 << 18);
  } else {
    pl::parallel_for(c->N, [&](int64_t b, int64_t e, unsigned) {
      for (int64_t i = b; i < e; ++i) std::memcpy(host + 6 * (size_t)c->perm[i], st + 6 * i, 6 * sizeof(double));
    }, 1 << 15);
  }
  return PL_OK;
}

}  // namespace
