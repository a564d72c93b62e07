// block.inc.hpp -- block-diagonal / block-triangular preconditioners and the outer Krylov solve on a
// block system (included at the end of gmg_amd.hip; everything runs on ONE stream of ONE device).
//
// This is SURVEY 8(f)(2): the glue that calls the GMG hot path once per outer FGMRES iteration in
// the reference's Stokes / Navier-Stokes / Darcy applications (test/Applications/StokesGMG.jl:142-153):
//   BlockDiagonalSolver   solve!: src/BlockSolvers/BlockDiagonalSolvers.jl:165-177
//   BlockTriangularSolver solve!: src/BlockSolvers/BlockTriangularSolvers.jl:186-242
// Block vectors are contiguous on the device, block i at off[i]..off[i+1].  Diagonal-block solvers:
// a set-up gmg handle, CGSolver(JacobiLinearSolver()), LUSolver() (small blocks) or JacobiLinearSolver().

struct BlockDiag {
  int kind = 0;                 // gmg_block_diag_kind, 0 = not set
  gmg_solver *g = nullptr;      // GMG_BLOCK_GMG (borrowed)
  HostCSR hM;                   // matrix of the block solver (MatrixBlock / BiformBlock); empty: system block (i,i)
  bool hasM = false;
  DevCSR M;
  double *dinv = nullptr, *Minv = nullptr;
  int maxiter = 1000;
  double atol = 1e-12, rtol = 1e-6;
  double *w = nullptr, *p = nullptr, *z = nullptr, *r = nullptr;   // CGSolvers.jl:42-48
  ConvLog log;
};

struct gmg_block_solver {
  gmg_solver eng;               // stream, allocator, reductions, SpMV launchers (no levels)
  int nb = 0;
  std::vector<int64_t> off;
  int kind = GMG_BLOCK_DIAGONAL;
  std::vector<BlockDiag> diag;
  std::map<std::pair<int, int>, HostCSR> hsys, hpre;
  std::map<std::pair<int, int>, DevCSR> sys, pre;
  std::vector<double> coeff;    // nb*nb row-major, BlockTriangularSolvers.jl:66 (default 1.0)
  double *w = nullptr, *y = nullptr, *tmp = nullptr;               // BlockTriangularSolvers.jl:145-150
  double *kw = nullptr, *kp = nullptr, *kz = nullptr, *kr = nullptr, *st_b = nullptr, *st_x = nullptr;
  std::vector<double *> fg_V, fg_Z;
  bool setup_done = false;
  std::string err;
  // distributed runs (one handle per rank): block vectors hold the OWNED entries of every block; a block whose columns reach
  // into other ranks' entries gets an exchange plan and a work vector with ghost space ([own | ghost], as on GMG levels)
  std::vector<HaloPlan> plan;
  std::vector<double *> xg;

  int64_t N() const { return off.empty() ? 0 : off.back(); }
  int64_t bsize(int i) const { return off[i + 1] - off[i]; }
  int64_t ncols_of(int j) const { return bsize(j) + ((size_t)j < plan.size() && plan[j].present ? plan[j].n_ghost : 0); }
  bool distributed() const { return eng.comm.nranks > 1; }
  // x_j with its ghost entries filled (consistent!): own part copied into the work vector, halo exchanged
  const double *with_ghosts(int j, const double *xj)
  {
    // (a rank without ghosts of its own may still have to SEND: discontinuous pressure spaces -- only "no neighbours" skips the exchange)
    if (!distributed() || (size_t)j >= plan.size() || !plan[j].present || plan[j].nbr.empty()) return xj;
    eng.copy(xg[j], xj, bsize(j));
    eng.exchange_plan(plan[j], xg[j], eng.stream);
    return xg[j];
  }

  void detach()
  {
    for (auto &D : diag)
      if (D.g) {                                           // D.g is nulled by block_forget when its owner destroys it first
        if (D.g->stream == eng.stream) D.g->stream = D.g->own_stream;
        if (D.g->attached_to == this) D.g->attached_to = nullptr;
      }
  }

  void setup()
  {
    HIP_CHECK(hipSetDevice(eng.device));
    detach();
    eng.free_all();
    sys.clear(); pre.clear();
    eng.read_tuning();
    eng.init_reductions();
    int64_t nmax = 1;
    for (int i = 0; i < nb; ++i) nmax = std::max(nmax, bsize(i));
    xg.assign((size_t)nb, nullptr);
    if (distributed()) {
      plan.resize((size_t)nb);
      for (int j = 0; j < nb; ++j)
        if (plan[j].present) {
          REQUIRE(plan[j].n_own == bsize(j), GMG_ERR_INVALID, "partition of block " + std::to_string(j) + " does not match its size");
          plan[j].d_snd_idx = nullptr; plan[j].d_sendbuf = nullptr; plan[j].d_recvbuf = nullptr;
          eng.alloc_plan_buffers(plan[j]);
          xg[j] = eng.dvec(bsize(j) + plan[j].n_ghost);
        }
    }
    for (auto &kv : hsys) sys[kv.first] = eng.upload_csr(kv.second);
    for (auto &kv : hpre) pre[kv.first] = eng.upload_csr(kv.second);
    for (int i = 0; i < nb; ++i) {
      BlockDiag &D = diag[i];
      REQUIRE(D.kind != 0, GMG_ERR_STATE, "no solver set for diagonal block " + std::to_string(i));
      const int64_t n = bsize(i);
      if (D.kind == GMG_BLOCK_GMG) {
        gmg_solver *g = D.g;
        REQUIRE(g && g->setup_done, GMG_ERR_STATE, "the GMG handle of block " + std::to_string(i) + " is not set up");
        REQUIRE(g->device == eng.device, GMG_ERR_INVALID, "GMG handle lives on another device");
        REQUIRE(g->comm.nranks == eng.comm.nranks, GMG_ERR_INVALID, "the GMG handle of block " + std::to_string(i) + " and the block solver must span the same ranks");
        REQUIRE(g->lev[0].n == n, GMG_ERR_INVALID, "GMG handle size does not match block " + std::to_string(i));
        HIP_CHECK(hipStreamSynchronize(g->stream));
        REQUIRE(g->attached_to == nullptr || g->attached_to == this, GMG_ERR_STATE, "GMG handle is already attached to another block solver");
        g->stream = eng.stream;                           // one stream: block glue and V-cycles are ordered without events
        g->attached_to = this;
        continue;
      }
      const HostCSR *src = nullptr;
      if (D.hasM) src = &D.hM;
      else {
        auto it = hsys.find({i, i});
        REQUIRE(it != hsys.end(), GMG_ERR_STATE, "block " + std::to_string(i) + ": no solver matrix and no system block (i,i)");
        src = &it->second;
      }
      REQUIRE(src->nrows == n && src->ncols == ncols_of(i), GMG_ERR_INVALID, "solver matrix of block " + std::to_string(i) + " has the wrong shape");
      if (D.kind == GMG_BLOCK_LU) {
        REQUIRE(!distributed(), GMG_ERR_UNSUPPORTED, "LUSolver() diagonal blocks are single-GPU (a dense inverse of a distributed block is not formed)");
        D.Minv = eng.build_dense_inverse(*src, "diagonal block " + std::to_string(i));
        continue;
      }
      // LinearSystemBlock: the solver streams the system's own (i,i) block, uploaded once
      DevCSR &M = D.hasM ? (D.M = eng.upload_csr(*src)) : sys[{i, i}];
      int nzero = 0;
      D.dinv = eng.build_inv_diag(M, nzero);
      REQUIRE(nzero == 0, GMG_ERR_SINGULAR, "zero diagonal entry in block " + std::to_string(i));
      if (D.hasM) eng.drop_csr_stream(D.M);
      if (D.kind == GMG_BLOCK_CG_JACOBI) { D.w = eng.dvec(n); D.p = eng.dvec(n); D.z = eng.dvec(n); D.r = eng.dvec(n); }
    }
    for (auto &kv : sys) eng.drop_csr_stream(kv.second);
    for (auto &kv : pre) eng.drop_csr_stream(kv.second);
    for (int i = 0; i < nb; ++i)
      if (!diag[i].hasM && (diag[i].kind == GMG_BLOCK_CG_JACOBI || diag[i].kind == GMG_BLOCK_JACOBI)) diag[i].M = sys[{i, i}];
    const int64_t n = N();
    w = eng.dvec(n); y = eng.dvec(n); tmp = eng.dvec(nmax);
    kw = eng.dvec(n); kp = eng.dvec(n); kz = eng.dvec(n); kr = eng.dvec(n);
    st_b = eng.dvec(n); st_x = eng.dvec(n);
    fg_V.clear(); fg_Z.clear();
    HIP_CHECK(hipStreamSynchronize(eng.stream));
    setup_done = true;
  }

  const DevCSR *sys_block(int i, int j) const
  {
    auto it = sys.find({i, j});
    return it == sys.end() ? nullptr : &it->second;
  }
  // off-diagonal block of the PRECONDITIONER: an explicit one (MatrixBlock / BiformBlock) or the
  // system's (LinearSystemBlock, BlockSolverInterfaces.jl)
  const DevCSR *pre_block(int i, int j) const
  {
    auto it = pre.find({i, j});
    if (it != pre.end()) return &it->second;
    return sys_block(i, j);
  }

  // y = A x, block-wise
  void sys_apply(const double *x, double *yv)
  {
    for (int i = 0; i < nb; ++i) {
      bool first = true;
      double *yi = yv + off[i];
      for (int j = 0; j < nb; ++j) {
        const DevCSR *M = sys_block(i, j);
        if (!M) continue;
        const double *xj = with_ghosts(j, x + off[j]);
        if (first) eng.spmv_set(*M, xj, yi);
        else eng.spmv_addto(*M, xj, tmp, yi);
        first = false;
      }
      if (first) eng.zero(yi, bsize(i));
    }
  }
  // r = b - A x
  void sys_resid(const double *x, const double *b, double *r)
  {
    for (int i = 0; i < nb; ++i) {
      bool first = true;
      double *ri = r + off[i];
      for (int j = 0; j < nb; ++j) {
        const DevCSR *M = sys_block(i, j);
        if (!M) continue;
        const double *xj = with_ghosts(j, x + off[j]);
        if (first) eng.spmv_resid(*M, xj, b + off[i], ri);
        else eng.spmv_sub(*M, xj, ri);
        first = false;
      }
      if (first) eng.copy(ri, b + off[i], bsize(i));
    }
  }

  // solve!(yi,nsi,wi) for one diagonal block
  void diag_solve(int i, double *yi, const double *wi)
  {
    BlockDiag &D = diag[i];
    const int64_t n = bsize(i);
    switch (D.kind) {
    case GMG_BLOCK_GMG:
      if (D.g->comm.nranks > 1) {                           // distributed handle: its solution vector carries ghost space
        if (D.g->mode == GMG_MODE_SOLVER) eng.copy(D.g->cg_x, yi, n);
        D.g->gmg_solve_dev(D.g->cg_x, wi, -1.0);
        eng.copy(yi, D.g->cg_x, n);
      } else
      D.g->gmg_solve_dev(yi, wi, -1.0);
      break;
    case GMG_BLOCK_CG_JACOBI: {                           // yi keeps its previous content: CG's initial guess
      KrylovOps ops;
      ops.resid = [&, i](double *x, const double *b, double *r) { eng.spmv_resid(D.M, with_ghosts(i, x), b, r); };
      ops.apply = [&, i](double *x, double *yv) { eng.spmv_set(D.M, with_ghosts(i, x), yv); };
      ops.precond = [&, n](double *z, const double *r, double) {
        hipLaunchKernelGGL(jacobi_apply_kernel, dim3(gmg_solver::grid_for(n)), dim3(256), 0, eng.stream, n, D.dinv, r, z);
        HIP_CHECK(hipGetLastError());
      };
      D.log.configure(D.maxiter, D.atol, D.rtol);
      cg_core(eng, n, wi, yi, D.w, D.p, D.z, D.r, ops, false, D.log);
      break;
    }
    case GMG_BLOCK_LU:
      eng.dense_solve(D.Minv, (int)n, wi, yi);
      break;
    case GMG_BLOCK_JACOBI:
      hipLaunchKernelGGL(jacobi_apply_kernel, dim3(gmg_solver::grid_for(n)), dim3(256), 0, eng.stream, n, D.dinv, wi, yi);
      HIP_CHECK(hipGetLastError());
      break;
    default:
      throw GmgError{GMG_ERR_STATE, "diagonal block solver not set"};
    }
  }

  // solve!(x,ns,b): BlockDiagonalSolvers.jl:165-177 / BlockTriangularSolvers.jl:186-242
  void precond_apply(double *x, const double *b)
  {
    for (int step = 0; step < nb; ++step) {
      const int iB = (kind == GMG_BLOCK_UPPER) ? nb - 1 - step : step;       // :218 NB:-1:1 / :190 1:NB
      const int64_t n = bsize(iB);
      const double *rhs = b + off[iB];
      double *yi = y + off[iB];
      if (kind != GMG_BLOCK_DIAGONAL) {
        double *wi = w + off[iB];
        bool touched = false;
        const int j0 = (kind == GMG_BLOCK_UPPER) ? iB + 1 : 0, j1 = (kind == GMG_BLOCK_UPPER) ? nb : iB;
        for (int jB = j0; jB < j1; ++jB) {
          const double cij = coeff[(size_t)iB * nb + jB];
          const double eps = std::nextafter(std::fabs(cij), INFINITY) - std::fabs(cij);   // eps(cij)
          const DevCSR *M = pre_block(iB, jB);
          if (!(std::fabs(cij) > eps) || !M) continue;                       // :194,223
          if (!touched) { eng.copy(wi, rhs, n); touched = true; }            // :192,221 copy!(wi,bi)
          const double *xj = with_ghosts(jB, x + off[jB]);
          if (cij == 1.0) eng.spmv_sub(*M, xj, wi);                          // :196,225 mul!(wi,M,xj,-cij,1.0)
          else {
            eng.spmv_set(*M, xj, tmp);
            hipLaunchKernelGGL(axpy_kernel, dim3(gmg_solver::grid_for(n)), dim3(256), 0, eng.stream, n, -cij, tmp, wi);
            HIP_CHECK(hipGetLastError());
          }
        }
        if (touched) rhs = wi;                                               // no contribution: wi == bi, skip the copy
      }
      diag_solve(iB, yi, rhs);                                               // :202-205,231-234 solve!(yi,nsi,wi)
      eng.copy(x + off[iB], yi, n);                                          // copy!(xi,yi)
    }
  }

  KrylovOps ops(bool use_precond)
  {
    KrylovOps o;
    o.resid = [this](double *x, const double *b, double *r) { sys_resid(x, b, r); };
    o.apply = [this](double *x, double *yv) { sys_apply(x, yv); };
    if (use_precond) o.precond = [this](double *z, const double *r, double) { precond_apply(z, r); };
    return o;
  }
};

// gmg_destroy on a handle that a block solver still borrows: drop the pointer, the block solver must be set up again
static void block_forget(gmg_block_solver *B, gmg_solver *g)
{
  for (auto &D : B->diag)
    if (D.g == g) { D.g = nullptr; D.kind = 0; B->setup_done = false; }
  g->attached_to = nullptr;
  if (g->stream == B->eng.stream) {
    (void)hipStreamSynchronize(B->eng.stream);
    g->stream = g->own_stream;
  }
}

namespace {
template <typename F>
int guarded_b(gmg_block_handle_t h, F &&f)
{
  try {
    if (h) HIP_CHECK(hipSetDevice(h->eng.device));
    f();
    return GMG_OK;
  } catch (const GmgError &e) {
    if (h) h->err = e.msg;
    g_last_error = e.msg;
    return e.code;
  } catch (const std::bad_alloc &) {
    if (h) h->err = "host allocation failed";
    g_last_error = "host allocation failed";
    return GMG_ERR_ALLOC;
  } catch (const std::exception &e) {
    if (h) h->err = e.what();
    g_last_error = e.what();
    return GMG_ERR_INVALID;
  }
}
void check_block(gmg_block_handle_t h, int i)
{
  REQUIRE(h, GMG_ERR_INVALID, "null handle");
  REQUIRE(i >= 0 && i < h->nb, GMG_ERR_INVALID, "block index out of range");
}
void check_block_ready(gmg_block_handle_t h)
{
  REQUIRE(h, GMG_ERR_INVALID, "null handle");
  REQUIRE(h->setup_done, GMG_ERR_STATE, "gmg_block_setup has not been called (numerical_setup missing)");
}
} // namespace

extern "C" {

int gmg_block_create(gmg_block_handle_t *out, int nblocks, const int64_t *block_sizes, int kind, int device_id)
{
  return guarded_b(nullptr, [&] {
    REQUIRE(out, GMG_ERR_INVALID, "null handle pointer");
    *out = nullptr;
    REQUIRE(nblocks >= 1 && block_sizes, GMG_ERR_INVALID, "at least one block required");
    REQUIRE(kind == GMG_BLOCK_DIAGONAL || kind == GMG_BLOCK_LOWER || kind == GMG_BLOCK_UPPER, GMG_ERR_INVALID,
            "kind must be diagonal, :lower or :upper (BlockTriangularSolvers.jl:63)");
    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    REQUIRE(ndev > 0, GMG_ERR_HIP, "no HIP device visible: libgmgamd has no CPU path");
    REQUIRE(device_id >= 0 && device_id < ndev, GMG_ERR_INVALID, "device_id out of range");
    HIP_CHECK(hipSetDevice(device_id));
    gmg_block_solver *s = new gmg_block_solver();
    s->eng.device = device_id;
    s->nb = nblocks;
    s->kind = kind;
    s->off.assign((size_t)nblocks + 1, 0);
    for (int i = 0; i < nblocks; ++i) {
      if (block_sizes[i] < 0) { delete s; throw GmgError{GMG_ERR_INVALID, "negative block size"}; }
      s->off[i + 1] = s->off[i] + block_sizes[i];
    }
    s->diag.resize(nblocks);
    s->coeff.assign((size_t)nblocks * nblocks, 1.0);
    hipError_t e = hipStreamCreateWithFlags(&s->eng.stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete s;
      throw GmgError{GMG_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)};
    }
    s->eng.own_stream = s->eng.stream;
    *out = s;
  });
}

int gmg_block_destroy(gmg_block_handle_t h)
{
  // borrowed GMG handles destroyed earlier have already been forgotten (block_forget): no dangling pointer is touched
  if (!h) return GMG_OK;
  (void)hipSetDevice(h->eng.device);
  (void)hipStreamSynchronize(h->eng.stream);
  h->detach();                                            // GMG handles get their own stream back
  h->eng.free_all();
  for (auto &P : h->plan) {
    if (P.h_send) (void)hipHostFree(P.h_send);
    if (P.h_recv) (void)hipHostFree(P.h_recv);
  }
  if (h->eng.comm.kind == COMM_RCCL && h->eng.comm.comm) (void)h->eng.comm.api.CommDestroy(h->eng.comm.comm);
  if (h->eng.comm_stream) (void)hipStreamSynchronize(h->eng.comm_stream);   // created by gmg_block_comm_init_* on the engine
  if (h->eng.ev_ready) (void)hipEventDestroy(h->eng.ev_ready);
  if (h->eng.ev_done) (void)hipEventDestroy(h->eng.ev_done);
  if (h->eng.comm_stream) (void)hipStreamDestroy(h->eng.comm_stream);
  if (h->eng.h_rep_full) (void)hipHostFree(h->eng.h_rep_full);
  if (h->eng.h_scalars) (void)hipHostFree(h->eng.h_scalars);
  if (h->eng.own_stream) (void)hipStreamDestroy(h->eng.own_stream);
  delete h;
  return GMG_OK;
}

const char *gmg_block_last_error(gmg_block_handle_t h) { return h ? h->err.c_str() : g_last_error.c_str(); }

// ---- distributed block systems: one handle per rank ------------------------------------------------------------------
int gmg_block_comm_init_rccl(gmg_block_handle_t h, const char *rccl_path, const char *unique_id128, int rank, int nranks)
{
  if (!h) return GMG_ERR_INVALID;
  const int st = gmg_comm_init_rccl(&h->eng, rccl_path, unique_id128, rank, nranks);
  if (st != GMG_OK) h->err = h->eng.err;
  h->setup_done = false;
  return st;
}
int gmg_block_comm_init_host(gmg_block_handle_t h, int rank, int nranks, gmg_host_exchange_fn xfn, gmg_host_allreduce_fn rfn, void *ctx)
{
  if (!h) return GMG_ERR_INVALID;
  const int st = gmg_comm_init_host(&h->eng, rank, nranks, xfn, rfn, ctx);
  if (st != GMG_OK) h->err = h->eng.err;
  h->setup_done = false;
  return st;
}
int gmg_block_comm_set_loopback(gmg_block_handle_t h, int virtual_nranks)
{
  if (!h) return GMG_ERR_INVALID;
  const int st = gmg_comm_set_loopback(&h->eng, virtual_nranks);
  if (st != GMG_OK) h->err = h->eng.err;
  h->setup_done = false;
  return st;
}
int gmg_block_set_partition(gmg_block_handle_t h, int j, int64_t n_own, int64_t n_ghost, int nnbr, const int32_t *nbr_rank,
                            const int64_t *snd_ptr, const int64_t *snd_idx, const int64_t *rcv_ptr)
{
  return guarded_b(h, [&] {
    check_block(h, j);
    REQUIRE(h->eng.comm.nranks > 1, GMG_ERR_STATE, "gmg_block_comm_init_* first");
    REQUIRE(n_own == h->bsize(j), GMG_ERR_INVALID, "n_own must equal the block size given to gmg_block_create");
    if (h->plan.size() < (size_t)h->nb) h->plan.resize((size_t)h->nb);
    fill_plan(h->plan[j], h->eng.comm, n_own, n_ghost, nnbr, nbr_rank, snd_ptr, snd_idx, rcv_ptr);
    h->setup_done = false;
  });
}

int gmg_block_set_system_block(gmg_block_handle_t h, int i, int j, int64_t nrows, int64_t ncols, int64_t nnz, const void *ptr,
                               const void *idx, const double *val, int layout, int index_base, int index_bytes)
{
  return guarded_b(h, [&] {
    check_block(h, i); check_block(h, j);
    REQUIRE(nrows == h->bsize(i) && ncols == h->ncols_of(j), GMG_ERR_INVALID, "block shape does not match the block sizes (columns = own + ghost entries of block j)");
    h->hsys[{i, j}] = convert_input(nrows, ncols, nnz, ptr, idx, val, layout, index_base, index_bytes);
    h->setup_done = false;
  });
}

int gmg_block_set_precond_block(gmg_block_handle_t h, int i, int j, int64_t nrows, int64_t ncols, int64_t nnz, const void *ptr,
                                const void *idx, const double *val, int layout, int index_base, int index_bytes)
{
  return guarded_b(h, [&] {
    check_block(h, i); check_block(h, j);
    REQUIRE(i != j, GMG_ERR_INVALID, "diagonal blocks are given with gmg_block_set_diag_*");
    REQUIRE(nrows == h->bsize(i) && ncols == h->ncols_of(j), GMG_ERR_INVALID, "block shape does not match the block sizes (columns = own + ghost entries of block j)");
    h->hpre[{i, j}] = convert_input(nrows, ncols, nnz, ptr, idx, val, layout, index_base, index_bytes);
    h->setup_done = false;
  });
}

int gmg_block_set_coeff(gmg_block_handle_t h, int i, int j, double c)
{
  return guarded_b(h, [&] {
    check_block(h, i); check_block(h, j);
    h->coeff[(size_t)i * h->nb + j] = c;
  });
}

int gmg_block_set_diag_gmg(gmg_block_handle_t h, int i, gmg_handle_t g)
{
  return guarded_b(h, [&] {
    check_block(h, i);
    REQUIRE(g, GMG_ERR_INVALID, "null GMG handle");
    BlockDiag &D = h->diag[i];
    if (D.g && D.g != g) {
      if (D.g->stream == h->eng.stream) D.g->stream = D.g->own_stream;
      if (D.g->attached_to == h) D.g->attached_to = nullptr;
    }
    D = BlockDiag();
    D.kind = GMG_BLOCK_GMG; D.g = g;
    h->setup_done = false;
  });
}

int gmg_block_set_diag_solver(gmg_block_handle_t h, int i, int kind, int maxiter, double atol, double rtol)
{
  return guarded_b(h, [&] {
    check_block(h, i);
    REQUIRE(kind == GMG_BLOCK_CG_JACOBI || kind == GMG_BLOCK_LU || kind == GMG_BLOCK_JACOBI, GMG_ERR_INVALID, "bad diagonal solver kind");
    REQUIRE(maxiter >= 0, GMG_ERR_INVALID, "maxiter < 0");
    BlockDiag &D = h->diag[i];
    if (D.g && D.g->stream == h->eng.stream) D.g->stream = D.g->own_stream;
    HostCSR keep;
    const bool had = D.hasM;
    if (had) keep = std::move(D.hM);
    D = BlockDiag();
    if (had) { D.hM = std::move(keep); D.hasM = true; }
    D.kind = kind; D.maxiter = maxiter; D.atol = atol; D.rtol = rtol;
    h->setup_done = false;
  });
}

int gmg_block_set_diag_matrix(gmg_block_handle_t h, int i, int64_t n, int64_t nnz, const void *ptr, const void *idx,
                              const double *val, int layout, int index_base, int index_bytes)
{
  return guarded_b(h, [&] {
    check_block(h, i);
    REQUIRE(n == h->bsize(i), GMG_ERR_INVALID, "matrix size does not match the block size");
    h->diag[i].hM = convert_input(n, h->ncols_of(i), nnz, ptr, idx, val, layout, index_base, index_bytes);   // distributed: own rows x [own | ghost]
    h->diag[i].hasM = true;
    h->setup_done = false;
  });
}

int gmg_block_setup(gmg_block_handle_t h)
{
  return guarded_b(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    h->setup();
  });
}

int gmg_block_precond_apply(gmg_block_handle_t h, const double *b, double *x, int memspace)
{
  return guarded_b(h, [&] {
    check_block_ready(h);
    REQUIRE(b && x, GMG_ERR_INVALID, "null vector");
    const int64_t n = h->N();
    const double *db = h->eng.in_vec(b, n, memspace, h->st_b);
    double *dx = (memspace == GMG_MEM_DEVICE) ? x : h->st_x;
    h->precond_apply(dx, db);
    h->eng.out_vec(x, dx, n, memspace);
  });
}

int gmg_block_apply_system(gmg_block_handle_t h, const double *x, double *y, int memspace)
{
  return guarded_b(h, [&] {
    check_block_ready(h);
    REQUIRE(x && y, GMG_ERR_INVALID, "null vector");
    const int64_t n = h->N();
    const double *dx = h->eng.in_vec(x, n, memspace, h->st_b);
    double *dy = (memspace == GMG_MEM_DEVICE) ? y : h->st_x;
    h->sys_apply(dx, dy);
    h->eng.out_vec(y, dy, n, memspace);
  });
}

int gmg_block_fgmres_solve(gmg_block_handle_t h, const double *b, double *x, int memspace, int m0, int restart, int m_add,
                           int maxiter, double atol, double rtol, int use_precond, gmg_result *res, double *hist, int hist_cap)
{
  return guarded_b(h, [&] {
    check_block_ready(h);
    REQUIRE(b && x, GMG_ERR_INVALID, "null vector");
    REQUIRE(m0 >= 1 && m_add >= 1 && maxiter >= 0, GMG_ERR_INVALID, "bad FGMRES sizes");
    const int64_t n = h->N();
    const double *db = h->eng.in_vec(b, n, memspace, h->st_b);
    double *dx = (memspace == GMG_MEM_DEVICE) ? x : h->st_x;
    if (memspace == GMG_MEM_HOST) h->eng.h2d(dx, x, n);
    ConvLog log;
    log.configure(maxiter, atol, rtol);
    KrylovOps ops = h->ops(use_precond != 0);
    const double beta = fgmres_core(h->eng, n, n, db, dx, h->fg_V, h->fg_Z, ops, m0, restart != 0, m_add, log);
    h->eng.out_vec(x, dx, n, memspace);
    log.export_to(res, hist, hist_cap, beta);
  });
}

int gmg_block_cg_solve(gmg_block_handle_t h, const double *b, double *x, int memspace, int maxiter, double atol, double rtol,
                       int flexible, int use_precond, gmg_result *res, double *hist, int hist_cap)
{
  return guarded_b(h, [&] {
    check_block_ready(h);
    REQUIRE(b && x, GMG_ERR_INVALID, "null vector");
    REQUIRE(maxiter >= 0, GMG_ERR_INVALID, "maxiter < 0");
    const int64_t n = h->N();
    const double *db = h->eng.in_vec(b, n, memspace, h->st_b);
    double *dx = (memspace == GMG_MEM_DEVICE) ? x : h->st_x;
    if (memspace == GMG_MEM_HOST) h->eng.h2d(dx, x, n);
    ConvLog log;
    log.configure(maxiter, atol, rtol);
    KrylovOps ops = h->ops(use_precond != 0);
    const double resn = cg_core(h->eng, n, db, dx, h->kw, h->kp, h->kz, h->kr, ops, flexible != 0, log);
    h->eng.out_vec(x, dx, n, memspace);
    log.export_to(res, hist, hist_cap, resn);
  });
}

int gmg_block_diag_log(gmg_block_handle_t h, int i, gmg_result *res)
{
  return guarded_b(h, [&] {
    check_block_ready(h);
    check_block(h, i);
    REQUIRE(res, GMG_ERR_INVALID, "null result");
    BlockDiag &D = h->diag[i];
    const ConvLog *L = (D.kind == GMG_BLOCK_GMG) ? &D.g->log : &D.log;
    REQUIRE(!L->residuals.empty(), GMG_ERR_STATE, "this block solver keeps no convergence log");
    const size_t k = std::min<size_t>((size_t)L->num_iters, L->residuals.size() - 1);
    L->export_to(res, nullptr, 0, L->residuals[k]);
  });
}

} // extern "C"
