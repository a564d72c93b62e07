# GridapSolversAMD.jl -- Julia binding of libgmgamd.so (include/gmg_amd.h).
#
# Host code stays Julia: this module keeps the Gridap.Algebra surface
#     LinearSolver -> symbolic_setup -> numerical_setup / numerical_setup! -> solve!
# of GridapSolvers' GMGLinearSolver (src/LinearSolvers/GMGLinearSolvers.jl:48-69,164-210,
# 612-649), CGSolver (Krylov/CGSolvers.jl:19-120) and FGMRESSolver
# (Krylov/FGMRESSolvers.jl:26-199) and forwards the work to the MI355X through `ccall`.
#
# NOTE: Julia is not installed in the build image, so this file cannot be executed
# there.  It is deliberately thin and mechanical: every `ccall` below binds one symbol
# of include/gmg_amd.h with the argument order of that header (a CPU test checks that
# only declared symbols are referenced), and the same ABI is exercised end-to-end by
# the Python mirror (gridapsolvers.jl_amd/solvers.py) in the GPU test-suite.
#
# Usage (drop-in for the reference's test/LinearSolvers/GMGTests.jl:109-124):
#
#     using GridapSolversAMD
#     smatrices, A, b = compute_hierarchy_matrices(trials,tests,biform,liform,qdegree)
#     Ps  = explicit_prolongations(tests)          # SparseMatrixCSC per level (see INTEGRATION.md)
#     gmg = HipGMGLinearSolver(smatrices, Ps;
#             pre_smoothers  = Fill(RichardsonSmoother(JacobiLinearSolver(),10,2.0/3.0),nlev-1),
#             maxiter=1, mode=:preconditioner, cycle_type=:v_cycle)
#     solver = HipCGSolver(gmg; maxiter=20, atol=1e-14, rtol=1e-6)
#     ns = numerical_setup(symbolic_setup(solver,A),A)
#     x  = zeros(size(A,2)); solve!(x,ns,b)
module GridapSolversAMD

using LinearAlgebra
using SparseArrays
using BlockArrays: blocks
using Gridap
using Gridap.Algebra
using GridapSolvers
using GridapSolvers.SolverInterfaces: ConvergenceLog, SolverTolerances
using GridapSolvers.LinearSolvers: RichardsonSmoother, JacobiLinearSolver

export HipGMGLinearSolver, HipCGSolver, HipFGMRESSolver, PatchTable, HipPatchProlongation
export HipRichardsonLinearSolver, HipBlockTriangularSolver, HipBlockDiagonalSolver, HipBlockFGMRESSolver, block_mul!, block_cg_solve!

const libgmgamd = get(ENV, "LIBGMGAMD", joinpath(@__DIR__, "..", "libgmgamd.so"))

# enums of gmg_amd.h
const GMG_CSR, GMG_CSC = Cint(0), Cint(1)
const GMG_MEM_HOST, GMG_MEM_DEVICE = Cint(0), Cint(1)
const GMG_PRE, GMG_POST, GMG_PRE_AND_POST = Cint(0), Cint(1), Cint(2)
const GMG_PATCH_LU, GMG_PATCH_NOPIVOT = Cint(0), Cint(1)

struct GmgResult
  niters::Int32
  flag::Int32
  res0::Float64
  res::Float64
end

function check(h::Ptr{Cvoid}, status::Cint)
  if status != 0
    msg = unsafe_string(ccall((:gmg_last_error, libgmgamd), Cstring, (Ptr{Cvoid},), h))
    error("libgmgamd status $status: $msg")   # reference behaviour: @check / @assert -> exception
  end
  return nothing
end

# ---------------------------------------------------------------------------------
# smoother descriptions.  RichardsonSmoother / JacobiLinearSolver are the reference's
# own types; patch smoothers are passed as the dof tables that reach solve!
# (PatchSolvers.jl:279-300 patch_cols, BlockJacobiSolvers.jl:141-170).
# ---------------------------------------------------------------------------------
struct PatchTable
  patch_ptr  :: Vector{Int64}   # npatch+1, 1-based like Gridap's Table.ptrs
  patch_dofs :: Vector{Int64}   # 1-based dof ids
  pivoting   :: Bool            # true: PatchSolver (lu!) ; false: BlockJacobiSolver (NoPivot)
end
PatchTable(t::Gridap.Arrays.Table; pivoting=true) = PatchTable(Int64.(t.ptrs), Int64.(t.data), pivoting)

# Patch-corrected prolongation (PatchProlongationOperator, PatchTransferOperators.jl:153-172): the plain
# prolongation matrix plus the patch dof table; pass it in `interp[l]`.
struct HipPatchProlongation{M}
  P       :: M
  patches :: PatchTable
end

struct HipGMGLinearSolver{A,B,C,D,E,F} <: Gridap.Algebra.LinearSolver
  smatrices      :: A
  interp         :: B           # explicit sparse prolongations, level l+1 -> l
  restrict       :: C           # explicit sparse restrictions or nothing (=> P^T on the device)
  pre_smoothers  :: D
  post_smoothers :: E
  coarsest_solver :: F          # GMGLinearSolvers.jl:54: LUSolver() (dense inverse on the device), CGSolver(JacobiLinearSolver())
                                # (device CG), anything else runs on the host through a callback (set_coarsest_solver!)
  mode           :: Symbol
  cycle_type     :: Symbol
  log            :: ConvergenceLog{Float64}
  device         :: Int
  options        :: Dict{String,Float64}   # device-side policy handed to gmg_set_option before the operators are set (keys: include/gmg_amd.h)
  pin_vectors    :: Bool                   # page-lock the Vector{Float64}s solve! is called with, once (PETScCaches.jl:23-36 pins its x / b the same way)
end

# Same keyword surface as GMGLinearSolver(smatrices,interp,restrict;...) GMGLinearSolvers.jl:48-69, plus three keywords the
# reference has no use for: `device`, `options` (layout / schedule policy of THIS solver: e.g. Dict("pat_tile"=>2, "persist"=>0,
# "x0_zero"=>1); unknown keys are rejected by the library) and `pin_vectors` (default false; true: the vectors solve! sees are page-locked
# once and then move over PCIe by DMA at the link rate -- 3.6 ms instead of 4.7 ms per config-2 solve; the setup keeps them alive.
# For a top-level solve! that is called again and again with the SAME x and b; not for a preconditioner applied to rotating work vectors)
function HipGMGLinearSolver(
  smatrices::AbstractArray{<:AbstractMatrix}, interp::AbstractArray, restrict = nothing;
  pre_smoothers  = fill(RichardsonSmoother(JacobiLinearSolver(),10),length(smatrices)-1),
  post_smoothers = pre_smoothers,
  coarsest_solver = Gridap.Algebra.LUSolver(),
  mode = :preconditioner, cycle_type = :v_cycle,
  maxiter = 100, atol = 1.0e-14, rtol = 1.0e-08, verbose = false, device = 0,
  options = Dict{String,Float64}(), pin_vectors = false,
)
  nlev = length(smatrices)
  @assert nlev-1 == length(interp) == length(pre_smoothers) == length(post_smoothers)
  @assert isnothing(restrict) || length(restrict) == nlev-1
  @assert mode ∈ [:preconditioner,:solver]
  @assert cycle_type ∈ [:v_cycle,:w_cycle,:f_cycle]
  tols = SolverTolerances{Float64}(;maxiter=maxiter,atol=atol,rtol=rtol)
  log  = ConvergenceLog("GMG-MI355X",tols;verbose=verbose)
  opts = Dict{String,Float64}(string(k) => Float64(v) for (k,v) in pairs(options))
  return HipGMGLinearSolver(smatrices,interp,restrict,pre_smoothers,post_smoothers,coarsest_solver,mode,cycle_type,log,device,opts,pin_vectors)
end

struct HipGMGSymbolicSetup{A} <: Gridap.Algebra.SymbolicSetup
  solver :: A
end
Gridap.Algebra.symbolic_setup(s::HipGMGLinearSolver,::AbstractMatrix) = HipGMGSymbolicSetup(s)

mutable struct HipGMGNumericalSetup{A} <: Gridap.Algebra.NumericalSetup
  solver :: A
  handle :: Ptr{Cvoid}
  n      :: Int
  keepalive :: Any          # Julia objects the handle points at (coarse-solver callback context)
  pinned :: Vector{Tuple{Vector{Float64},Ptr{Float64},Int}}   # vectors registered with gmg_host_register (+ the pointer and byte count they
                                                              # were registered with): referenced here so that they outlive their registration
  pin_stats :: Vector{Int}                                    # [registrations, hits]: pinning switches itself off when it thrashes
  HipGMGNumericalSetup(solver::A, handle, n) where A = new{A}(solver, handle, n, nothing, Tuple{Vector{Float64},Ptr{Float64},Int}[], [0,0])
end

# gmg_set_option / gmg_get_option: per-handle layout and schedule policy (live options act at the next call, the others at the
# next gmg_setup -- call numerical_setup! or set them through the solver's `options` keyword instead)
function set_option!(ns::HipGMGNumericalSetup, key, value::Real)
  check(ns.handle, ccall((:gmg_set_option, libgmgamd), Cint, (Ptr{Cvoid},Cstring,Float64), ns.handle, string(key), Float64(value)))
  return ns
end
function get_option(ns::HipGMGNumericalSetup, key)
  v, src = Ref(0.0), Ref(Cint(0))
  check(ns.handle, ccall((:gmg_get_option, libgmgamd), Cint, (Ptr{Cvoid},Cstring,Ref{Float64},Ref{Cint}), ns.handle, string(key), v, src))
  return (isnan(v[]) ? nothing : v[], (:default, :handle, :environment)[src[]+1])
end

# gmg_set_stream: issue the handle's work on the caller's HIP stream (e.g. `AMDGPU.stream().stream` as a Ptr{Cvoid}): device vectors
# (HipDeviceVector) produced / consumed by the caller's kernels on that stream need no synchronisation around solve! / ldiv!.
# A null hipStream_t IS a stream -- HIP's default one, what AMDGPU.jl reports for its default stream -- so it is passed on as
# GMG_STREAM_LEGACY (hipStreamLegacy); `reset_stream!` returns to the handle's own stream (gmg_set_stream(h, NULL)).
const GMG_STREAM_LEGACY = Ptr{Cvoid}(UInt(1))
function set_stream!(ns::HipGMGNumericalSetup, stream::Ptr{Cvoid})
  s = stream == C_NULL ? GMG_STREAM_LEGACY : stream
  check(ns.handle, ccall((:gmg_set_stream, libgmgamd), Cint, (Ptr{Cvoid},Ptr{Cvoid}), ns.handle, s))
  return ns
end
function reset_stream!(ns::HipGMGNumericalSetup)
  check(ns.handle, ccall((:gmg_set_stream, libgmgamd), Cint, (Ptr{Cvoid},Ptr{Cvoid}), ns.handle, C_NULL))
  return ns
end
function get_stream(ns::HipGMGNumericalSetup)
  s = Ref{Ptr{Cvoid}}(C_NULL)
  check(ns.handle, ccall((:gmg_get_stream, libgmgamd), Cint, (Ptr{Cvoid},Ref{Ptr{Cvoid}}), ns.handle, s))
  return s[]
end

# Page-lock the vectors of a solve once (gmg_host_register); at most 8 are held, the oldest is released first.  The setup holds a
# reference, and every entry remembers the (pointer, bytes) it was registered with: a vector that was resize!d / push!ed since is
# released and registered again.  A host-language Krylov loop that rotates through more work vectors than the table holds
# (FGMRES: solve!(Z[j], Pr, V[j])) would re-register on every call -- slower than the pageable path --, so pinning switches itself off
# for this setup once registrations outnumber hits 4 : 1 after the first 32.
function _unpin_entry(ns, e)
  st = ccall((:gmg_host_unregister, libgmgamd), Cint, (Ptr{Cvoid},Ptr{Cvoid}), ns.handle, e[2])
  return st          # (GMG_ERR_INVALID: the range was released already -- nothing to undo)
end
function pin!(ns::HipGMGNumericalSetup, vs::Vector{Float64}...)
  for v in vs
    isempty(v) && continue
    i = findfirst(e -> e[1] === v, ns.pinned)
    if !isnothing(i)
      e = ns.pinned[i]
      if e[2] == pointer(v) && e[3] == sizeof(v)
        ns.pin_stats[2] += 1
        continue
      end
      _unpin_entry(ns, e)                                  # the array was reallocated: the old range is stale
      deleteat!(ns.pinned, i)
    end
    if ns.pin_stats[1] >= 32 && ns.pin_stats[1] > 4*ns.pin_stats[2]
      continue                                             # thrashing: leave the vectors pageable
    end
    if length(ns.pinned) >= 8
      _unpin_entry(ns, popfirst!(ns.pinned))
    end
    GC.@preserve v check(ns.handle, ccall((:gmg_host_register, libgmgamd), Cint, (Ptr{Cvoid},Ptr{Cvoid},Int64), ns.handle, pointer(v), sizeof(v)))
    push!(ns.pinned, (v, pointer(v), sizeof(v)))
    ns.pin_stats[1] += 1
  end
  return ns
end
_maybe_pin!(ns::HipGMGNumericalSetup, vs...) = ns.solver.pin_vectors ? pin!(ns, vs...) : ns

# Vectors that already live on the device (e.g. AMDGPU.ROCArray): wrap their pointer; solve! then passes GMG_MEM_DEVICE and nothing
# crosses PCIe.  `owner` keeps the array alive.
struct HipDeviceVector
  ptr   :: Ptr{Float64}
  n     :: Int
  owner :: Any
end

# --- operator upload: SparseMatrixCSC{Float64,Ti} is CSC / 1-based / sizeof(Ti) bytes ----
function _set_op(sym::Symbol, h, lev, M::SparseMatrixCSC{Float64,Ti}) where Ti
  nb = Cint(sizeof(Ti))
  GC.@preserve M begin
    if sym === :matrix
      st = ccall((:gmg_set_matrix, libgmgamd), Cint,
        (Ptr{Cvoid},Cint,Int64,Int64,Int64,Ptr{Cvoid},Ptr{Cvoid},Ptr{Float64},Cint,Cint,Cint),
        h, lev, size(M,1), size(M,2), nnz(M), M.colptr, M.rowval, M.nzval, GMG_CSC, 1, nb)
    elseif sym === :prolongation
      st = ccall((:gmg_set_prolongation, libgmgamd), Cint,
        (Ptr{Cvoid},Cint,Int64,Int64,Int64,Ptr{Cvoid},Ptr{Cvoid},Ptr{Float64},Cint,Cint,Cint),
        h, lev, size(M,1), size(M,2), nnz(M), M.colptr, M.rowval, M.nzval, GMG_CSC, 1, nb)
    else
      st = ccall((:gmg_set_restriction, libgmgamd), Cint,
        (Ptr{Cvoid},Cint,Int64,Int64,Int64,Ptr{Cvoid},Ptr{Cvoid},Ptr{Float64},Cint,Cint,Cint),
        h, lev, size(M,1), size(M,2), nnz(M), M.colptr, M.rowval, M.nzval, GMG_CSC, 1, nb)
    end
  end
  check(h, st)
end
# SparseMatricesCSR.SparseMatrixCSR{Bi,Float64,Ti}: fields rowptr, colval, nzval (PAExtras.jl:147-159)
function _set_op(sym::Symbol, h, lev, M::AbstractMatrix)
  if hasproperty(M,:rowptr) && hasproperty(M,:colval)
    Ti = eltype(M.rowptr); Bi = Int(first(M.rowptr))          # index base = first row pointer
    f = sym === :matrix ? :gmg_set_matrix : (sym === :prolongation ? :gmg_set_prolongation : :gmg_set_restriction)
    GC.@preserve M begin
      st = if f === :gmg_set_matrix
        ccall((:gmg_set_matrix, libgmgamd), Cint,
          (Ptr{Cvoid},Cint,Int64,Int64,Int64,Ptr{Cvoid},Ptr{Cvoid},Ptr{Float64},Cint,Cint,Cint),
          h, lev, size(M,1), size(M,2), length(M.nzval), M.rowptr, M.colval, M.nzval, GMG_CSR, Bi, sizeof(Ti))
      elseif f === :gmg_set_prolongation
        ccall((:gmg_set_prolongation, libgmgamd), Cint,
          (Ptr{Cvoid},Cint,Int64,Int64,Int64,Ptr{Cvoid},Ptr{Cvoid},Ptr{Float64},Cint,Cint,Cint),
          h, lev, size(M,1), size(M,2), length(M.nzval), M.rowptr, M.colval, M.nzval, GMG_CSR, Bi, sizeof(Ti))
      else
        ccall((:gmg_set_restriction, libgmgamd), Cint,
          (Ptr{Cvoid},Cint,Int64,Int64,Int64,Ptr{Cvoid},Ptr{Cvoid},Ptr{Float64},Cint,Cint,Cint),
          h, lev, size(M,1), size(M,2), length(M.nzval), M.rowptr, M.colval, M.nzval, GMG_CSR, Bi, sizeof(Ti))
      end
    end
    check(h, st)
  else
    _set_op(sym, h, lev, SparseMatrixCSC{Float64,Int64}(sparse(M)))
  end
end

function _set_smoother(h, lev, which, sm::RichardsonSmoother)
  M = sm.M
  if M isa JacobiLinearSolver
    check(h, ccall((:gmg_set_smoother_jacobi, libgmgamd), Cint, (Ptr{Cvoid},Cint,Cint,Cint,Float64),
                   h, lev, which, sm.niter, sm.ω))
  elseif M isa PatchTable
    GC.@preserve M begin
      check(h, ccall((:gmg_set_smoother_patch, libgmgamd), Cint,
        (Ptr{Cvoid},Cint,Cint,Cint,Float64,Cint,Int64,Ptr{Cvoid},Ptr{Cvoid},Cint,Cint),
        h, lev, which, sm.niter, sm.ω, M.pivoting ? GMG_PATCH_LU : GMG_PATCH_NOPIVOT,
        length(M.patch_ptr)-1, M.patch_ptr, M.patch_dofs, 1, 8))
    end
  else
    error("HipGMGLinearSolver: smoother inner solver $(typeof(M)) is not available on the device")
  end
end

# numerical_setup(ss,A): GMGLinearSolvers.jl:183-210
function Gridap.Algebra.numerical_setup(ss::HipGMGSymbolicSetup, mat::AbstractMatrix)
  s = ss.solver
  nlev = length(s.smatrices)
  href = Ref{Ptr{Cvoid}}(C_NULL)
  check(C_NULL, ccall((:gmg_create, libgmgamd), Cint, (Ref{Ptr{Cvoid}},Cint,Cint), href, nlev, s.device))
  h = href[]
  ns = HipGMGNumericalSetup(s, h, size(mat,1))
  finalizer(x -> (x.handle != C_NULL && ccall((:gmg_destroy, libgmgamd), Cint, (Ptr{Cvoid},), x.handle); x.handle = C_NULL; empty!(x.pinned)), ns)
  for (k,v) in s.options                                          # policy first: some options act while the operators are handed over
    set_option!(ns, k, v)
  end
  for l in 1:nlev
    _set_op(:matrix, h, l-1, l == 1 ? mat : s.smatrices[l])       # smatrices[1] = mat (:338)
  end
  for l in 1:nlev-1
    ip = s.interp[l]
    if ip isa HipPatchProlongation
      _set_op(:prolongation, h, l-1, ip.P)
      T = ip.patches
      GC.@preserve T begin
        check(h, ccall((:gmg_set_prolongation_patch_correction, libgmgamd), Cint,
          (Ptr{Cvoid},Cint,Cint,Int64,Ptr{Cvoid},Ptr{Cvoid},Cint,Cint),
          h, l-1, T.pivoting ? GMG_PATCH_LU : GMG_PATCH_NOPIVOT, length(T.patch_ptr)-1, T.patch_ptr, T.patch_dofs, 1, 8))
      end
    else
      _set_op(:prolongation, h, l-1, ip)
    end
    !isnothing(s.restrict) && _set_op(:restriction, h, l-1, s.restrict[l])
    if s.post_smoothers[l] === s.pre_smoothers[l]
      _set_smoother(h, l-1, GMG_PRE_AND_POST, s.pre_smoothers[l])
    else
      _set_smoother(h, l-1, GMG_PRE,  s.pre_smoothers[l])
      _set_smoother(h, l-1, GMG_POST, s.post_smoothers[l])
    end
  end
  tols = s.log.tols
  mode  = s.mode == :preconditioner ? 0 : 1
  cycle = s.cycle_type == :v_cycle ? 0 : (s.cycle_type == :w_cycle ? 1 : 2)
  check(h, ccall((:gmg_set_options, libgmgamd), Cint, (Ptr{Cvoid},Cint,Cint,Cint,Float64,Float64),
                 h, mode, cycle, tols.maxiter, tols.atol, tols.rtol))
  if s.coarsest_solver isa Gridap.Algebra.LUSolver          # the default: one setup, dense inverse of the coarsest matrix on the device
    check(h, ccall((:gmg_setup, libgmgamd), Cint, (Ptr{Cvoid},), h))
  else                                                      # GMGLinearSolvers.jl:423-434: the caller's coarsest solver (sets up the handle itself)
    set_coarsest_solver!(ns, s.coarsest_solver, s.smatrices[nlev])
  end
  return ns
end

# numerical_setup!(ns,A[,smatrices]): new values on the same pattern.  The reference's FromMatrices variant only logs an @error
# (GMGLinearSolvers.jl:249-258); its weak-form variant (:260-297) re-assembles EVERY level and recomputes the smoothers and the
# coarsest solver: pass the re-assembled level matrices as `smatrices` (entry 1 is ignored, `mat` is level 1 as in :338) and
# every level is refreshed -- gmg_setup rebuilds D^-1, the patch blocks and the coarse inverse from the new values.  CSC values go
# over as they are (gmg_update_values_csc: the library scatters them into its row order; no sparse(transpose(A)) here).
function _update_values(h, lev, M::SparseMatrixCSC{Float64,Ti}) where Ti
  GC.@preserve M begin
    check(h, ccall((:gmg_update_values_csc, libgmgamd), Cint, (Ptr{Cvoid},Cint,Ptr{Cvoid},Ptr{Cvoid},Ptr{Float64},Cint,Cint),
                   h, lev, M.colptr, M.rowval, M.nzval, 1, sizeof(Ti)))
  end
end
function _update_values(h, lev, M::AbstractMatrix)
  if hasproperty(M,:rowptr) && hasproperty(M,:colval)       # SparseMatrixCSR: already the handle's order
    GC.@preserve M begin
      check(h, ccall((:gmg_update_values, libgmgamd), Cint, (Ptr{Cvoid},Cint,Ptr{Float64}), h, lev, M.nzval))
    end
  else
    _update_values(h, lev, SparseMatrixCSC{Float64,Int64}(sparse(M)))
  end
end
function Gridap.Algebra.numerical_setup!(ns::HipGMGNumericalSetup, mat::AbstractMatrix, smatrices = nothing)
  _update_values(ns.handle, 0, mat)
  if !isnothing(smatrices)
    @assert length(smatrices) == length(ns.solver.smatrices)
    for l in 2:length(smatrices)
      _update_values(ns.handle, l-1, smatrices[l])
    end
  end
  if ns.solver.coarsest_solver isa Gridap.Algebra.LUSolver
    check(ns.handle, ccall((:gmg_setup, libgmgamd), Cint, (Ptr{Cvoid},), ns.handle))
  else                                                      # the caller's coarsest solver is set up again on the new coarsest matrix (:291-294)
    nlev = length(ns.solver.smatrices)
    set_coarsest_solver!(ns, ns.solver.coarsest_solver, isnothing(smatrices) ? ns.solver.smatrices[nlev] : smatrices[nlev])
  end
  return ns
end

function _fill_log!(log::ConvergenceLog, res::GmgResult, hist::Vector{Float64})
  log.num_iters = res.niters
  fill!(log.residuals, 0.0)
  log.residuals[1:res.niters+1] .= hist[1:res.niters+1]
  return log
end

# solve!(x,ns,b): GMGLinearSolvers.jl:612-645 ; b untouched, x overwritten (zeroed first in
# :preconditioner mode), returns x.
function Gridap.Algebra.solve!(x::Vector{Float64}, ns::HipGMGNumericalSetup, b::Vector{Float64})
  @assert length(x) == length(b) == ns.n
  log  = ns.solver.log
  res  = Ref(GmgResult(0,0,0.0,0.0))
  hist = zeros(log.tols.maxiter+1)
  _maybe_pin!(ns, x, b)
  GC.@preserve x b hist begin
    check(ns.handle, ccall((:gmg_apply, libgmgamd), Cint,
      (Ptr{Cvoid},Ptr{Float64},Ptr{Float64},Cint,Ref{GmgResult},Ptr{Float64},Cint),
      ns.handle, b, x, GMG_MEM_HOST, res, hist, length(hist)))
  end
  _fill_log!(log, res[], hist)
  return x
end
# the same on device-resident vectors: one V-cycle per call costs no PCIe traffic (the way to use the library as the preconditioner
# of a Krylov loop that itself runs on the GPU)
function Gridap.Algebra.solve!(x::HipDeviceVector, ns::HipGMGNumericalSetup, b::HipDeviceVector)
  @assert x.n == b.n == ns.n
  log  = ns.solver.log
  res  = Ref(GmgResult(0,0,0.0,0.0))
  hist = zeros(log.tols.maxiter+1)
  GC.@preserve x b hist begin
    check(ns.handle, ccall((:gmg_apply, libgmgamd), Cint,
      (Ptr{Cvoid},Ptr{Float64},Ptr{Float64},Cint,Ref{GmgResult},Ptr{Float64},Cint),
      ns.handle, b.ptr, x.ptr, GMG_MEM_DEVICE, res, hist, length(hist)))
  end
  _fill_log!(log, res[], hist)
  return x
end
LinearAlgebra.ldiv!(x::AbstractVector, ns::HipGMGNumericalSetup, b::AbstractVector) = solve!(x,ns,b)   # :647-649

# duck-typed `mul!` for the transfer operators of a set-up hierarchy (GMGLinearSolvers.jl:484,491)
struct HipLevelOperator
  ns  :: HipGMGNumericalSetup
  lev :: Int      # 1-based
  op  :: Cint     # 0 = A, 1 = P, 2 = R
end
function LinearAlgebra.mul!(y::Vector{Float64}, o::HipLevelOperator, x::Vector{Float64})
  GC.@preserve x y begin
    check(o.ns.handle, ccall((:gmg_op_apply, libgmgamd), Cint, (Ptr{Cvoid},Cint,Cint,Ptr{Float64},Ptr{Float64},Cint),
                             o.ns.handle, o.lev-1, o.op, x, y, GMG_MEM_HOST))
  end
  return y
end

# ---------------------------------------------------------------------------------
# Outer Krylov solvers that keep the whole iteration on the device
# ---------------------------------------------------------------------------------
struct HipCGSolver{A} <: Gridap.Algebra.LinearSolver
  Pl       :: A
  log      :: ConvergenceLog{Float64}
  flexible :: Bool
end
function HipCGSolver(Pl::HipGMGLinearSolver; maxiter=1000, atol=1e-12, rtol=1.e-6, flexible=false, verbose=0, name="CG-MI355X")
  tols = SolverTolerances{Float64}(;maxiter=maxiter,atol=atol,rtol=rtol)     # CGSolvers.jl:19-23
  return HipCGSolver(Pl, ConvergenceLog(name,tols;verbose=verbose), flexible)
end

struct HipFGMRESSolver{A} <: Gridap.Algebra.LinearSolver
  m       :: Int
  restart :: Bool
  m_add   :: Int
  Pr      :: A
  log     :: ConvergenceLog{Float64}
end
function HipFGMRESSolver(m, Pr::HipGMGLinearSolver; restart=false, m_add=1, maxiter=100, atol=1e-12, rtol=1.e-6, verbose=false, name="FGMRES-MI355X")
  tols = SolverTolerances{Float64}(maxiter=maxiter,atol=atol,rtol=rtol)      # FGMRESSolvers.jl:26-30
  return HipFGMRESSolver(m, restart, m_add, Pr, ConvergenceLog(name,tols,verbose=verbose))
end

struct HipKrylovSymbolicSetup{A} <: Gridap.Algebra.SymbolicSetup
  solver :: A
end
Gridap.Algebra.symbolic_setup(s::Union{HipCGSolver,HipFGMRESSolver}, ::AbstractMatrix) = HipKrylovSymbolicSetup(s)

mutable struct HipKrylovNumericalSetup{A,B} <: Gridap.Algebra.NumericalSetup
  solver :: A
  P_ns   :: B
end
function Gridap.Algebra.numerical_setup(ss::HipKrylovSymbolicSetup, A::AbstractMatrix)
  P = ss.solver isa HipFGMRESSolver ? ss.solver.Pr : ss.solver.Pl
  P_ns = numerical_setup(symbolic_setup(P,A),A)                              # CGSolvers.jl:52
  return HipKrylovNumericalSetup(ss.solver, P_ns)
end
function Gridap.Algebra.numerical_setup!(ns::HipKrylovNumericalSetup, A::AbstractMatrix)
  numerical_setup!(ns.P_ns, A)
  return ns
end

function Gridap.Algebra.solve!(x::Vector{Float64}, ns::HipKrylovNumericalSetup{<:HipCGSolver}, b::Vector{Float64})
  s, h = ns.solver, ns.P_ns.handle
  tols = s.log.tols
  res  = Ref(GmgResult(0,0,0.0,0.0))
  hist = zeros(tols.maxiter+1)
  _maybe_pin!(ns.P_ns, x, b)
  GC.@preserve x b hist begin
    check(h, ccall((:gmg_cg_solve, libgmgamd), Cint,
      (Ptr{Cvoid},Ptr{Float64},Ptr{Float64},Cint,Cint,Float64,Float64,Cint,Cint,Ref{GmgResult},Ptr{Float64},Cint),
      h, b, x, GMG_MEM_HOST, tols.maxiter, tols.atol, tols.rtol, s.flexible ? 1 : 0, 1, res, hist, length(hist)))
  end
  _fill_log!(s.log, res[], hist)
  return x
end

function Gridap.Algebra.solve!(x::Vector{Float64}, ns::HipKrylovNumericalSetup{<:HipFGMRESSolver}, b::Vector{Float64})
  s, h = ns.solver, ns.P_ns.handle
  tols = s.log.tols
  res  = Ref(GmgResult(0,0,0.0,0.0))
  hist = zeros(tols.maxiter+1)
  _maybe_pin!(ns.P_ns, x, b)
  GC.@preserve x b hist begin
    check(h, ccall((:gmg_fgmres_solve, libgmgamd), Cint,
      (Ptr{Cvoid},Ptr{Float64},Ptr{Float64},Cint,Cint,Cint,Cint,Cint,Float64,Float64,Cint,Ref{GmgResult},Ptr{Float64},Cint),
      h, b, x, GMG_MEM_HOST, s.m, s.restart ? 1 : 0, s.m_add, tols.maxiter, tols.atol, tols.rtol, 1, res, hist, length(hist)))
  end
  _fill_log!(s.log, res[], hist)
  return x
end

# RichardsonLinearSolver(omega,maxiter;Pl=gmg) on the device: RichardsonLinearSolvers.jl:79-106
struct HipRichardsonLinearSolver{A} <: Gridap.Algebra.LinearSolver
  omega :: Float64
  Pl    :: A
  log   :: ConvergenceLog{Float64}
end
function HipRichardsonLinearSolver(omega::Real, maxiter::Integer, Pl::HipGMGLinearSolver; rtol=1e-10, atol=1e-6, verbose=true, name="Richardson-MI355X")
  tols = SolverTolerances{Float64}(maxiter=maxiter,atol=atol,rtol=rtol)
  return HipRichardsonLinearSolver(Float64(omega), Pl, ConvergenceLog(name,tols;verbose=verbose))
end
Gridap.Algebra.symbolic_setup(s::HipRichardsonLinearSolver, ::AbstractMatrix) = HipKrylovSymbolicSetup(s)
function Gridap.Algebra.solve!(x::Vector{Float64}, ns::HipKrylovNumericalSetup{<:HipRichardsonLinearSolver}, b::Vector{Float64})
  s, h = ns.solver, ns.P_ns.handle
  tols = s.log.tols
  res  = Ref(GmgResult(0,0,0.0,0.0))
  hist = zeros(tols.maxiter+1)
  _maybe_pin!(ns.P_ns, x, b)
  GC.@preserve x b hist begin
    check(h, ccall((:gmg_richardson_solve, libgmgamd), Cint,
      (Ptr{Cvoid},Ptr{Float64},Ptr{Float64},Cint,Float64,Cint,Float64,Float64,Cint,Ref{GmgResult},Ptr{Float64},Cint),
      h, b, x, GMG_MEM_HOST, s.omega, tols.maxiter, tols.atol, tols.rtol, 1, res, hist, length(hist)))
  end
  _fill_log!(s.log, res[], hist)
  return x
end

# ------------------------------------------------------------------------------------------------
# Block preconditioners on the device (include/gmg_amd.h, "block preconditioners"):
# BlockDiagonalSolver / BlockTriangularSolver (BlockSolvers/BlockTriangularSolvers.jl:55-85,186-242)
# with the outer FGMRES on the BlockArrays system, the solver shape of test/Applications/StokesGMG.jl:142-153:
#
#     P      = HipBlockTriangularSolver([solver_u, solver_p]; mats=Dict((2,2)=>Mp), coeffs=[1.0 1.0; 0.0 1.0], half=:upper)
#     solver = HipBlockFGMRESSolver(20, P; atol=1e-10, rtol=1e-12)
#     ns = numerical_setup(symbolic_setup(solver,A),A); solve!(x,ns,b)       # A::BlockMatrix, x,b::BlockVector
#
# solvers[i]: HipGMGLinearSolver | CGSolver(JacobiLinearSolver()) | LUSolver() | JacobiLinearSolver().
# `mats[(i,j)]` replaces the system block in the preconditioner (MatrixBlock / an assembled BiformBlock).
# ------------------------------------------------------------------------------------------------
const GMG_BLOCK_DIAGONAL, GMG_BLOCK_LOWER, GMG_BLOCK_UPPER = Cint(0), Cint(1), Cint(2)
const GMG_BLOCK_GMG, GMG_BLOCK_CG_JACOBI, GMG_BLOCK_LU, GMG_BLOCK_JACOBI = Cint(1), Cint(2), Cint(3), Cint(4)

function check_block(h::Ptr{Cvoid}, status::Cint)
  if status != 0
    msg = unsafe_string(ccall((:gmg_block_last_error, libgmgamd), Cstring, (Ptr{Cvoid},), h))
    error("libgmgamd status $status: $msg")
  end
  return nothing
end

struct HipBlockTriangularSolver <: Gridap.Algebra.LinearSolver
  solvers :: Vector{Any}
  mats    :: Dict{Tuple{Int,Int},Any}
  coeffs  :: Matrix{Float64}
  half    :: Symbol                      # :upper | :lower | :diagonal (BlockDiagonalSolver)
end
function HipBlockTriangularSolver(solvers::AbstractVector; mats=Dict{Tuple{Int,Int},Any}(),
                                  coeffs=fill(1.0,length(solvers),length(solvers)), half=:upper)
  @assert half in (:upper,:lower,:diagonal)                                   # BlockTriangularSolvers.jl:65
  @assert size(coeffs) == (length(solvers),length(solvers))
  HipBlockTriangularSolver(collect(Any,solvers), Dict{Tuple{Int,Int},Any}(mats), Matrix{Float64}(coeffs), half)
end
HipBlockDiagonalSolver(solvers::AbstractVector; mats=Dict{Tuple{Int,Int},Any}()) =
  HipBlockTriangularSolver(solvers; mats=mats, half=:diagonal)

struct HipBlockFGMRESSolver <: Gridap.Algebra.LinearSolver
  m :: Int; restart :: Bool; m_add :: Int
  Pr :: HipBlockTriangularSolver
  log :: ConvergenceLog{Float64}
end
HipBlockFGMRESSolver(m, Pr::HipBlockTriangularSolver; restart=false, m_add=1, maxiter=100, atol=1e-12, rtol=1.e-6,
                     verbose=false, name="FGMRES-MI355X") =
  HipBlockFGMRESSolver(m, restart, m_add, Pr, ConvergenceLog(name, SolverTolerances{Float64}(maxiter=maxiter,atol=atol,rtol=rtol); verbose=verbose))

struct HipBlockSymbolicSetup{A} <: Gridap.Algebra.SymbolicSetup
  solver :: A
end
Gridap.Algebra.symbolic_setup(s::Union{HipBlockTriangularSolver,HipBlockFGMRESSolver}, ::AbstractMatrix) = HipBlockSymbolicSetup(s)

mutable struct HipBlockNumericalSetup{A} <: Gridap.Algebra.NumericalSetup
  solver   :: A
  handle   :: Ptr{Cvoid}
  block_ns :: Vector{Any}                # the GMG numerical setups the handle borrows
  n        :: Int
end

function _set_block(sym::Symbol, h, i, j, M::AbstractMatrix)
  C = M isa SparseMatrixCSC{Float64} ? M : SparseMatrixCSC{Float64,Int64}(sparse(M))
  nb = Cint(sizeof(eltype(C.colptr)))
  GC.@preserve C begin
    st = if sym === :system
      ccall((:gmg_block_set_system_block, libgmgamd), Cint,
        (Ptr{Cvoid},Cint,Cint,Int64,Int64,Int64,Ptr{Cvoid},Ptr{Cvoid},Ptr{Float64},Cint,Cint,Cint),
        h, i, j, size(C,1), size(C,2), nnz(C), C.colptr, C.rowval, C.nzval, GMG_CSC, 1, nb)
    elseif sym === :precond
      ccall((:gmg_block_set_precond_block, libgmgamd), Cint,
        (Ptr{Cvoid},Cint,Cint,Int64,Int64,Int64,Ptr{Cvoid},Ptr{Cvoid},Ptr{Float64},Cint,Cint,Cint),
        h, i, j, size(C,1), size(C,2), nnz(C), C.colptr, C.rowval, C.nzval, GMG_CSC, 1, nb)
    else
      ccall((:gmg_block_set_diag_matrix, libgmgamd), Cint,
        (Ptr{Cvoid},Cint,Int64,Int64,Ptr{Cvoid},Ptr{Cvoid},Ptr{Float64},Cint,Cint,Cint),
        h, i, size(C,1), nnz(C), C.colptr, C.rowval, C.nzval, GMG_CSC, 1, nb)
    end
  end
  check_block(h, st)
end

# `mat` is a BlockArrays.BlockMatrix: blocks(mat)[i,j] (BlockTriangularSolvers.jl:132-152)
function _block_numerical_setup(P::HipBlockTriangularSolver, mat, owner)
  B  = blocks(mat); NB = length(P.solvers)
  sizes = Int64[size(B[i,i],1) for i in 1:NB]
  kind = P.half === :upper ? GMG_BLOCK_UPPER : (P.half === :lower ? GMG_BLOCK_LOWER : GMG_BLOCK_DIAGONAL)
  href = Ref{Ptr{Cvoid}}(C_NULL)
  check_block(C_NULL, ccall((:gmg_block_create, libgmgamd), Cint, (Ref{Ptr{Cvoid}},Cint,Ptr{Int64},Cint,Cint), href, NB, sizes, kind, 0))
  ns = HipBlockNumericalSetup(owner, href[], Any[], sum(sizes))
  # Finalizers of objects that die together run in no particular order; the library copes with either order
  # (gmg_destroy on a borrowed handle makes the block handle forget it, gmg_block_destroy skips forgotten handles).
  finalizer(x -> ccall((:gmg_block_destroy, libgmgamd), Cint, (Ptr{Cvoid},), x.handle), ns)
  h = ns.handle
  for i in 1:NB, j in 1:NB
    iszero(nnz(sparse(B[i,j]))) || _set_block(:system, h, i-1, j-1, B[i,j])
    if i != j
      haskey(P.mats,(i,j)) && _set_block(:precond, h, i-1, j-1, P.mats[(i,j)])
      check_block(h, ccall((:gmg_block_set_coeff, libgmgamd), Cint, (Ptr{Cvoid},Cint,Cint,Float64), h, i-1, j-1, P.coeffs[i,j]))
    end
  end
  for (i,s) in enumerate(P.solvers)
    Mi = get(P.mats, (i,i), B[i,i])
    if s isa HipGMGLinearSolver
      g = numerical_setup(symbolic_setup(s,Mi),Mi); push!(ns.block_ns, g)
      check_block(h, ccall((:gmg_block_set_diag_gmg, libgmgamd), Cint, (Ptr{Cvoid},Cint,Ptr{Cvoid}), h, i-1, g.handle))
      continue
    end
    kind_i, maxiter, atol, rtol = if s isa GridapSolvers.LinearSolvers.CGSolver
      @assert s.Pl isa JacobiLinearSolver && !s.flexible
      GMG_BLOCK_CG_JACOBI, s.log.tols.maxiter, s.log.tols.atol, s.log.tols.rtol
    elseif s isa Gridap.Algebra.LUSolver
      GMG_BLOCK_LU, 0, 0.0, 0.0
    else
      @assert s isa JacobiLinearSolver
      GMG_BLOCK_JACOBI, 0, 0.0, 0.0
    end
    check_block(h, ccall((:gmg_block_set_diag_solver, libgmgamd), Cint, (Ptr{Cvoid},Cint,Cint,Cint,Float64,Float64), h, i-1, kind_i, maxiter, atol, rtol))
    haskey(P.mats,(i,i)) && _set_block(:diag, h, i-1, 0, Mi)
  end
  check_block(h, ccall((:gmg_block_setup, libgmgamd), Cint, (Ptr{Cvoid},), h))
  return ns
end
Gridap.Algebra.numerical_setup(ss::HipBlockSymbolicSetup{HipBlockTriangularSolver}, mat::AbstractMatrix) = _block_numerical_setup(ss.solver, mat, ss.solver)
Gridap.Algebra.numerical_setup(ss::HipBlockSymbolicSetup{HipBlockFGMRESSolver}, mat::AbstractMatrix) = _block_numerical_setup(ss.solver.Pr, mat, ss.solver)

# block vectors are passed as their contiguous parent (BlockArrays stores a BlockVector's blocks back to back)
_flat(v::AbstractVector) = v isa Vector{Float64} ? v : parent(v)

function _fill_block_logs!(ns::HipBlockNumericalSetup, P::HipBlockTriangularSolver)
  for (i,s) in enumerate(P.solvers)
    (s isa HipGMGLinearSolver || s isa GridapSolvers.LinearSolvers.CGSolver) || continue
    res = Ref(GmgResult(0,0,0.0,0.0))
    check_block(ns.handle, ccall((:gmg_block_diag_log, libgmgamd), Cint, (Ptr{Cvoid},Cint,Ref{GmgResult}), ns.handle, i-1, res))
    s.log.num_iters = res[].niters
  end
end

# solve!(x,ns::BlockTriangularSolverNS,b)
function Gridap.Algebra.solve!(x::AbstractVector, ns::HipBlockNumericalSetup{HipBlockTriangularSolver}, b::AbstractVector)
  xf, bf = _flat(x), _flat(b)
  GC.@preserve xf bf check_block(ns.handle, ccall((:gmg_block_precond_apply, libgmgamd), Cint,
    (Ptr{Cvoid},Ptr{Float64},Ptr{Float64},Cint), ns.handle, bf, xf, GMG_MEM_HOST))
  _fill_block_logs!(ns, ns.solver)
  return x
end
# mul!(y,A,x) on the block system held by the handle
function block_mul!(y::AbstractVector, ns::HipBlockNumericalSetup, x::AbstractVector)
  yf, xf = _flat(y), _flat(x)
  GC.@preserve yf xf check_block(ns.handle, ccall((:gmg_block_apply_system, libgmgamd), Cint,
    (Ptr{Cvoid},Ptr{Float64},Ptr{Float64},Cint), ns.handle, xf, yf, GMG_MEM_HOST))
  return y
end
# solve!(x,ns::FGMRESNumericalSetup,b) with Pr = the block preconditioner
function Gridap.Algebra.solve!(x::AbstractVector, ns::HipBlockNumericalSetup{HipBlockFGMRESSolver}, b::AbstractVector)
  s = ns.solver; tols = s.log.tols
  res  = Ref(GmgResult(0,0,0.0,0.0)); hist = zeros(tols.maxiter+1)
  xf, bf = _flat(x), _flat(b)
  GC.@preserve xf bf hist check_block(ns.handle, ccall((:gmg_block_fgmres_solve, libgmgamd), Cint,
    (Ptr{Cvoid},Ptr{Float64},Ptr{Float64},Cint,Cint,Cint,Cint,Cint,Float64,Float64,Cint,Ref{GmgResult},Ptr{Float64},Cint),
    ns.handle, bf, xf, GMG_MEM_HOST, s.m, s.restart ? 1 : 0, s.m_add, tols.maxiter, tols.atol, tols.rtol, 1, res, hist, length(hist)))
  _fill_log!(s.log, res[], hist)
  _fill_block_logs!(ns, s.Pr)
  return x
end
# CGSolver(P) on an SPD block system
function block_cg_solve!(x::AbstractVector, ns::HipBlockNumericalSetup, b::AbstractVector, log::ConvergenceLog; flexible=false)
  tols = log.tols
  res  = Ref(GmgResult(0,0,0.0,0.0)); hist = zeros(tols.maxiter+1)
  xf, bf = _flat(x), _flat(b)
  GC.@preserve xf bf hist check_block(ns.handle, ccall((:gmg_block_cg_solve, libgmgamd), Cint,
    (Ptr{Cvoid},Ptr{Float64},Ptr{Float64},Cint,Cint,Float64,Float64,Cint,Cint,Ref{GmgResult},Ptr{Float64},Cint),
    ns.handle, bf, xf, GMG_MEM_HOST, tols.maxiter, tols.atol, tols.rtol, flexible ? 1 : 0, 1, res, hist, length(hist)))
  _fill_log!(log, res[], hist)
  return x
end

# ------------------------------------------------------------------------------------------------
# Options the reference passes as keyword arguments / solver objects
# ------------------------------------------------------------------------------------------------
const GMG_COARSE_DENSE_INVERSE, GMG_COARSE_CG_JACOBI, GMG_COARSE_HOST_CALLBACK = Cint(0), Cint(1), Cint(2)

# coarsest_solver (GMGLinearSolvers.jl:54,423-434).  `LUSolver()` -> dense inverse on the device (default);
# `CGSolver(JacobiLinearSolver();maxiter,atol,rtol)` -> the same iteration on the device; any other Gridap LinearSolver
# (PETScLinearSolver, PardisoSolver, ...) -> solved by Julia itself through a callback on host vectors.
mutable struct CoarseCallback
  ns   :: Any                 # numerical setup of the caller's solver on the coarsest matrix
  x    :: Vector{Float64}
  b    :: Vector{Float64}
end
function _coarse_trampoline(ctx::Ptr{Cvoid}, n::Int64, r::Ptr{Float64}, x::Ptr{Float64})::Cint
  cb = unsafe_pointer_to_objref(ctx)::CoarseCallback
  try
    copyto!(cb.b, unsafe_wrap(Array, r, n))
    fill!(cb.x, 0.0)
    solve!(cb.x, cb.ns, cb.b)                               # solve!(xh, ns.coarsest_solver_cache, rh), GMGLinearSolvers.jl:474
    copyto!(unsafe_wrap(Array, x, n), cb.x)
    return Cint(0)
  catch
    return Cint(1)                                          # never unwind through the C frames
  end
end
function set_coarsest_solver!(ns::HipGMGNumericalSetup, solver, Acoarse::AbstractMatrix)
  h = ns.handle
  if solver isa Gridap.Algebra.LUSolver
    check(h, ccall((:gmg_set_coarse_solver, libgmgamd), Cint, (Ptr{Cvoid},Cint,Cint,Float64,Float64,Ptr{Cvoid},Ptr{Cvoid}),
                   h, GMG_COARSE_DENSE_INVERSE, 0, 0.0, 0.0, C_NULL, C_NULL))
  elseif solver isa GridapSolvers.LinearSolvers.CGSolver && solver.Pl isa JacobiLinearSolver && !solver.flexible
    t = solver.log.tols
    check(h, ccall((:gmg_set_coarse_solver, libgmgamd), Cint, (Ptr{Cvoid},Cint,Cint,Float64,Float64,Ptr{Cvoid},Ptr{Cvoid}),
                   h, GMG_COARSE_CG_JACOBI, t.maxiter, t.atol, t.rtol, C_NULL, C_NULL))
  else
    n  = size(Acoarse,1)
    cb = CoarseCallback(numerical_setup(symbolic_setup(solver,Acoarse),Acoarse), zeros(n), zeros(n))
    ns.keepalive = cb                                       # rooted for as long as the handle lives
    fptr = @cfunction(_coarse_trampoline, Cint, (Ptr{Cvoid},Int64,Ptr{Float64},Ptr{Float64}))
    check(h, ccall((:gmg_set_coarse_solver, libgmgamd), Cint, (Ptr{Cvoid},Cint,Cint,Float64,Float64,Ptr{Cvoid},Ptr{Cvoid}),
                   h, GMG_COARSE_HOST_CALLBACK, 0, 0.0, 0.0, fptr, pointer_from_objref(cb)))
  end
  check(h, ccall((:gmg_setup, libgmgamd), Cint, (Ptr{Cvoid},), h))
  return ns
end
coarse_log(ns::HipGMGNumericalSetup) = (r = Ref(GmgResult(0,0,0.0,0.0));
  check(ns.handle, ccall((:gmg_get_coarse_log, libgmgamd), Cint, (Ptr{Cvoid},Ref{GmgResult}), ns.handle, r)); r[])

# verbose kwarg: the library never prints; verbose > 0 keeps the GMG's own ConvergenceLog complete when it runs as a
# preconditioner inside the device Krylov solvers (GMGLinearSolvers.jl:627-640); gmg_log! copies it into ns.solver.log.
set_verbose!(ns::HipGMGNumericalSetup, v::Integer) =
  check(ns.handle, ccall((:gmg_set_verbose, libgmgamd), Cint, (Ptr{Cvoid},Cint), ns.handle, v))
function gmg_log!(ns::HipGMGNumericalSetup)
  log = ns.solver.log
  res = Ref(GmgResult(0,0,0.0,0.0)); hist = zeros(log.tols.maxiter+1)
  GC.@preserve hist check(ns.handle, ccall((:gmg_get_log, libgmgamd), Cint, (Ptr{Cvoid},Ref{GmgResult},Ptr{Float64},Cint),
                                          ns.handle, res, hist, length(hist)))
  return _fill_log!(log, res[], hist)
end

# PatchSolver numerical setup as the reference holds it (PatchSolvers.jl:100-150): patch_rows, patch_cols, and either the
# assembled patch matrices (Vector of Matrix{Float64}) or their lu! factorizations (collect_factorizations=true).
function set_patch_smoother!(ns::HipGMGNumericalSetup, lev::Integer, which::Cint, niter::Integer, ω::Real,
                             patch_rows::Gridap.Arrays.Table, patch_cols::Gridap.Arrays.Table;
                             patch_mats = nothing, factorizations = nothing, pivoting = true)
  ptrs = Int64.(patch_rows.ptrs); rows = Int64.(patch_rows.data); cols = Int64.(patch_cols.data)
  blocks = Float64[]; piv = Int32[]; are_factors = Cint(0)
  if !isnothing(factorizations)                              # LinearAlgebra.LU objects: F.factors (packed L\U), F.ipiv (1-based)
    for F in factorizations
      append!(blocks, vec(F.factors)); append!(piv, Int32.(F.ipiv))
    end
    are_factors = Cint(1)
  elseif !isnothing(patch_mats)
    for M in patch_mats
      append!(blocks, vec(Matrix{Float64}(M)))               # column-major, as stored
    end
  end
  GC.@preserve ptrs rows cols blocks piv begin
    check(ns.handle, ccall((:gmg_set_smoother_patch_matrices, libgmgamd), Cint,
      (Ptr{Cvoid},Cint,Cint,Cint,Float64,Cint,Int64,Ptr{Cvoid},Ptr{Cvoid},Ptr{Cvoid},Cint,Cint,Ptr{Float64},Cint,Ptr{Int32}),
      ns.handle, lev-1, which, niter, ω, pivoting ? GMG_PATCH_LU : GMG_PATCH_NOPIVOT, length(ptrs)-1, ptrs, rows, cols, 1, 8,
      isempty(blocks) ? C_NULL : pointer(blocks), are_factors, isempty(piv) ? C_NULL : pointer(piv)))
  end
  return ns
end

# Operators too large to hold at once: hand the rows over block by block (CSR blocks, 1-based, e.g. the row slabs an
# assembler produces); op = 0 (A), 1 (P), 2 (R).  The library keeps only the row-pattern form.
function set_operator_rows!(ns::HipGMGNumericalSetup, lev::Integer, op::Integer, nrows::Integer, ncols::Integer, row0::Integer,
                            rowptr::Vector{Int64}, colval::Vector{Int64}, nzval::Vector{Float64})
  GC.@preserve rowptr colval nzval check(ns.handle, ccall((:gmg_set_operator_rows, libgmgamd), Cint,
    (Ptr{Cvoid},Cint,Cint,Int64,Int64,Int64,Int64,Ptr{Cvoid},Ptr{Cvoid},Ptr{Float64},Cint,Cint),
    ns.handle, lev-1, op, nrows, ncols, row0-1, length(rowptr)-1, rowptr, colval, nzval, 1, 8))
  return ns
end

# ------------------------------------------------------------------------------------------------
# Distributed: PSparseMatrix / PVector (one MPI rank = one GPU), the shape of test/LinearSolvers/mpi/GMGTests.jl:5-8.
#
# The reference row-partitions every level with PartitionedArrays: local ids = own then ghost
# (JacobiLinearSolvers.jl:29-56 uses own_values / partition), `consistent!` moves owner -> ghost, `assemble!` adds
# ghost -> owner, dot/norm reduce over parts.  The library needs, per level and rank:
#   * the LOCAL operator rows of the owned dofs with columns numbered [own | ghost], ghosts grouped by owner rank
#     (PartitionedArrays orders ghosts arbitrarily -> `_ghost_permutation` below renumbers them once);
#   * the exchange plan of `consistent!`: neighbours, owned ids to send, ghost sub-ranges to receive.
# Communication runs over RCCL/xGMI (communicator seeded with an id broadcast over MPI) -- or, where ranks share a GPU,
# through MPI itself via the host-callback transport.
# ------------------------------------------------------------------------------------------------
import PartitionedArrays as PA
import MPI

struct HipDistributedLevel
  n_own    :: Int
  n_ghost  :: Int
  gperm    :: Vector{Int}          # PartitionedArrays ghost position -> library ghost position (grouped by owner)
  nbr      :: Vector{Int32}        # 0-based MPI ranks
  snd_ptr  :: Vector{Int64}
  snd_idx  :: Vector{Int64}        # 0-based owned local ids, per neighbour
  rcv_ptr  :: Vector{Int64}
end

# exchange plan of one part of a PRange (`indices = partition(axes(A,2))[part]`)
function _exchange_plan(indices, all_indices)
  own_to_local   = PA.own_to_local(indices)
  ghost_to_local = PA.ghost_to_local(indices)
  ghost_owner    = PA.ghost_to_owner(indices)               # 1-based part ids
  n_own, n_ghost = length(own_to_local), length(ghost_to_local)
  # ghosts grouped by owner (stable inside an owner: ascending global id)
  gids  = PA.ghost_to_global(indices)
  order = sortperm(collect(zip(ghost_owner, gids)))
  gperm = invperm(order)
  owners_sorted = ghost_owner[order]
  nbr_rcv = unique(owners_sorted)
  rcv_ptr = Int64[0]
  for q in nbr_rcv
    push!(rcv_ptr, rcv_ptr[end] + count(==(q), owners_sorted))
  end
  # what I send: the assembly graph reversed -- assembly sends ghosts to owners, consistent! sends owned values back
  nbrs_snd, nbrs_rcv = PA.assembly_neighbors(all_indices) |> x -> (map(identity,x[1]), map(identity,x[2]))
  lids_snd, lids_rcv = PA.assembly_local_indices(all_indices, nbrs_snd, nbrs_rcv)
  me = PA.part_id(indices)
  my_rcv_nbrs = PA.getany(nbrs_rcv); my_rcv_lids = PA.getany(lids_rcv)   # owned local ids that RECEIVE in assemble! = SEND in consistent!
  local_to_own = zeros(Int, PA.local_length(indices)); local_to_own[own_to_local] .= 1:n_own
  nbr = sort(unique(vcat(collect(my_rcv_nbrs), nbr_rcv)))
  snd_ptr = Int64[0]; snd_idx = Int64[]; rp = Int64[0]
  for q in nbr
    k = findfirst(==(q), my_rcv_nbrs)
    ids = isnothing(k) ? Int[] : collect(my_rcv_lids[k])
    append!(snd_idx, local_to_own[ids] .- 1)
    push!(snd_ptr, length(snd_idx))
    j = findfirst(==(q), nbr_rcv)
    push!(rp, rp[end] + (isnothing(j) ? 0 : rcv_ptr[j+1]-rcv_ptr[j]))
  end
  # NOTE: both sides must agree on the ORDER inside a message: PartitionedArrays sorts assembly lists by global id on
  # both ends, which is also the ghost order chosen above.
  return HipDistributedLevel(n_own, n_ghost, gperm, Int32.(nbr .- 1), snd_ptr, snd_idx, rp)
end

# local operator rows of the owned dofs, columns [own | ghost(grouped by owner)], as 1-based CSC for _set_op
function _local_operator(A::PA.PSparseMatrix, plan_cols::HipDistributedLevel)
  Aoo = PA.getany(PA.own_own_values(A)); Aog = PA.getany(PA.own_ghost_values(A))
  Pm  = sparse(1:plan_cols.n_ghost, plan_cols.gperm, ones(plan_cols.n_ghost), plan_cols.n_ghost, plan_cols.n_ghost)
  return SparseMatrixCSC{Float64,Int64}(hcat(sparse(Aoo), sparse(Aog) * Pm))
end

mutable struct HipDistributedGMGNumericalSetup <: Gridap.Algebra.NumericalSetup
  solver  :: HipGMGLinearSolver
  inner   :: HipGMGNumericalSetup
  plans   :: Vector{HipDistributedLevel}
  comm    :: MPI.Comm
end

const _MPI_CTX = Ref{Any}(nothing)
# host-staged transport (several ranks per GPU, or no RCCL): MPI point-to-point / Allreduce from callbacks
function _mpi_exchange(ctx::Ptr{Cvoid}, nnbr::Cint, nbr::Ptr{Int32}, snd::Ptr{Float64}, sp::Ptr{Int64}, rcv::Ptr{Float64}, rp::Ptr{Int64})::Cvoid
  comm = _MPI_CTX[]::MPI.Comm
  nb = unsafe_wrap(Array, nbr, nnbr); spv = unsafe_wrap(Array, sp, nnbr+1); rpv = unsafe_wrap(Array, rp, nnbr+1)
  sbuf = unsafe_wrap(Array, snd, spv[end]); rbuf = unsafe_wrap(Array, rcv, rpv[end])
  reqs = MPI.Request[]
  for k in 1:nnbr
    rpv[k+1] > rpv[k] && push!(reqs, MPI.Irecv!(view(rbuf, rpv[k]+1:rpv[k+1]), comm; source=nb[k], tag=7))
    spv[k+1] > spv[k] && push!(reqs, MPI.Isend(view(sbuf, spv[k]+1:spv[k+1]), comm; dest=nb[k], tag=7))
  end
  MPI.Waitall(reqs)
  return nothing
end
function _mpi_allreduce(ctx::Ptr{Cvoid}, vals::Ptr{Float64}, n::Cint)::Cvoid
  v = unsafe_wrap(Array, vals, n)
  MPI.Allreduce!(v, +, _MPI_CTX[]::MPI.Comm)
  return nothing
end

"""
    numerical_setup(ss::HipGMGSymbolicSetup, A::PSparseMatrix; comm=MPI.COMM_WORLD, transport=:rccl, replicate_from=nlev)

Distributed counterpart of `numerical_setup(ss, A::AbstractMatrix)`: `s.smatrices`, `s.interp`, `s.restrict` hold the
PSparseMatrix of every level (explicit R is required: a local Pᵀ misses off-rank rows).  Levels `replicate_from:nlev`
are gathered (`PA.to_trivial_partition`-style, here `_gather_global`) and computed redundantly on every rank -- the
analogue of coarse levels living on fewer ranks (ModelHierarchies.jl:80-148) without the redistribution traffic.
"""
function Gridap.Algebra.numerical_setup(ss::HipGMGSymbolicSetup, A::PA.PSparseMatrix;
                                        comm::MPI.Comm = MPI.COMM_WORLD, transport::Symbol = :rccl,
                                        replicate_from::Integer = length(ss.solver.smatrices))
  s = ss.solver
  nlev = length(s.smatrices)
  @assert !isnothing(s.restrict) "distributed runs need explicit restriction matrices"
  rank, nranks = MPI.Comm_rank(comm), MPI.Comm_size(comm)
  href = Ref{Ptr{Cvoid}}(C_NULL)
  check(C_NULL, ccall((:gmg_create, libgmgamd), Cint, (Ref{Ptr{Cvoid}},Cint,Cint), href, nlev, s.device))
  h = href[]
  inner = HipGMGNumericalSetup(s, h, PA.own_length(PA.getany(PA.partition(axes(A,1)))))
  finalizer(x -> (x.handle != C_NULL && ccall((:gmg_destroy, libgmgamd), Cint, (Ptr{Cvoid},), x.handle); x.handle = C_NULL), inner)
  # --- communicator -------------------------------------------------------------------------------
  if transport === :rccl
    uid = zeros(UInt8, 128)
    rank == 0 && check(C_NULL, ccall((:gmg_comm_unique_id, libgmgamd), Cint, (Cstring,Ptr{UInt8}), C_NULL, uid))
    MPI.Bcast!(uid, 0, comm)
    check(h, ccall((:gmg_comm_init_rccl, libgmgamd), Cint, (Ptr{Cvoid},Cstring,Ptr{UInt8},Cint,Cint), h, C_NULL, uid, rank, nranks))
  else
    _MPI_CTX[] = comm
    xf = @cfunction(_mpi_exchange, Cvoid, (Ptr{Cvoid},Cint,Ptr{Int32},Ptr{Float64},Ptr{Int64},Ptr{Float64},Ptr{Int64}))
    rf = @cfunction(_mpi_allreduce, Cvoid, (Ptr{Cvoid},Ptr{Float64},Cint))
    check(h, ccall((:gmg_comm_init_host, libgmgamd), Cint, (Ptr{Cvoid},Cint,Cint,Ptr{Cvoid},Ptr{Cvoid},Ptr{Cvoid}), h, rank, nranks, xf, rf, C_NULL))
  end
  # --- levels -------------------------------------------------------------------------------------
  plans = HipDistributedLevel[]
  mats = [l == 1 ? A : s.smatrices[l] for l in 1:nlev]
  for l in 1:nlev
    if l < replicate_from
      cols = PA.partition(axes(mats[l],2))
      pl = _exchange_plan(PA.getany(cols), cols)
      push!(plans, pl)
      GC.@preserve pl check(h, ccall((:gmg_set_partition, libgmgamd), Cint,
        (Ptr{Cvoid},Cint,Int64,Int64,Cint,Ptr{Int32},Ptr{Int64},Ptr{Int64},Ptr{Int64}),
        h, l-1, pl.n_own, pl.n_ghost, length(pl.nbr), pl.nbr, pl.snd_ptr, pl.snd_idx, pl.rcv_ptr))
      _set_op(:matrix, h, l-1, _local_operator(mats[l], pl))
    else
      _set_op(:matrix, h, l-1, _gather_global(mats[l], comm))                   # replicated level: global operator on every rank
    end
  end
  for l in 1:nlev-1
    if l+1 < replicate_from
      _set_op(:prolongation, h, l-1, _local_operator(s.interp[l], plans[l+1]))    # fine owned rows x coarse [own|ghost]
      _set_op(:restriction,  h, l-1, _local_operator(s.restrict[l], plans[l]))    # coarse owned rows x fine [own|ghost]
    elseif l+1 == replicate_from
      _set_op(:prolongation, h, l-1, _rows_global_cols(s.interp[l], comm))        # fine owned rows x GLOBAL coarse columns
      _set_op(:restriction,  h, l-1, _local_operator(s.restrict[l], plans[l]))
      gids = Int64.(PA.own_to_global(PA.getany(PA.partition(axes(s.restrict[l],1))))) .- 1
      GC.@preserve gids check(h, ccall((:gmg_set_replication, libgmgamd), Cint, (Ptr{Cvoid},Cint,Ptr{Int64},Int64), h, l, gids, length(gids)))
    else
      _set_op(:prolongation, h, l-1, _gather_global(s.interp[l], comm))
      _set_op(:restriction,  h, l-1, _gather_global(s.restrict[l], comm))
    end
    s.post_smoothers[l] === s.pre_smoothers[l] ? _set_smoother(h, l-1, GMG_PRE_AND_POST, s.pre_smoothers[l]) :
      (_set_smoother(h, l-1, GMG_PRE, s.pre_smoothers[l]); _set_smoother(h, l-1, GMG_POST, s.post_smoothers[l]))
  end
  tols = s.log.tols
  check(h, ccall((:gmg_set_options, libgmgamd), Cint, (Ptr{Cvoid},Cint,Cint,Cint,Float64,Float64), h,
                 s.mode == :preconditioner ? 0 : 1, s.cycle_type == :v_cycle ? 0 : (s.cycle_type == :w_cycle ? 1 : 2),
                 tols.maxiter, tols.atol, tols.rtol))
  check(h, ccall((:gmg_setup, libgmgamd), Cint, (Ptr{Cvoid},), h))
  return HipDistributedGMGNumericalSetup(s, inner, plans, comm)
end

# gather a (small) PSparseMatrix on every rank as one SparseMatrixCSC in GLOBAL numbering
function _gather_global(M::PA.PSparseMatrix, comm::MPI.Comm)
  I, J, V = PA.getany(map(PA.partition(M), PA.partition(axes(M,1)), PA.partition(axes(M,2))) do Ml, rows, cols
    i, j, v = findnz(sparse(Ml))
    keep = [PA.local_to_owner(rows)[ii] == PA.part_id(rows) for ii in i]      # owned rows only
    (PA.local_to_global(rows)[i[keep]], PA.local_to_global(cols)[j[keep]], v[keep])
  end)
  Ia = MPI.Allgatherv!(MPI.VBuffer(Int64.(I), nothing), comm); Ja = MPI.Allgatherv!(MPI.VBuffer(Int64.(J), nothing), comm)
  Va = MPI.Allgatherv!(MPI.VBuffer(Float64.(V), nothing), comm)
  return SparseMatrixCSC{Float64,Int64}(sparse(Ia, Ja, Va, size(M,1), size(M,2)))
end
# owned rows of M with GLOBAL column ids (the prolongation across the distributed -> replicated boundary)
function _rows_global_cols(M::PA.PSparseMatrix, comm::MPI.Comm)
  rows = PA.getany(PA.partition(axes(M,1))); cols = PA.getany(PA.partition(axes(M,2)))
  Ml = sparse(PA.getany(PA.partition(M)))
  i, j, v = findnz(Ml[PA.own_to_local(rows), :])
  return SparseMatrixCSC{Float64,Int64}(sparse(i, PA.local_to_global(cols)[j], v, PA.own_length(rows), size(M,2)))
end

# solve!(x::PVector, ns, b::PVector): vectors cross the boundary as their OWNED values (JacobiLinearSolvers.jl:49-56)
function Gridap.Algebra.solve!(x::PA.PVector, ns::HipDistributedGMGNumericalSetup, b::PA.PVector)
  xo = PA.getany(PA.own_values(x)); bo = PA.getany(PA.own_values(b))
  xv, bv = Vector{Float64}(xo), Vector{Float64}(bo)
  solve!(xv, ns.inner, bv)                                                   # gmg_apply on the owned entries
  copyto!(xo, xv)
  PA.consistent!(x) |> wait
  return x
end
function Gridap.Algebra.solve!(x::PA.PVector, ns::HipKrylovNumericalSetup{<:HipCGSolver,<:HipDistributedGMGNumericalSetup}, b::PA.PVector)
  xo = PA.getany(PA.own_values(x)); bo = PA.getany(PA.own_values(b))
  xv, bv = Vector{Float64}(xo), Vector{Float64}(bo)
  solve!(xv, HipKrylovNumericalSetup(ns.solver, ns.P_ns.inner), bv)          # gmg_cg_solve: dots are all-reduced in the library
  copyto!(xo, xv)
  PA.consistent!(x) |> wait
  return x
end
function Gridap.Algebra.numerical_setup(ss::HipKrylovSymbolicSetup, A::PA.PSparseMatrix; kwargs...)
  P = ss.solver isa HipFGMRESSolver ? ss.solver.Pr : ss.solver.Pl
  return HipKrylovNumericalSetup(ss.solver, numerical_setup(symbolic_setup(P,A),A; kwargs...))
end
Gridap.Algebra.symbolic_setup(s::HipGMGLinearSolver, ::PA.PSparseMatrix) = HipGMGSymbolicSetup(s)
Gridap.Algebra.symbolic_setup(s::Union{HipCGSolver,HipFGMRESSolver}, ::PA.PSparseMatrix) = HipKrylovSymbolicSetup(s)

# Distributed block systems (BlockPMatrix / BlockPVector): the block handle gets its own communicator and one exchange plan
# per block; blocks are passed as local rows with [own | ghost] columns (_local_operator), vectors as owned values.
function block_comm_init!(bh::Ptr{Cvoid}, comm::MPI.Comm; transport::Symbol = :rccl)
  rank, nranks = MPI.Comm_rank(comm), MPI.Comm_size(comm)
  if transport === :rccl
    uid = zeros(UInt8, 128)
    rank == 0 && check(C_NULL, ccall((:gmg_comm_unique_id, libgmgamd), Cint, (Cstring,Ptr{UInt8}), C_NULL, uid))
    MPI.Bcast!(uid, 0, comm)
    check_block(bh, ccall((:gmg_block_comm_init_rccl, libgmgamd), Cint, (Ptr{Cvoid},Cstring,Ptr{UInt8},Cint,Cint), bh, C_NULL, uid, rank, nranks))
  else
    _MPI_CTX[] = comm
    xf = @cfunction(_mpi_exchange, Cvoid, (Ptr{Cvoid},Cint,Ptr{Int32},Ptr{Float64},Ptr{Int64},Ptr{Float64},Ptr{Int64}))
    rf = @cfunction(_mpi_allreduce, Cvoid, (Ptr{Cvoid},Ptr{Float64},Cint))
    check_block(bh, ccall((:gmg_block_comm_init_host, libgmgamd), Cint, (Ptr{Cvoid},Cint,Cint,Ptr{Cvoid},Ptr{Cvoid},Ptr{Cvoid}), bh, rank, nranks, xf, rf, C_NULL))
  end
end
function block_set_partition!(bh::Ptr{Cvoid}, j::Integer, pl::HipDistributedLevel)
  GC.@preserve pl check_block(bh, ccall((:gmg_block_set_partition, libgmgamd), Cint,
    (Ptr{Cvoid},Cint,Int64,Int64,Cint,Ptr{Int32},Ptr{Int64},Ptr{Int64},Ptr{Int64}),
    bh, j-1, pl.n_own, pl.n_ghost, length(pl.nbr), pl.nbr, pl.snd_ptr, pl.snd_idx, pl.rcv_ptr))
end
# rhs form of a patch prolongation when it is not the level operator (StokesGMG.jl:125-127)
function set_prolongation_correction_rhs!(ns::HipGMGNumericalSetup, lev::Integer, G::SparseMatrixCSC{Float64,Int64})
  GC.@preserve G check(ns.handle, ccall((:gmg_set_prolongation_patch_correction_rhs, libgmgamd), Cint,
    (Ptr{Cvoid},Cint,Int64,Int64,Ptr{Cvoid},Ptr{Cvoid},Ptr{Float64},Cint,Cint,Cint),
    ns.handle, lev-1, size(G,1), nnz(G), G.colptr, G.rowval, G.nzval, GMG_CSC, 1, 8))
  return ns
end
# FGMRES with a left preconditioner as well (KrylovUtils.jl:14-18,46-50): pl = 0 none, 2 Jacobi, 3 the finest pre-smoother
function fgmres_solve_pl!(x::Vector{Float64}, ns::HipKrylovNumericalSetup{<:HipFGMRESSolver}, b::Vector{Float64}, pl::Integer)
  s, h = ns.solver, ns.P_ns.handle
  tols = s.log.tols
  res  = Ref(GmgResult(0,0,0.0,0.0)); hist = zeros(tols.maxiter+1)
  GC.@preserve x b hist check(h, ccall((:gmg_fgmres_solve_pl, libgmgamd), Cint,
    (Ptr{Cvoid},Ptr{Float64},Ptr{Float64},Cint,Cint,Cint,Cint,Cint,Float64,Float64,Cint,Cint,Ref{GmgResult},Ptr{Float64},Cint),
    h, b, x, GMG_MEM_HOST, s.m, s.restart ? 1 : 0, s.m_add, tols.maxiter, tols.atol, tols.rtol, 1, pl, res, hist, length(hist)))
  _fill_log!(s.log, res[], hist)
  return x
end

end # module
