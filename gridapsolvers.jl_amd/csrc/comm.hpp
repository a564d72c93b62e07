// comm.hpp -- inter-GPU transport of the row-partitioned hierarchy.
//
// The reference gets its communication from PartitionedArrays over MPI:
//   consistent!(v)  owner -> ghost copy      (PatchSolvers.jl:231,256; inside mul!(::PVector,::PSparseMatrix,::PVector))
//   dot / norm      sum over parts           (CGSolvers.jl:85,95,105,111)
// Here one process drives one GPU and the two primitives are
//   halo exchange : pack kernel -> grouped ncclSend/ncclRecv over xGMI, received straight
//                   into the ghost segment of the vector (ghosts are ordered by owner rank)
//   all-reduce    : ncclAllReduce(sum) of a few doubles on the handle's stream.
// RCCL is bound with dlopen so that the library a host process already uses (e.g. the
// librccl.so bundled with PyTorch-ROCm, or /opt/rocm/lib/librccl.so.1 for a Julia host)
// is the one that gets loaded; the communicator is created from a caller-supplied
// ncclUniqueId (broadcast by the host language: MPI.bcast / torch.distributed).
//
// A second transport ("host") stages through pinned host memory and calls back into the
// host language (MPI in Julia, gloo in the Python tests).  It exists so that the whole
// distributed algorithm can be run and tested where RCCL cannot (several ranks sharing
// one GPU); it is not meant for production throughput.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <string>
#include <vector>

namespace gmg {

// minimal RCCL ABI (rccl.h: ncclUniqueId is 128 opaque bytes, ncclDouble = 8, ncclSum = 0)
struct NcclUniqueId { char internal[128]; };
typedef void *NcclComm;

struct RcclApi {
  void *dl = nullptr;
  int (*GetUniqueId)(NcclUniqueId *) = nullptr;
  int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*Send)(const void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  // optional (diagnostics: gmg_get_comm_info)
  int (*CommCount)(NcclComm, int *) = nullptr;
  int (*CommCuDevice)(NcclComm, int *) = nullptr;
  int (*CommUserRank)(NcclComm, int *) = nullptr;

  bool load(const char *path, std::string &err)
  {
    if (dl) return true;
    const char *cands[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *c : cands) {
      if (!c || !*c) continue;
      dl = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
      if (dl) break;
    }
    if (!dl) { err = std::string("cannot dlopen librccl: ") + dlerror(); return false; }
    auto sym = [&](const char *n) { void *p = dlsym(dl, n); if (!p) err = std::string("missing RCCL symbol ") + n; return p; };
    GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    Send = (decltype(Send))sym("ncclSend");
    Recv = (decltype(Recv))sym("ncclRecv");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    CommCount = (decltype(CommCount))dlsym(dl, "ncclCommCount");
    CommCuDevice = (decltype(CommCuDevice))dlsym(dl, "ncclCommCuDevice");
    CommUserRank = (decltype(CommUserRank))dlsym(dl, "ncclCommUserRank");
    return GetUniqueId && CommInitRank && CommDestroy && Send && Recv && GroupStart && GroupEnd && AllReduce;
  }
};

constexpr int kNcclDouble = 8;
constexpr int kNcclSum = 0;

enum CommKind { COMM_NONE = 0, COMM_RCCL = 1, COMM_HOST = 2 };

typedef void (*HostExchangeFn)(void *ctx, int nnbr, const int32_t *nbr_rank, const double *sendbuf,
                               const int64_t *snd_ptr, double *recvbuf, const int64_t *rcv_ptr);
typedef void (*HostAllreduceFn)(void *ctx, double *vals, int n);

struct Comm {
  int kind = COMM_NONE;
  int rank = 0, nranks = 1;
  RcclApi api;
  NcclComm comm = nullptr;
  HostExchangeFn xfn = nullptr;
  HostAllreduceFn rfn = nullptr;
  void *ctx = nullptr;
  // gmg_comm_set_loopback: the communicator has ONE real rank, the partition it serves was written for `nranks` of them and folded
  // onto this one (every neighbour is this rank itself).  Every message is then a self send / receive -- the whole exchange path
  // (pack, grouped ncclSend/ncclRecv on the communication stream, events, boundary fix-up, ncclAllReduce) runs on one GPU.
  bool loopback = false;
  int peer(int r) const { return loopback ? 0 : r; }
  int real_ranks() const { return loopback ? 1 : nranks; }
};

// Per-level exchange plan (PartitionedArrays: assembly_neighbors + local indices)
struct HaloPlan {
  bool present = false;
  int64_t n_own = 0, n_ghost = 0;
  std::vector<int32_t> nbr;
  std::vector<int64_t> snd_ptr, rcv_ptr, h_snd_idx;
  int64_t *d_snd_idx = nullptr;
  double *d_sendbuf = nullptr;
  double *d_recvbuf = nullptr;                 // assemble!: ghost contributions arriving for the rows in snd_idx
  double *h_send = nullptr, *h_recv = nullptr; // pinned, host transport only
  // boundary row -> its slots in the send buffer (pack fused into ghost_fix_kernel); null when some sent row has no ghost column
  int64_t *d_pk_ptr = nullptr;
  int32_t *d_pk_slot = nullptr;
  // Overlapping layout (gmg_set_partition_overlap): the local vector holds owned and ghost entries in ONE numbering chosen by
  // the caller (a structured box partition: the extended box in lexicographic order), the local matrix has a row for every
  // local entry, and `depth` ghost layers are kept consistent by one exchange -- so `depth` sweeps of a distance-1 operator
  // run between two exchanges (rows of ghost layer j are recomputed redundantly and stay exact for depth - j sweeps).
  // Received values are scattered through rcv_idx instead of landing in a contiguous ghost segment.
  bool ovl = false;
  int depth = 1;
  std::vector<int64_t> h_rcv_idx;
  int64_t *d_rcv_idx = nullptr;
  double *d_unpack = nullptr;                  // [n_ghost] landing buffer of the receives
  int64_t nsend() const { return snd_ptr.empty() ? 0 : snd_ptr.back(); }
};

// sendbuf[i] = v[idx[i]]
__global__ void halo_pack_kernel(int64_t n, const int64_t *__restrict__ idx, const double *__restrict__ v,
                                 double *__restrict__ sendbuf)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sendbuf[i] = v[idx[i]];
}
// v[idx[i]] = buf[i]  (overlapping layout: received ghost values into their places)
__global__ void halo_unpack_kernel(int64_t n, const int64_t *__restrict__ idx, const double *__restrict__ buf,
                                   double *__restrict__ v)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[idx[i]] = buf[i];
}
// dst[di[i]] = src[si[i]]  (redistribution between two partitions of a level: the entries that stay on this rank)
__global__ void gather_scatter_kernel(int64_t n, const int64_t *__restrict__ si, const int64_t *__restrict__ di,
                                      const double *__restrict__ src, double *__restrict__ dst)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[di[i]] = src[si[i]];
}
// assemble!(v) (PatchSolvers.jl:254): v[idx[i]] += buf[i] -- the ghost copies' contributions added to the owner's entry.
// One launch per neighbour, in neighbour order: targets are unique inside a neighbour's list, so the sum order is fixed.
__global__ void halo_unpack_add_kernel(int64_t n, const int64_t *__restrict__ idx, const double *__restrict__ buf,
                                       double *__restrict__ v)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[idx[i]] = v[idx[i]] + buf[i];
}
// dx .= omega .* dx ; x .+= dx  (RichardsonSmoothers.jl:92-93) after the patch contributions have been assembled
__global__ void relax_update_kernel(int64_t n, double omega, double *__restrict__ dx, double *__restrict__ x)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double d = omega * dx[i];
    dx[i] = d;
    x[i] = x[i] + d;
  }
}
__global__ void scale_inplace_kernel(int64_t n, double omega, double *__restrict__ dx)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dx[i] = omega * dx[i];
}
// full[gid[i]] = r[i]  (coarse-level gather into the replicated vector)
__global__ void scatter_gid_kernel(int64_t n, const int64_t *__restrict__ gid, const double *__restrict__ r,
                                   double *__restrict__ full)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) full[gid[i]] = r[i];
}
// rectangular dense GEMV: x[i] = sum_j M[i*ncols+j] r[j], one wave per row
__global__ __launch_bounds__(256) void dense_gemv_rect_kernel(int nrows, int ncols, const double *__restrict__ M,
                                                              const double *__restrict__ r, double *__restrict__ x)
{
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= nrows) return;
  const double *row = M + (size_t)wave * ncols;
  double s = 0.0;
  for (int j = lane; j < ncols; j += 64) s += row[j] * r[j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) x[wave] = s;
}
__global__ void sqrt_inplace_kernel(double *v) { v[0] = sqrt(v[0]); }

// Rows that reference ghost columns ("boundary rows") finish their mat-vec after the halo has
// arrived: g = sum_k val[k] * v[col[k]] over the row's ghost entries, then
//   MODE 0: y[row] += g                         (y = A x)
//   MODE 1: y[row] -= g                         (y -= A x ; y = b - A x)
//   MODE 2: y[row] -= g ; s[row] = omega*(dinv[row]*y[row])   (fused sweep: r and s = w*Dinv*r)
//   MODE 3: g = sum_k val[k] * (omega*(du*v[col[k]])) ; y[row] -= g   (the sweep that gathers r itself, uniform 1/diag du:
//           sells_rsweep_kernel; the send buffer then carries the new r of the boundary rows)
// The owned-column part ran while the exchange was in flight (own x own / own x ghost split of
// the local matrix, the standard PartitionedArrays-style overlap).
// pk_ptr / pk_slot / sendbuf (MODE 2, optional): the boundary rows are exactly the rows the neighbours need, so the
// fix-up also writes the row's new s into its slots of the send buffer -- the next sweep's exchange starts without a
// pack kernel.
template <int MODE>
__global__ void ghost_fix_kernel(int64_t nb, const int32_t *__restrict__ rows, const int64_t *__restrict__ ptr,
                                 const int32_t *__restrict__ col, const double *__restrict__ val,
                                 const double *__restrict__ v, double *__restrict__ y, const double *__restrict__ dinv,
                                 double omega, double *__restrict__ s_out, const int64_t *__restrict__ pk_ptr = nullptr,
                                 const int32_t *__restrict__ pk_slot = nullptr, double *__restrict__ sendbuf = nullptr, double du = 0.0)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb) return;
  double g = 0.0;
  if (MODE == 3) { for (int64_t k = ptr[i]; k < ptr[i + 1]; ++k) g += val[k] * (omega * (du * v[col[k]])); }
  else
  for (int64_t k = ptr[i]; k < ptr[i + 1]; ++k) g += val[k] * v[col[k]];
  const int32_t row = rows[i];
  if (MODE == 0) y[row] = y[row] + g;
  else if (MODE == 1) y[row] = y[row] - g;
  else if (MODE == 3) {
    const double rn = y[row] - g;
    y[row] = rn;
    if (pk_ptr)
      for (int64_t k = pk_ptr[i]; k < pk_ptr[i + 1]; ++k) sendbuf[pk_slot[k]] = rn;
  } else {
    const double rn = y[row] - g;
    y[row] = rn;
    const double sn = omega * (dinv[row] * rn);
    s_out[row] = sn;
    if (pk_ptr)
      for (int64_t k = pk_ptr[i]; k < pk_ptr[i + 1]; ++k) sendbuf[pk_slot[k]] = sn;
  }
}

// The same with the boundary rows' ghost entries in slices of 64 rows, column-major (SELL-64: entry e of the slice's lane l at
// soff[slice] + 64 e + l, rows padded to the slice's longest with (column 0, value 0.0) that the row's own length keeps out of the sum):
// a wave's loads of one entry are contiguous and the entries of a row are independent loads -- the CSR form above walks each row with one
// thread, 12-byte loads at a stride of the row length, one dependent chain per row (measured on one rank of BASELINE config 4, alone on the
// GPU: 25.5 us per sweep for 2.5e5 boundary rows, 0.5 ms of a 6.1 ms iteration; profiles/r05_tuning.md section 9).  Same products in the
// same order: bit-identical.
// DICT: the values come from a dictionary of <= 256 distinct doubles (constant-coefficient operators have a handful), one byte per
// entry instead of eight -- 12 -> 5 bytes per ghost entry; `sval` then points at the codes, `dict` at the values (staged in LDS).
template <int MODE, bool DICT = false>
__global__ __launch_bounds__(256) void ghost_fix_sell_kernel(int64_t nb, const int32_t *__restrict__ rows, const int32_t *__restrict__ glen,
                                                             const int64_t *__restrict__ soff, const int32_t *__restrict__ scol,
                                                             const void *__restrict__ sval_, const double *__restrict__ dict, const double *__restrict__ v,
                                                             double *__restrict__ y, const double *__restrict__ dinv, double omega,
                                                             double *__restrict__ s_out, const int64_t *__restrict__ pk_ptr = nullptr,
                                                             const int32_t *__restrict__ pk_slot = nullptr, double *__restrict__ sendbuf = nullptr,
                                                             double du = 0.0)
{
  __shared__ double sd[DICT ? 256 : 1];
  if (DICT) { sd[threadIdx.x] = dict[threadIdx.x]; __syncthreads(); }
  const double *__restrict__ sval = static_cast<const double *>(sval_);
  const uint8_t *__restrict__ scode = static_cast<const uint8_t *>(sval_);
  auto val = [&](int64_t at) -> double { return DICT ? sd[scode[at]] : sval[at]; };
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb) return;
  const int64_t slice = i >> 6;
  const int64_t base = soff[slice] + (i & 63);
  const int w = (int)((soff[slice + 1] - soff[slice]) >> 6);
  const int len = glen[i];
  const int32_t row = rows[i];
  const double y0 = y[row];
  auto f = [&](double x) -> double { return MODE == 3 ? omega * (du * x) : x; };
  double g = 0.0;
  int e = 0;
  for (; e + 4 <= w; e += 4) {
    const int64_t b = base + (int64_t)e * 64;
    const int32_t c0 = scol[b], c1 = scol[b + 64], c2 = scol[b + 128], c3 = scol[b + 192];
    const double a0 = val(b), a1 = val(b + 64), a2 = val(b + 128), a3 = val(b + 192);
    const double x0 = v[c0], x1 = v[c1], x2 = v[c2], x3 = v[c3];
    if (e < len) g += a0 * f(x0);
    if (e + 1 < len) g += a1 * f(x1);
    if (e + 2 < len) g += a2 * f(x2);
    if (e + 3 < len) g += a3 * f(x3);
  }
  for (; e < w; ++e) {
    const int64_t b = base + (int64_t)e * 64;
    const int32_t c0 = scol[b];
    const double a0 = val(b);
    const double x0 = v[c0];
    if (e < len) g += a0 * f(x0);
  }
  if (MODE == 0) y[row] = y0 + g;
  else if (MODE == 1) y[row] = y0 - g;
  else if (MODE == 3) {
    const double rn = y0 - g;
    y[row] = rn;
    if (pk_ptr)
      for (int64_t k = pk_ptr[i]; k < pk_ptr[i + 1]; ++k) sendbuf[pk_slot[k]] = rn;
  } else {
    const double rn = y0 - g;
    y[row] = rn;
    const double sn = omega * (dinv[row] * rn);
    s_out[row] = sn;
    if (pk_ptr)
      for (int64_t k = pk_ptr[i]; k < pk_ptr[i + 1]; ++k) sendbuf[pk_slot[k]] = sn;
  }
}

} // namespace gmg
