// prost/prox/prox_elem_operation.inl -- the generic gfx950 kernel behind ProxElemOperation<T, ELEM_OPERATION> and the
// member definitions of the template.  Include it from the plugin's .hip source (hipcc --offload-arch=gfx950), after
// the operation's own header; the reference's counterpart is include/prost/prox/prox_elem_operation.inl:32-222.
//
// What the reference does per launch (:59-94): one thread per element group, Vector views straight over global memory,
// coefficients gathered per thread, `ELEM_OPERATION op(coeffs, dim, sh_mem); op(res, arg, tau_diag, tau, invert_tau)`.
// The same contract here, laid out for CDNA4:
//
//   * TILE path (dim <= 8, the operation writes every res[i]).  A lane owns VEC = 16 / sizeof(T) consecutive element
//     groups and keeps their arg / tau_diag / res components in a private tile that the Vector views index with
//     compile-time constants -- i.e. registers.  Planar layout: per component ONE 16-byte load per lane (a wavefront
//     reads 1 KiB contiguous per instruction); interleaved layout: the lane's VEC * dim values are contiguous and move
//     as dim 16-byte accesses.  Loads whose values the operation never reads (e.g. tau_diag[1..] of a norm-type
//     operation) are removed by the compiler, so HBM traffic is what the OPERATION touches, as with direct views.
//     Per-group coefficient vectors are read 16 bytes per lane as well.  Operands that are not 16-byte aligned (or a
//     plane stride that is not) run the same kernel with VEC = 1; the count % VEC tail runs it with VEC = 1 too.
//   * DIRECT path (any dim, or ELEM_OPERATION::kPartialResult): one group per lane, views over HBM -- coalesced for
//     the planar layout (component i of 64 consecutive groups = 64 consecutive values).
//   * SharedMem slices: GetSharedMemCount(dim) entries per lane in dynamic LDS, entry-major (shared_mem.hpp), reused
//     by the VEC groups of a lane one after the other.
//
// One workgroup = 256 lanes = 4 wavefronts; the launch goes to prost::CurrentStream() and returns without waiting.
#ifndef PROST_PROX_ELEM_OPERATION_INL_
#define PROST_PROX_ELEM_OPERATION_INL_
#include <hip/hip_runtime.h>

#include <sstream>

#include "prost/common.hpp"
#include "prost/exception.hpp"
#include "prost/prox/prox_elem_operation.hpp"
#include "prost/prox/shared_mem.hpp"
#include "prost/prox/vector.hpp"

namespace prost {
namespace elemop_kernel {

constexpr int kLanes = 256;

template <typename T, int N>
struct alignas(sizeof(T) * N) Pack {
  T v[N];
};
/// N consecutive values; N * sizeof(T) = 16 bytes moves as one dwordx4 access, N = 1 as an element access
template <typename T, int N>
__device__ __forceinline__ void LoadPack(const T* p, T* out) {
  const Pack<T, N> q = *reinterpret_cast<const Pack<T, N>*>(p);
#pragma unroll
  for (int j = 0; j < N; j++) out[j] = q.v[j];
}
template <typename T, int N>
__device__ __forceinline__ void StorePack(T* p, const T* in) {
  Pack<T, N> q;
#pragma unroll
  for (int j = 0; j < N; j++) q.v[j] = in[j];
  *reinterpret_cast<Pack<T, N>*>(p) = q;
}

/// the scalar step of a launch: the host's value, or -- inside a batch of iterations whose step-size rule runs on the device (prox.hpp:
/// StepView) -- a device scalar read when the kernel runs; a raised stop word ends the kernel before it touches the result
template <typename T>
struct StepArg {
  T tau; const T* step; const int* stop;
  __device__ __forceinline__ bool resolve(T& t) const {
    t = tau;
    if (step == nullptr) return true;
    if (*stop != 0) return false;
    t = *step;
    return true;
  }
};

/// constructs the operation the way the reference kernel does (prox_elem_operation.inl:53, :91) and applies it
template <typename T, class OP>
__device__ __forceinline__ void Apply(Vector<T>& res, const Vector<const T>& arg, const Vector<const T>& tau_diag, T tau, bool invert_tau,
                                      T* coeffs_local, size_t dim) {
  SharedMem<typename OP::SharedMemType, typename OP::GetSharedMemCount> sh_mem(dim, threadIdx.x);
  if constexpr (OP::kCoeffsCount != 0) {
    OP op(coeffs_local, dim, sh_mem);
    op(res, arg, tau_diag, tau, invert_tau);
  } else {
    (void)coeffs_local;
    OP op(dim, sh_mem);
    op(res, arg, tau_diag, tau, invert_tau);
  }
}

// ---- TILE path: VEC element groups per lane in registers; groups [first, first + n) of `count`, n % VEC == 0 -------
template <typename T, class OP, int DIM, int VEC, bool INTERLEAVED>
__global__ void __launch_bounds__(kLanes) ProxElemOperationTileKernel(T* d_res, const T* d_arg, const T* d_tau, StepArg<T> step, bool invert_tau,
                                                                      size_t count, size_t first, size_t n,
                                                                      ElemOpCoefficients<T, OP> coeffs) {
  const size_t lane = (size_t)blockIdx.x * kLanes + threadIdx.x;
  if (lane * VEC >= n) return;
  T tau;
  if (!step.resolve(tau)) return;
  const size_t e0 = first + lane * VEC;          // first group of this lane
  constexpr int NC = OP::kCoeffsCount ? (int)OP::kCoeffsCount : 1;

  // private tile, indexed by the SAME rule as the global vector with count := VEC (vector.hpp)
  T a[DIM * VEC], t[DIM * VEC], r[DIM * VEC];
  if (INTERLEAVED) {
#pragma unroll
    for (int k = 0; k < DIM; k++) {
      LoadPack<T, VEC>(d_arg + e0 * DIM + k * VEC, a + k * VEC);
      LoadPack<T, VEC>(d_tau + e0 * DIM + k * VEC, t + k * VEC);
    }
  } else {
#pragma unroll
    for (int i = 0; i < DIM; i++) {
      LoadPack<T, VEC>(d_arg + e0 + count * i, a + i * VEC);
      LoadPack<T, VEC>(d_tau + e0 + count * i, t + i * VEC);
    }
  }
  if constexpr (!OP::kWritesAllComponents) {
    // safe default: components the operation does not assign keep their old content, as with the reference's views over global memory
    if (INTERLEAVED) {
#pragma unroll
      for (int k = 0; k < DIM; k++) LoadPack<T, VEC>(d_res + e0 * DIM + k * VEC, r + k * VEC);
    } else {
#pragma unroll
      for (int i = 0; i < DIM; i++) LoadPack<T, VEC>(d_res + e0 + count * i, r + i * VEC);
    }
  }
  T cv[NC][VEC];
  if constexpr (OP::kCoeffsCount != 0) {
#pragma unroll
    for (int k = 0; k < NC; k++) {
      if (coeffs.dev_p[k] != nullptr) LoadPack<T, VEC>(coeffs.dev_p[k] + e0, cv[k]);      // wave-uniform branch
      else {
#pragma unroll
        for (int j = 0; j < VEC; j++) cv[k][j] = coeffs.val[k];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; j++) {
    Vector<T> res(VEC, DIM, INTERLEAVED, j, r);
    const Vector<const T> arg(VEC, DIM, INTERLEAVED, j, a);
    const Vector<const T> tau_diag(VEC, DIM, INTERLEAVED, j, t);
    T coeffs_local[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) coeffs_local[k] = cv[k][j];
    Apply<T, OP>(res, arg, tau_diag, tau, invert_tau, coeffs_local, DIM);
  }
  if (INTERLEAVED) {
#pragma unroll
    for (int k = 0; k < DIM; k++) StorePack<T, VEC>(d_res + e0 * DIM + k * VEC, r + k * VEC);
  } else {
#pragma unroll
    for (int i = 0; i < DIM; i++) StorePack<T, VEC>(d_res + e0 + count * i, r + i * VEC);
  }
}

// ---- DIRECT path: one element group per lane, views over HBM (the reference's access pattern) ----------------------
template <typename T, class OP>
__global__ void __launch_bounds__(kLanes) ProxElemOperationKernel(T* d_res, const T* d_arg, const T* d_tau, StepArg<T> step, bool invert_tau, size_t count,
                                                                  size_t dim, ElemOpCoefficients<T, OP> coeffs, bool interleaved) {
  const size_t tx = (size_t)blockIdx.x * kLanes + threadIdx.x;
  if (tx >= count) return;
  T tau;
  if (!step.resolve(tau)) return;
  constexpr int NC = OP::kCoeffsCount ? (int)OP::kCoeffsCount : 1;
  Vector<T> res(count, dim, interleaved, tx, d_res);
  const Vector<const T> arg(count, dim, interleaved, tx, d_arg);
  const Vector<const T> tau_diag(count, dim, interleaved, tx, d_tau);
  T coeffs_local[NC];
  if constexpr (OP::kCoeffsCount != 0) {
#pragma unroll
    for (int k = 0; k < NC; k++) coeffs_local[k] = coeffs.dev_p[k] != nullptr ? coeffs.dev_p[k][tx] : coeffs.val[k];
  }
  Apply<T, OP>(res, arg, tau_diag, tau, invert_tau, coeffs_local, dim);
}

inline bool Aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline void CheckLaunch(const char* what) {
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    std::stringstream ss;
    ss << "HIP error (" << what << "): " << hipGetErrorString(err) << std::endl;
    throw Exception(ss.str());
  }
}

template <typename T, class OP, int DIM, int VEC, bool INTERLEAVED>
void LaunchTile(T* res, const T* arg, const T* tau_diag, const StepArg<T>& tau, bool invert_tau, size_t count, size_t first, size_t n,
                const ElemOpCoefficients<T, OP>& coeffs, size_t lds_bytes, hipStream_t stream) {
  if (n == 0) return;
  const size_t lanes = n / VEC;
  hipLaunchKernelGGL((ProxElemOperationTileKernel<T, OP, DIM, VEC, INTERLEAVED>), dim3((unsigned)((lanes + kLanes - 1) / kLanes)), dim3(kLanes),
                     lds_bytes, stream, res, arg, tau_diag, tau, invert_tau, count, first, n, coeffs);
}

/// DIM-th instance of the tile path if the operation admits that dimension (kDim == 0: any)
template <typename T, class OP, int DIM>
bool TryTile(size_t dim, bool interleaved, T* res, const T* arg, const T* tau_diag, const StepArg<T>& tau, bool invert_tau, size_t count,
             const ElemOpCoefficients<T, OP>& coeffs, size_t lds_bytes, hipStream_t stream) {
  if constexpr (OP::kDim != 0 && OP::kDim != DIM) {
    return false;
  } else {
    if (dim != (size_t)DIM) return false;
    constexpr int V = 16 / (int)sizeof(T);
    bool wide = Aligned16(res) && Aligned16(arg) && Aligned16(tau_diag) && count >= (size_t)V;
    if (!interleaved && DIM > 1) wide = wide && (count * sizeof(T)) % 16 == 0;      // every component plane starts 16-byte aligned
    for (size_t k = 0; k < OP::kCoeffsCount; k++) wide = wide && (coeffs.dev_p[k] == nullptr || Aligned16(coeffs.dev_p[k]));
    const size_t body = wide ? count - count % V : 0;
    if constexpr (DIM > 1) {
      if (interleaved) {
        LaunchTile<T, OP, DIM, V, true>(res, arg, tau_diag, tau, invert_tau, count, 0, body, coeffs, lds_bytes, stream);
        LaunchTile<T, OP, DIM, 1, true>(res, arg, tau_diag, tau, invert_tau, count, body, count - body, coeffs, lds_bytes, stream);
        return true;
      }
    }
    LaunchTile<T, OP, DIM, V, false>(res, arg, tau_diag, tau, invert_tau, count, 0, body, coeffs, lds_bytes, stream);      // dim 1: the layouts coincide
    LaunchTile<T, OP, DIM, 1, false>(res, arg, tau_diag, tau, invert_tau, count, body, count - body, coeffs, lds_bytes, stream);
    return true;
  }
}

/// what both EvalLocal specialisations do (prox_elem_operation.inl:96-198): grid over the element groups, dynamic LDS
/// for the operation's per-thread scratch, launch, error check -- without the device synchronisation
template <typename T, class OP>
void Launch(T* res, const T* arg, const T* tau_diag, const StepArg<T>& tau, bool invert_tau, size_t count, size_t dim, bool interleaved,
            const ElemOpCoefficients<T, OP>& coeffs) {
  if (count == 0) return;
  hipStream_t stream = static_cast<hipStream_t>(CurrentStream());
  typename OP::GetSharedMemCount get_shared_mem_count;
  const size_t lds_bytes = get_shared_mem_count(dim) * kLanes * sizeof(typename OP::SharedMemType);
  if (lds_bytes > 160 * 1024) throw Exception("ProxElemOperation: the operation asks for more than the 160 KiB of LDS a workgroup can have.");
  bool done = false;
  if (!OP::kPartialResult && lds_bytes <= 64 * 1024) {
    done = TryTile<T, OP, 1>(dim, interleaved, res, arg, tau_diag, tau, invert_tau, count, coeffs, lds_bytes, stream) ||
           TryTile<T, OP, 2>(dim, interleaved, res, arg, tau_diag, tau, invert_tau, count, coeffs, lds_bytes, stream) ||
           TryTile<T, OP, 3>(dim, interleaved, res, arg, tau_diag, tau, invert_tau, count, coeffs, lds_bytes, stream) ||
           TryTile<T, OP, 4>(dim, interleaved, res, arg, tau_diag, tau, invert_tau, count, coeffs, lds_bytes, stream) ||
           TryTile<T, OP, 5>(dim, interleaved, res, arg, tau_diag, tau, invert_tau, count, coeffs, lds_bytes, stream) ||
           TryTile<T, OP, 6>(dim, interleaved, res, arg, tau_diag, tau, invert_tau, count, coeffs, lds_bytes, stream) ||
           TryTile<T, OP, 7>(dim, interleaved, res, arg, tau_diag, tau, invert_tau, count, coeffs, lds_bytes, stream) ||
           TryTile<T, OP, 8>(dim, interleaved, res, arg, tau_diag, tau, invert_tau, count, coeffs, lds_bytes, stream);
  }
  if (!done) {
    auto kernel = ProxElemOperationKernel<T, OP>;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
      throw Exception("ProxElemOperation: cannot reserve the LDS the operation asks for.");
    hipLaunchKernelGGL(kernel, dim3((unsigned)((count + kLanes - 1) / kLanes)), dim3(kLanes), lds_bytes, stream, res, arg, tau_diag, tau,
                       invert_tau, count, dim, coeffs, interleaved);
  }
  CheckLaunch("ProxElemOperationKernel");
}

}  // namespace elemop_kernel

// ---- kCoeffsCount == 0 (prox_elem_operation.inl:96-140) ----
template <typename T, class ELEM_OPERATION>
void ProxElemOperation<T, ELEM_OPERATION, typename std::enable_if<ELEM_OPERATION::kCoeffsCount == 0>::type>::EvalLocal(
    T* result_beg, T* /*result_end*/, const T* arg_beg, const T* /*arg_end*/, const T* tau_beg, const T* /*tau_end*/, T tau, bool invert_tau) {
  ElemOpCoefficients<T, ELEM_OPERATION> coeffs;
  coeffs.dev_p[0] = nullptr;
  coeffs.val[0] = 0;
  elemop_kernel::Launch<T, ELEM_OPERATION>(result_beg, arg_beg, tau_beg, elemop_kernel::StepArg<T>{tau, nullptr, nullptr}, invert_tau, this->count_, this->dim_,
                                           this->interleaved_, coeffs);
}
template <typename T, class ELEM_OPERATION>
void ProxElemOperation<T, ELEM_OPERATION, typename std::enable_if<ELEM_OPERATION::kCoeffsCount == 0>::type>::EvalLocalStepView(
    T* result_beg, T* /*result_end*/, const T* arg_beg, const T* /*arg_end*/, const T* tau_beg, const T* /*tau_end*/, const typename Prox<T>::StepView& view,
    bool invert_tau) {
  ElemOpCoefficients<T, ELEM_OPERATION> coeffs;
  coeffs.dev_p[0] = nullptr;
  coeffs.val[0] = 0;
  elemop_kernel::Launch<T, ELEM_OPERATION>(result_beg, arg_beg, tau_beg, elemop_kernel::StepArg<T>{(T)0, view.step, view.stop}, invert_tau, this->count_, this->dim_,
                                           this->interleaved_, coeffs);
}

// ---- kCoeffsCount != 0 (prox_elem_operation.inl:142-222) ----
template <typename T, class ELEM_OPERATION>
void ProxElemOperation<T, ELEM_OPERATION, typename std::enable_if<ELEM_OPERATION::kCoeffsCount != 0>::type>::KernelCoefficients(
    ElemOpCoefficients<T, ELEM_OPERATION>& coeffs) const {
  for (size_t i = 0; i < ELEM_OPERATION::kCoeffsCount; i++) {
    if (coeffs_[i].size() > 1) {
      if (d_coeffs_[i].size() != coeffs_[i].size()) throw Exception("ProxElemOperation used before Initialize().");
      coeffs.dev_p[i] = d_coeffs_[i].data();
      coeffs.val[i] = 0;
    } else {
      if (coeffs_[i].empty()) throw Exception("ProxElemOperation: empty coefficient vector.");
      coeffs.dev_p[i] = nullptr;
      coeffs.val[i] = coeffs_[i][0];
    }
  }
}
template <typename T, class ELEM_OPERATION>
void ProxElemOperation<T, ELEM_OPERATION, typename std::enable_if<ELEM_OPERATION::kCoeffsCount != 0>::type>::EvalLocal(
    T* result_beg, T* /*result_end*/, const T* arg_beg, const T* /*arg_end*/, const T* tau_beg, const T* /*tau_end*/, T tau, bool invert_tau) {
  ElemOpCoefficients<T, ELEM_OPERATION> coeffs;
  KernelCoefficients(coeffs);
  elemop_kernel::Launch<T, ELEM_OPERATION>(result_beg, arg_beg, tau_beg, elemop_kernel::StepArg<T>{tau, nullptr, nullptr}, invert_tau, this->count_, this->dim_,
                                           this->interleaved_, coeffs);
}
template <typename T, class ELEM_OPERATION>
void ProxElemOperation<T, ELEM_OPERATION, typename std::enable_if<ELEM_OPERATION::kCoeffsCount != 0>::type>::EvalLocalStepView(
    T* result_beg, T* /*result_end*/, const T* arg_beg, const T* /*arg_end*/, const T* tau_beg, const T* /*tau_end*/, const typename Prox<T>::StepView& view,
    bool invert_tau) {
  ElemOpCoefficients<T, ELEM_OPERATION> coeffs;
  KernelCoefficients(coeffs);
  elemop_kernel::Launch<T, ELEM_OPERATION>(result_beg, arg_beg, tau_beg, elemop_kernel::StepArg<T>{(T)0, view.step, view.stop}, invert_tau, this->count_, this->dim_,
                                           this->interleaved_, coeffs);
}

template <typename T, class ELEM_OPERATION>
void ProxElemOperation<T, ELEM_OPERATION, typename std::enable_if<ELEM_OPERATION::kCoeffsCount != 0>::type>::Initialize() {
  for (size_t i = 0; i < ELEM_OPERATION::kCoeffsCount; i++) {
    if (coeffs_[i].size() > 1) {
      if (coeffs_[i].size() < this->count_) throw Exception("Size of coefficients should be either 1 or count.");
      d_coeffs_[i] = coeffs_[i];        // device_vector upload; throws prost::Exception when HBM runs out
    }
  }
}

template <typename T, class ELEM_OPERATION>
void ProxElemOperation<T, ELEM_OPERATION, typename std::enable_if<ELEM_OPERATION::kCoeffsCount != 0>::type>::Release() {
  for (size_t i = 0; i < ELEM_OPERATION::kCoeffsCount; i++) d_coeffs_[i].clear();
}

}  // namespace prost
#endif
