// pdmp_consume.hip -- what callers do next with the chains, on the device: path integrals at probe coordinates (ESS estimators on the
// host see N x B x 32 numbers instead of N x d records) and, below, the streaming trace consumers (discretize / mean of src/trace.jl).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pdmp_engine.hpp"

namespace pdmp {

// J_i(T) = ∫_{t0}^{T} x_i(s) ds of every chain at `nprobe` coordinates (the integrand of mean(trace), src/trace.jl:182-200): the record
// carries the integral up to the coordinate's own clock and the linear piece from there (first sector of ZzRec and TrRec alike).
__global__ __launch_bounds__(256) void zz_path_integrals_kernel(const ZzRec* rec0, int64_t rec_stride, int64_t d, int64_t nchains,
                                                                const int64_t* __restrict__ probes, int64_t nprobe, double T,
                                                                double* __restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nchains * nprobe) return;
    const int64_t ch = k / nprobe, p = k - ch * nprobe;
    const int64_t i = probes[p];
    const ZzRec* r = reinterpret_cast<const ZzRec*>(reinterpret_cast<const char*>(rec0) + (ch * d + i) * rec_stride);
    const double dt = T - r->t;
    out[k] = r->I + dt * (r->x + r->th * (dt * 0.5));
}

int launch_zz_path_integrals(const ZzRec* rec, int64_t rec_stride, int64_t d, int64_t nchains, const int64_t* probes, int64_t nprobe,
                             double T, double* out, void* stream) {
    const int64_t n = nchains * nprobe;
    hipLaunchKernelGGL(zz_path_integrals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rec, rec_stride, d,
                       nchains, probes, nprobe, T, out);
    return (int)hipGetLastError();
}

}  // namespace pdmp
