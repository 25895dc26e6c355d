// gfx950 kernels of the wavefront-per-sequence variant (kalman_wave.h): one
// 64-thread workgroup per sequence, all matrices of the sequence in LDS.
// Used for state dimensions too large for the lane-per-sequence kernels
// (z_dim = 32: a covariance is 4 KiB, the filter's working set ~30 KiB).
//
// Grid = B workgroups.  Parameter-gradient sums are kept in registers
// (distributed over the lanes), written as one row per sequence and summed by
// reduce_rows_kernel in a fixed order (no float atomics).
#include "wave_api.h"

#include "kalman_wave.h"
#include "launch_common.h"

namespace kvae {

constexpr size_t LDS_LIMIT = 160 * 1024;

template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(64) wave_filter_forward_kernel(FilterFwdArgs<R> g) {
    __shared__ wave::FilterScratch<R, N, P, K> s;
    wave::filter_forward_sequence<R, N, P, K>(g, blockIdx.x, s);
}

template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(64) wave_filter_backward_kernel(FilterBwdArgs<R> g) {
    __shared__ wave::FilterBwdScratch<R, N, P, K> s;
    wave::FilterGradAcc<R, N, P, K> acc;
    acc.zero();
    wave::filter_backward_sequence<R, N, P, K>(g, blockIdx.x, s, acc);
    acc.store(g.partial + (long)blockIdx.x * wave::FilterGradAcc<R, N, P, K>::LEN);
}

// ---- mixing adjoint, second pass of the filter backward ------------------------
// (arithmetic defined by wave::mix_adjoint_host)
constexpr int MIX_ROWS_PER_CHUNK = 1024;

// d_alpha[r][k] = <bar_As[r], A^k> + <bar_Cs[r], C^k>.  One wavefront per row
// (grid-stride), bases staged once per workgroup in LDS.
template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(256) mix_dalpha_kernel(long rows, const R* bar_As,
                                                        const R* bar_Cs, const R* A_K,
                                                        const R* C_K, R* d_alpha) {
    constexpr int LA = N * N, LC = P * N;
    __shared__ R bases[K * (LA + LC)];
    for (int i = threadIdx.x; i < K * LA; i += 256) bases[i] = A_K[i];
    for (int i = threadIdx.x; i < K * LC; i += 256) bases[K * LA + i] = C_K[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    if constexpr (sizeof(R) * K * (LA + LC) / 64 <= 768 && LA % 256 == 0 && LC % 256 == 0) {
        // the bases fit the registers (K (LA + LC) / 64 values per lane: 192 floats at n = 32,
        // p = 16, K = 8): a lane keeps the basis elements of the 16-byte pieces it reads, and
        // the loop over rows is loads and multiply-adds only (from LDS it was eight LDS reads
        // per element loaded, 2.0 TB/s: 3.1 -> 1.15 ms at a million rows)
        typedef R vec4 __attribute__((ext_vector_type(4)));
        constexpr int QA = LA / 256, QC = LC / 256;          // 16-byte pieces per lane and row
        vec4 bA[K][QA], bC[K][QC];
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int q = 0; q < QA; ++q)
                bA[k][q] = *reinterpret_cast<const vec4*>(&bases[k * LA + 4 * (lane + 64 * q)]);
#pragma unroll
            for (int q = 0; q < QC; ++q)
                bC[k][q] = *reinterpret_cast<const vec4*>(&bases[K * LA + k * LC + 4 * (lane + 64 * q)]);
        }
        for (long r = wave0; r < rows; r += nwaves) {
            vec4 xa[QA], xc[QC];
#pragma unroll
            for (int q = 0; q < QA; ++q)
                xa[q] = *reinterpret_cast<const vec4*>(bar_As + r * LA + 4 * (lane + 64 * q));
#pragma unroll
            for (int q = 0; q < QC; ++q)
                xc[q] = *reinterpret_cast<const vec4*>(bar_Cs + r * LC + 4 * (lane + 64 * q));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                R acc = R(0);
#pragma unroll
                for (int q = 0; q < QA; ++q)
                    acc += xa[q].x * bA[k][q].x + xa[q].y * bA[k][q].y + xa[q].z * bA[k][q].z + xa[q].w * bA[k][q].w;
#pragma unroll
                for (int q = 0; q < QC; ++q)
                    acc += xc[q].x * bC[k][q].x + xc[q].y * bC[k][q].y + xc[q].z * bC[k][q].z + xc[q].w * bC[k][q].w;
                const R v = wave_sum<R>(acc);
                if (lane == 0) d_alpha[r * K + k] = v;
            }
        }
        return;
    }
    for (long r = wave0; r < rows; r += nwaves) {
        R acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = R(0);
        for (int e = lane; e < LA; e += 64) {
            const R x = bar_As[r * LA + e];
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] += x * bases[k * LA + e];
        }
        for (int e = lane; e < LC; e += 64) {
            const R x = bar_Cs[r * LC + e];
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] += x * bases[K * LA + k * LC + e];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const R v = wave_sum<R>(acc[k]);
            if (lane == 0) d_alpha[r * K + k] = v;
        }
    }
}

// partial[chunk][k*LA + e] = sum_{r in chunk} alpha[r][k] bar_As[r][e] (and the
// same for C behind it).  Thread = one column, workgroup = 256 columns x one
// chunk of rows; every load of the bar_* stream is a fully coalesced row piece.
template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(256) mix_dbasis_kernel(long rows, const R* alpha,
                                                        const R* bar_As, const R* bar_Cs,
                                                        R* partial) {
    constexpr int LA = N * N, LC = P * N;
    const long r0 = (long)blockIdx.y * MIX_ROWS_PER_CHUNK;
    const long r1 = r0 + MIX_ROWS_PER_CHUNK < rows ? r0 + MIX_ROWS_PER_CHUNK : rows;
    if constexpr (LA % 1024 == 0 && LC % 4 == 0 && sizeof(R) == 4) {
        // large tiles: four adjacent columns per thread, 16-byte loads (1.8 -> 1.15 ms at a million rows; gridDim.x covers
        // (LA + LC) / 4 threads; a quad of columns lies in one of the two tensors)
        typedef R vec4 __attribute__((ext_vector_type(4)));
        const int c4 = 4 * (blockIdx.x * 256 + threadIdx.x);
        if (c4 >= LA + LC) return;
        const bool inA = c4 < LA;
        const R* src4 = inA ? bar_As + c4 : bar_Cs + (c4 - LA);
        const long stride4 = inA ? LA : LC;
        vec4 acc4[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc4[k] = vec4{R(0), R(0), R(0), R(0)};
        for (long r = r0; r < r1; ++r) {
            const vec4 x = *reinterpret_cast<const vec4*>(src4 + r * stride4);
#pragma unroll
            for (int k = 0; k < K; ++k) acc4[k] += alpha[r * K + k] * x;
        }
        R* out4 = partial + (long)blockIdx.y * (K * (LA + LC));
#pragma unroll
        for (int k = 0; k < K; ++k) {
            R* dst = inA ? out4 + k * LA + c4 : out4 + K * LA + k * LC + (c4 - LA);
            *reinterpret_cast<vec4*>(dst) = acc4[k];
        }
        return;
    }
    const int c = blockIdx.x * 256 + threadIdx.x;  // column of [bar_As | bar_Cs]
    if (c >= LA + LC) return;
    const bool isA = c < LA;
    const R* src = isA ? bar_As + c : bar_Cs + (c - LA);
    const long stride = isA ? LA : LC;
    R acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = R(0);
    for (long r = r0; r < r1; ++r) {
        const R x = src[r * stride];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += alpha[r * K + k] * x;
    }
    R* out = partial + (long)blockIdx.y * (K * (LA + LC));
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (isA) {
            out[k * LA + c] = acc[k];
        } else {
            out[K * LA + k * LC + (c - LA)] = acc[k];
        }
    }
}

template <typename R, int N, int P>
__global__ void __launch_bounds__(64) wave_smooth_forward_kernel(SmoothFwdArgs<R> g) {
    __shared__ wave::SmoothScratch<R, N, P> s;
    wave::smooth_forward_sequence<R, N, P>(g, blockIdx.x, s);
}

template <typename R, int N, int P>
__global__ void __launch_bounds__(64) wave_smooth_backward_kernel(SmoothBwdArgs<R> g) {
    __shared__ wave::SmoothBwdScratch<R, N, P> s;
    wave::Acc<R, P * P> acc;
    acc.zero();
    wave::smooth_backward_sequence<R, N, P>(g, blockIdx.x, s, acc);
    if (g.partial) acc.store(g.partial + (long)blockIdx.x * (P * P));
}

template <typename R, int N, int P>
__global__ void __launch_bounds__(64) wave_elbo_forward_kernel(ElboFwdArgs<R> g) {
    __shared__ wave::ElboScratch<R, N, P> s;
    wave::elbo_forward_sequence<R, N, P>(g, blockIdx.x, s);
}

template <typename R, int N, int P>
__global__ void __launch_bounds__(64) wave_elbo_backward_kernel(ElboBwdArgs<R> g) {
    __shared__ wave::ElboBwdScratch<R, N, P> s;
    wave::ElboGradAcc<R, N, P> acc;
    acc.zero();
    wave::elbo_backward_sequence<R, N, P>(g, blockIdx.x, s, acc);
    acc.store(g.partial + (long)blockIdx.x * ElboParamGrads<N, P>::LEN);
}

// ---- launchers ---------------------------------------------------------------
template <typename R, int N, int P, int K>
int run_wave_filter_forward(const kvae_dims* d, const kvae_filter_fwd* io, hipStream_t s) {
    if constexpr (sizeof(wave::FilterScratch<R, N, P, K>) > LDS_LIMIT) {
        return KVAE_E_UNSUPPORTED;
    } else {
        FilterFwdArgs<R> g = make_args<R>(d, io);
        hipLaunchKernelGGL((wave_filter_forward_kernel<R, N, P, K>), dim3(d->B), dim3(64), 0, s, g);
        return launch_status();
    }
}

template <typename R, int N, int P, int K>
int run_wave_filter_backward(const kvae_dims* d, const kvae_filter_bwd* io, void* ws,
                             size_t ws_bytes, hipStream_t s) {
    if constexpr (sizeof(wave::FilterBwdScratch<R, N, P, K>) > LDS_LIMIT) {
        return KVAE_E_UNSUPPORTED;
    } else {
        constexpr int LA = N * N, LC = P * N, QR = N * N + P * P;
        const long rows = (long)(d->T + 1) * d->B;
        const long chunks = (rows + MIX_ROWS_PER_CHUNK - 1) / MIX_ROWS_PER_CHUNK;
        const size_t need = ((size_t)rows * (LA + LC) + (size_t)d->B * QR +
                             (size_t)chunks * K * (LA + LC)) * sizeof(R);
        if (ws_bytes < need) return KVAE_E_WORKSPACE;
        R* bar_As = (R*)ws;
        R* bar_Cs = bar_As + rows * LA;
        R* qr_partial = bar_Cs + rows * LC;
        R* mix_partial = qr_partial + (long)d->B * QR;
        FilterBwdArgs<R> g = make_args<R>(d, io, qr_partial);
        g.bar_As = bar_As;
        g.bar_Cs = bar_Cs;
        bool launched = false;
        if constexpr (sizeof(R) == 4) launched = tile_filter_backward_main(d, g, s) == KVAE_OK;
        if (!launched)
            hipLaunchKernelGGL((wave_filter_backward_kernel<R, N, P, K>), dim3(d->B), dim3(64), 0, s,
                               g);
        {   // d_Q, d_R: sum the per-sequence rows
            void* outs[2] = {io->d_Q, io->d_R};
            const int begin[3] = {0, N * N, QR};
            launch_reduce<R>(qr_partial, d->B, QR, 2, outs, begin, s);
        }
        const long dalpha_blocks = rows / 4 + 1 < 2048 ? rows / 4 + 1 : 2048;
        hipLaunchKernelGGL((mix_dalpha_kernel<R, N, P, K>), dim3(dalpha_blocks), dim3(256), 0, s,
                           rows, (const R*)bar_As, (const R*)bar_Cs, g.f.A_K, g.f.C_K, g.d_alpha);
        hipLaunchKernelGGL((mix_dbasis_kernel<R, N, P, K>),
                           dim3((LA + LC + 255) / 256, chunks), dim3(256), 0, s, rows, g.f.alpha,
                           (const R*)bar_As, (const R*)bar_Cs, mix_partial);
        {   // d_A_K, d_C_K: sum the per-chunk rows
            void* outs[2] = {io->d_A_K, io->d_C_K};
            const int begin[3] = {0, K * LA, K * (LA + LC)};
            launch_reduce<R>(mix_partial, chunks, K * (LA + LC), 2, outs, begin, s);
        }
        return launch_status();
    }
}

template <typename R, int N, int P>
int run_wave_smooth_forward(const kvae_dims* d, const kvae_smooth_fwd* io, hipStream_t s) {
    if constexpr (sizeof(wave::SmoothScratch<R, N, P>) > LDS_LIMIT) {
        return KVAE_E_UNSUPPORTED;
    } else {
        SmoothFwdArgs<R> g = make_args<R>(d, io);
        hipLaunchKernelGGL((wave_smooth_forward_kernel<R, N, P>), dim3(d->B), dim3(64), 0, s, g);
        return launch_status();
    }
}

template <typename R, int N, int P>
int run_wave_smooth_backward(const kvae_dims* d, const kvae_smooth_bwd* io, void* ws,
                             size_t ws_bytes, hipStream_t s) {
    if constexpr (sizeof(wave::SmoothBwdScratch<R, N, P>) > LDS_LIMIT) {
        return KVAE_E_UNSUPPORTED;
    } else {
        if (io->d_LR && ws_bytes < (size_t)d->B * P * P * sizeof(R)) return KVAE_E_WORKSPACE;
        SmoothBwdArgs<R> g = make_args<R>(d, io, ws);
        hipLaunchKernelGGL((wave_smooth_backward_kernel<R, N, P>), dim3(d->B), dim3(64), 0, s, g);
        if (io->d_LR) reduce_smooth_grads<R, P>(io, ws, d->B, s);
        return launch_status();
    }
}

template <typename R, int N, int P>
int run_wave_elbo_forward(const kvae_dims* d, const kvae_elbo_fwd* io, hipStream_t s) {
    if constexpr (sizeof(wave::ElboScratch<R, N, P>) > LDS_LIMIT) {
        return KVAE_E_UNSUPPORTED;
    } else {
        ElboFwdArgs<R> g = make_args<R>(d, io);
        hipLaunchKernelGGL((wave_elbo_forward_kernel<R, N, P>), dim3(d->B), dim3(64), 0, s, g);
        return launch_status();
    }
}

template <typename R, int N, int P>
int run_wave_elbo_backward(const kvae_dims* d, const kvae_elbo_bwd* io, void* ws, size_t ws_bytes,
                           hipStream_t s) {
    if constexpr (sizeof(wave::ElboBwdScratch<R, N, P>) > LDS_LIMIT) {
        return KVAE_E_UNSUPPORTED;
    } else {
        using PG = ElboParamGrads<N, P>;
        if (ws_bytes < (size_t)d->B * PG::LEN * sizeof(R)) return KVAE_E_WORKSPACE;
        ElboBwdArgs<R> g = make_args<R>(d, io, ws);
        hipLaunchKernelGGL((wave_elbo_backward_kernel<R, N, P>), dim3(d->B), dim3(64), 0, s, g);
        reduce_elbo_grads<R, N, P>(io, ws, d->B, s);
        return launch_status();
    }
}

// (n, p, K) compiled for the wavefront-per-sequence path
#define KVAE_WAVE_DIM_LIST(X) \
    X(32, 16, 8)              \
    X(16, 8, 4)               \
    X(6, 3, 2)

// all six scratch structs of a combination must fit the LDS for it to be usable
template <typename R, int N, int P, int K>
constexpr bool wave_fits() {
    return sizeof(wave::FilterBwdScratch<R, N, P, K>) <= LDS_LIMIT &&
           sizeof(wave::SmoothBwdScratch<R, N, P>) <= LDS_LIMIT &&
           sizeof(wave::ElboBwdScratch<R, N, P>) <= LDS_LIMIT;
}

#define KVAE_WAVE_CASE(n_, p_, K_)                                     \
    if (d->n == n_ && d->p == p_ && d->K == K_) {                      \
        if (d->dtype == KVAE_F32) return KVAE_CALL(float, n_, p_, K_); \
        return KVAE_CALL(double, n_, p_, K_);                          \
    }

int wave_supported(int n, int p, int K, int dtype) {
#define KVAE_SUP(n_, p_, K_)                                                        \
    if (n == n_ && p == p_ && K == K_)                                              \
        return dtype == KVAE_F32 ? wave_fits<float, n_, p_, K_>()                   \
                                 : (dtype == KVAE_F64 ? wave_fits<double, n_, p_, K_>() : 0);
    KVAE_WAVE_DIM_LIST(KVAE_SUP)
#undef KVAE_SUP
    return 0;
}

size_t wave_workspace_bytes(const kvae_dims* d, int op) {
    const size_t el = d->dtype == KVAE_F32 ? 4 : 8;
    const long n = d->n, p = d->p, K = d->K;
    switch (op) {
        case KVAE_OP_FILTER_BWD: {
            const size_t rows = (size_t)(d->T + 1) * d->B;
            const size_t chunks = (rows + MIX_ROWS_PER_CHUNK - 1) / MIX_ROWS_PER_CHUNK;
            return (rows * (n * n + p * n) + (size_t)d->B * (n * n + p * p) +
                    chunks * K * (n * n + p * n)) * el;
        }
        case KVAE_OP_SMOOTH_BWD:
            return (size_t)d->B * (p * p) * el;
        case KVAE_OP_ELBO_BWD:   // a row per sequence and piece (tile kernels: tile_elbo_backward)
            return (size_t)d->B * tile_elbo_backward_pieces(d) * (n * n + p * p + n + n * n) * el;
        default:
            return 0;
    }
}

int wave_filter_forward(const kvae_dims* d, const kvae_filter_fwd* io, hipStream_t s) {
    if (tile_supported(d->n, d->p, d->dtype)) {   // e.g. refuses unaligned operands
        const int rc = tile_filter_forward(d, io, s);
        if (rc != KVAE_E_UNSUPPORTED) return rc;
    }
#define KVAE_CALL(R, n_, p_, K_) run_wave_filter_forward<R, n_, p_, K_>(d, io, s)
    KVAE_WAVE_DIM_LIST(KVAE_WAVE_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}
int wave_filter_backward(const kvae_dims* d, const kvae_filter_bwd* io, void* ws, size_t ws_bytes,
                         hipStream_t s) {
#define KVAE_CALL(R, n_, p_, K_) run_wave_filter_backward<R, n_, p_, K_>(d, io, ws, ws_bytes, s)
    KVAE_WAVE_DIM_LIST(KVAE_WAVE_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}
int wave_smooth_forward(const kvae_dims* d, const kvae_smooth_fwd* io, hipStream_t s) {
    if (tile_supported(d->n, d->p, d->dtype)) {   // e.g. refuses unaligned operands
        const int rc = tile_smooth_forward(d, io, s);
        if (rc != KVAE_E_UNSUPPORTED) return rc;
    }
#define KVAE_CALL(R, n_, p_, K_) run_wave_smooth_forward<R, n_, p_>(d, io, s)
    KVAE_WAVE_DIM_LIST(KVAE_WAVE_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}
int wave_smooth_backward(const kvae_dims* d, const kvae_smooth_bwd* io, void* ws, size_t ws_bytes,
                         hipStream_t s) {
    if (tile_supported(d->n, d->p, d->dtype)) {   // e.g. refuses unaligned operands
        const int rc = tile_smooth_backward(d, io, ws, ws_bytes, s);
        if (rc != KVAE_E_UNSUPPORTED) return rc;
    }
#define KVAE_CALL(R, n_, p_, K_) run_wave_smooth_backward<R, n_, p_>(d, io, ws, ws_bytes, s)
    KVAE_WAVE_DIM_LIST(KVAE_WAVE_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}
int wave_elbo_forward(const kvae_dims* d, const kvae_elbo_fwd* io, hipStream_t s) {
    if (tile_supported(d->n, d->p, d->dtype)) {   // e.g. refuses unaligned operands
        const int rc = tile_elbo_forward(d, io, s);
        if (rc != KVAE_E_UNSUPPORTED) return rc;
    }
#define KVAE_CALL(R, n_, p_, K_) run_wave_elbo_forward<R, n_, p_>(d, io, s)
    KVAE_WAVE_DIM_LIST(KVAE_WAVE_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}
int wave_elbo_backward(const kvae_dims* d, const kvae_elbo_bwd* io, void* ws, size_t ws_bytes,
                       hipStream_t s) {
    if (tile_supported(d->n, d->p, d->dtype)) {   // e.g. refuses unaligned operands
        const int rc = tile_elbo_backward(d, io, ws, ws_bytes, s);
        if (rc != KVAE_E_UNSUPPORTED) return rc;
    }
#define KVAE_CALL(R, n_, p_, K_) run_wave_elbo_backward<R, n_, p_>(d, io, ws, ws_bytes, s)
    KVAE_WAVE_DIM_LIST(KVAE_WAVE_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}

}  // namespace kvae
