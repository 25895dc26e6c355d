// Deterministic sums used around the kernels: column sums of a (rows, cols)
// matrix of per-workgroup partial results, and the total of a long vector (the
// (T,B) -> scalar reduction of the ELBO terms, kvae/misc.py:21-41).  Plain
// two-level tree sums, no atomics and no memset: the same bits on every run, and
// nothing in them needs a memset node when a step is captured into a HIP graph.
#include <hip/hip_runtime.h>

#include "../../include/kvae_kalman.h"
#include "launch_common.h"

namespace kvae {

constexpr int COLSUM_WAVES = 16;

// out[c] = sum_r in[r * cols + c].  Block = 16 wavefronts x 64 columns; wavefront w
// adds rows w, w+16, ... (four independent accumulators per lane).
template <typename T>
__global__ void __launch_bounds__(64 * COLSUM_WAVES)
column_sum_kernel(long rows, int cols, const T* __restrict__ in, T* __restrict__ out) {
    __shared__ T sm[COLSUM_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    T acc[4] = {T(0), T(0), T(0), T(0)};
    if (c < cols) {
        for (long r = wave; r < rows; r += 4 * COLSUM_WAVES) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long rr = r + (long)u * COLSUM_WAVES;
                if (rr < rows) acc[u] += in[rr * cols + c];
            }
        }
    }
    sm[wave][lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (wave == 0 && c < cols) {
        T s = T(0);
#pragma unroll
        for (int w = 0; w < COLSUM_WAVES; ++w) s += sm[w][lane];
        out[c] = s;
    }
}

// The same sums for up to four matrices in ONE launch (each launch of a dependent chain
// costs ~5 us whatever its size): workgroup b belongs to the matrix whose block range
// contains b.
constexpr int COLSUM_MAX_PARTS = 4;
template <typename T>
struct ColumnSumParts {
    long rows[COLSUM_MAX_PARTS];
    int cols[COLSUM_MAX_PARTS];
    int first_block[COLSUM_MAX_PARTS + 1];
    int wide[COLSUM_MAX_PARTS];     // 1: a lane sums four adjacent columns (16-byte loads)
    const T* in[COLSUM_MAX_PARTS];
    T* out[COLSUM_MAX_PARTS];
};
// Every column is summed in the same order in both forms (rows wave, wave + W, ... in four
// interleaved partial sums, then the wavefronts in order), so they give identical results.
template <typename T>
__global__ void __launch_bounds__(64 * COLSUM_WAVES)
column_sum_many_kernel(int parts, ColumnSumParts<T> p) {
    __shared__ T sm[COLSUM_WAVES][64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int k = 0;
#pragma unroll
    for (int i = 1; i < COLSUM_MAX_PARTS; ++i)
        if (i < parts && (int)blockIdx.x >= p.first_block[i]) k = i;
    const long rows = p.rows[k];
    const int cols = p.cols[k];
    const T* __restrict__ in = p.in[k];
    if (p.wide[k]) {                 // the same for the whole workgroup
        if constexpr (sizeof(T) == 4) {
            const int c = ((int)blockIdx.x - p.first_block[k]) * 256 + 4 * lane;
            float4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < cols) {
                for (long r = wave; r < rows; r += 4 * COLSUM_WAVES) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const long rr = r + (long)u * COLSUM_WAVES;
                        if (rr < rows) {
                            const float4 v = *reinterpret_cast<const float4*>(in + rr * cols + c);
                            acc[u].x += v.x;
                            acc[u].y += v.y;
                            acc[u].z += v.z;
                            acc[u].w += v.w;
                        }
                    }
                }
            }
            sm[wave][4 * lane + 0] = (acc[0].x + acc[1].x) + (acc[2].x + acc[3].x);
            sm[wave][4 * lane + 1] = (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y);
            sm[wave][4 * lane + 2] = (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z);
            sm[wave][4 * lane + 3] = (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w);
            __syncthreads();
            // 256 columns, one per lane of the first four wavefronts
            const int cc = threadIdx.x;
            const int col = ((int)blockIdx.x - p.first_block[k]) * 256 + cc;
            if (cc < 256 && col < cols) {
                T s = T(0);
#pragma unroll
                for (int w = 0; w < COLSUM_WAVES; ++w) s += sm[w][cc];
                p.out[k][col] = s;
            }
        }
        return;
    }
    const int c = ((int)blockIdx.x - p.first_block[k]) * 64 + lane;
    T acc[4] = {T(0), T(0), T(0), T(0)};
    if (c < cols) {
        for (long r = wave; r < rows; r += 4 * COLSUM_WAVES) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long rr = r + (long)u * COLSUM_WAVES;
                if (rr < rows) acc[u] += in[rr * cols + c];
            }
        }
    }
    sm[wave][lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (wave == 0 && c < cols) {
        T s = T(0);
#pragma unroll
        for (int w = 0; w < COLSUM_WAVES; ++w) s += sm[w][lane];
        p.out[k][c] = s;
    }
}

template <typename T>
int launch_column_sum_many(int parts, const long* rows, const int* cols, const void* const* in,
                           void* const* out, hipStream_t s) {
    ColumnSumParts<T> p = {};
    int blocks = 0;
    for (int i = 0; i < parts; ++i) {
        if (rows[i] < 0 || cols[i] < 1 || !in[i] || !out[i]) return KVAE_E_BADARG;
        p.rows[i] = rows[i];
        p.cols[i] = cols[i];
        p.first_block[i] = blocks;
        p.in[i] = (const T*)in[i];
        p.out[i] = (T*)out[i];
        // many columns, rows that start on 16-byte boundaries: four columns per lane
        p.wide[i] = sizeof(T) == 4 && cols[i] >= 1024 && cols[i] % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(in[i]) & 15) == 0;
        blocks += p.wide[i] ? (cols[i] + 255) / 256 : (cols[i] + 63) / 64;
    }
    p.first_block[parts] = blocks;
    hipLaunchKernelGGL((column_sum_many_kernel<T>), dim3(blocks), dim3(64 * COLSUM_WAVES), 0, s,
                       parts, p);
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

constexpr int VECSUM_BLOCK = 1024;
constexpr long VECSUM_CHUNK = 8192;       // elements per workgroup of the first level
constexpr int VECSUM_MAX_CHUNKS = 1024;

// out[b * gridDim.x + blockIdx.x] = sum of in[b * n + lo .. hi), lo = blockIdx.x * chunk
template <typename T>
__global__ void __launch_bounds__(VECSUM_BLOCK)
chunk_sum_kernel(long n, long chunk, const T* __restrict__ in, T* __restrict__ out) {
    __shared__ T sm[VECSUM_BLOCK / 64];
    const long lo = (long)blockIdx.x * chunk;
    const long hi = lo + chunk < n ? lo + chunk : n;
    const T* v = in + (long)blockIdx.y * n;
    T acc[4] = {T(0), T(0), T(0), T(0)};
    for (long i = lo + threadIdx.x; i < hi; i += 4 * VECSUM_BLOCK) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long ii = i + (long)u * VECSUM_BLOCK;
            if (ii < hi) acc[u] += v[ii];
        }
    }
    T s = wave_sum<T>((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        T tot = T(0);
#pragma unroll
        for (int w = 0; w < VECSUM_BLOCK / 64; ++w) tot += sm[w];
        out[(long)blockIdx.y * gridDim.x + blockIdx.x] = tot;
    }
}

inline long vecsum_chunks(long n) {
    long chunks = (n + VECSUM_CHUNK - 1) / VECSUM_CHUNK;
    return chunks > VECSUM_MAX_CHUNKS ? VECSUM_MAX_CHUNKS : (chunks < 1 ? 1 : chunks);
}

template <typename T>
void launch_vector_sum(int batch, long n, const T* in, T* scratch, T* out, hipStream_t s) {
    const long chunks = vecsum_chunks(n);
    if (chunks == 1) {
        hipLaunchKernelGGL((chunk_sum_kernel<T>), dim3(1, batch), dim3(VECSUM_BLOCK), 0, s, n, n, in,
                           out);
        return;
    }
    const long chunk = (n + chunks - 1) / chunks;
    hipLaunchKernelGGL((chunk_sum_kernel<T>), dim3((unsigned)chunks, batch), dim3(VECSUM_BLOCK), 0, s,
                       n, chunk, in, scratch);
    hipLaunchKernelGGL((chunk_sum_kernel<T>), dim3(1, batch), dim3(VECSUM_BLOCK), 0, s, chunks,
                       chunks, (const T*)scratch, out);
}

// Second level of the three-term total: out[k] = w[k] * (sum of partial[k][0..chunks)),
// out[3] = out[0] + out[1] + out[2]  (the aggregate() of the three ELBO terms and the
// weighted objective kalman_weight * (obs + trans - post), kvae/kalman_vae.py:244-260).
struct TermWeights {
    double w[3];
};
template <typename T>
__global__ void __launch_bounds__(VECSUM_BLOCK)
weighted_terms_kernel(long chunks, const T* __restrict__ partial, TermWeights tw,
                      T* __restrict__ out) {
    __shared__ T sm[3][VECSUM_BLOCK / 64];
    for (int k = 0; k < 3; ++k) {
        T acc = T(0);
        for (long i = threadIdx.x; i < chunks; i += VECSUM_BLOCK) acc += partial[k * chunks + i];
        acc = wave_sum<T>(acc);
        if ((threadIdx.x & 63) == 0) sm[k][threadIdx.x >> 6] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        T total = T(0);
        for (int k = 0; k < 3; ++k) {
            T t = T(0);
#pragma unroll
            for (int w = 0; w < VECSUM_BLOCK / 64; ++w) t += sm[k][w];
            t *= (T)tw.w[k];
            out[k] = t;
            total += t;
        }
        out[3] = total;
    }
}

template <typename T>
void launch_weighted_terms(long n, const T* terms, T* scratch, const double* w, T* out,
                           hipStream_t s) {
    const long chunks = vecsum_chunks(n);
    const long chunk = (n + chunks - 1) / chunks;
    hipLaunchKernelGGL((chunk_sum_kernel<T>), dim3((unsigned)chunks, 3), dim3(VECSUM_BLOCK), 0, s, n,
                       chunk, terms, scratch);
    TermWeights tw;
    for (int k = 0; k < 3; ++k) tw.w[k] = w[k];
    hipLaunchKernelGGL((weighted_terms_kernel<T>), dim3(1), dim3(VECSUM_BLOCK), 0, s, chunks,
                       (const T*)scratch, tw, out);
}

}  // namespace kvae

extern "C" {

int kvae_column_sum(long rows, int cols, int dtype, const void* in, void* out, void* stream) {
    if (rows < 0 || cols < 1 || !in || !out) return KVAE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((cols + 63) / 64), block(64 * kvae::COLSUM_WAVES);
    // many float columns: the four-columns-per-lane form of the multi-matrix kernel (the
    // same summation order, hence the same bits)
    if (dtype == KVAE_F32 && cols >= 1024 && cols % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(in) & 15) == 0)
        return kvae::launch_column_sum_many<float>(1, &rows, &cols, &in, &out, s);
    if (dtype == KVAE_F32)
        hipLaunchKernelGGL((kvae::column_sum_kernel<float>), grid, block, 0, s, rows, cols,
                           (const float*)in, (float*)out);
    else if (dtype == KVAE_F64)
        hipLaunchKernelGGL((kvae::column_sum_kernel<double>), grid, block, 0, s, rows, cols,
                           (const double*)in, (double*)out);
    else
        return KVAE_E_BADARG;
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

int kvae_column_sum_many(int parts, const long* rows, const int* cols, int dtype,
                         const void* const* in, void* const* out, void* stream) {
    if (parts < 1 || parts > kvae::COLSUM_MAX_PARTS || !rows || !cols || !in || !out)
        return KVAE_E_BADARG;
    if (dtype == KVAE_F32)
        return kvae::launch_column_sum_many<float>(parts, rows, cols, in, out, (hipStream_t)stream);
    if (dtype == KVAE_F64)
        return kvae::launch_column_sum_many<double>(parts, rows, cols, in, out, (hipStream_t)stream);
    return KVAE_E_BADARG;
}

long kvae_vector_sum_scratch_elems(int batch, long n) {
    if (batch < 1 || n < 1) return 0;
    const long chunks = kvae::vecsum_chunks(n);
    return chunks == 1 ? 0 : chunks * batch;
}

int kvae_vector_sum(int batch, long n, int dtype, const void* in, void* scratch, void* out,
                    void* stream) {
    if (batch < 1 || batch > 65535 || n < 1 || !in || !out) return KVAE_E_BADARG;
    if (kvae::vecsum_chunks(n) > 1 && !scratch) return KVAE_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KVAE_F32)
        kvae::launch_vector_sum<float>(batch, n, (const float*)in, (float*)scratch, (float*)out, s);
    else if (dtype == KVAE_F64)
        kvae::launch_vector_sum<double>(batch, n, (const double*)in, (double*)scratch, (double*)out,
                                        s);
    else
        return KVAE_E_BADARG;
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

long kvae_weighted_terms_scratch_elems(long n) { return n < 1 ? 0 : 3 * kvae::vecsum_chunks(n); }

int kvae_weighted_terms(long n, int dtype, const void* terms, void* scratch, double w_obs,
                        double w_trans, double w_post, void* out, void* stream) {
    if (n < 1 || !terms || !scratch || !out) return KVAE_E_BADARG;
    const double w[3] = {w_obs, w_trans, w_post};
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KVAE_F32)
        kvae::launch_weighted_terms<float>(n, (const float*)terms, (float*)scratch, w, (float*)out, s);
    else if (dtype == KVAE_F64)
        kvae::launch_weighted_terms<double>(n, (const double*)terms, (double*)scratch, w,
                                            (double*)out, s);
    else
        return KVAE_E_BADARG;
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

}  // extern "C"
