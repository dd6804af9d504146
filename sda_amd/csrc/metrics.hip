// Evaluation metrics of the sampling experiments (sda/utils.py:203-263: emd, mmd) -- SURVEY section 8(f)-4.
//
//   sda_pairwise_dist : D[i][j] = |x_i - y_j|^2  (or its square root), x: [m][d], y: [n][d] row-major.
//       The O(m n d) part of both metrics.  Differences are formed before squaring (no |x|^2+|y|^2-2xy cancellation), so
//       this is VALU work, not a GEMM: 64x64 output tile per workgroup, 4x4 outputs per lane, d staged through LDS in
//       chunks of 32 (each staged element is used 64 times).  Bound: VALU fp32 (3 flop per (i, j, k)).
//   sda_mmd_kernel_sums : sum over all entries of sum_sigma exp(-D/sigma), sigma in {1e-3 .. 1e3}, as one double per
//       workgroup (the caller adds the partials) -- the kernel means of utils.py:254-261 without the three m x n temporaries
//       per bandwidth.
//   sda_assignment_cost : HOST routine.  With uniform weights and equally many samples the optimal transport plan of
//       utils.py:203-219 (POT's network simplex, a CPU code in the reference as well) is a permutation, so the EMD is a
//       linear assignment problem: shortest augmenting paths with potentials (Jonker-Volgenant style), O(n^3), double
//       precision.  The cost matrix is computed on the device; only the n x n solve runs on the host.
#include "sda_common.hpp"

#include <vector>
#include <limits>

#define PD_TILE 64
#define PD_K 32

__global__ __launch_bounds__(256) void pairwise_dist_kernel(const float* __restrict__ x, int m, const float* __restrict__ y,
                                                            int n, int64_t d, int take_sqrt, float* __restrict__ out) {
    __shared__ float sx[PD_K][PD_TILE + 1];
    __shared__ float sy[PD_K][PD_TILE + 1];
    const int i0 = blockIdx.y * PD_TILE, j0 = blockIdx.x * PD_TILE;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // lane owns rows ty + 16 a, columns tx + 16 b
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (int64_t k0 = 0; k0 < d; k0 += PD_K) {
        // stage: 64 rows x 32 features of each operand; consecutive lanes read consecutive features of one row
        for (int e = threadIdx.x; e < PD_TILE * PD_K; e += 256) {
            const int r = e / PD_K, k = e - r * PD_K;
            const int64_t kk = k0 + k;
            sx[k][r] = (i0 + r < m && kk < d) ? x[(int64_t)(i0 + r) * d + kk] : 0.f;
            sy[k][r] = (j0 + r < n && kk < d) ? y[(int64_t)(j0 + r) * d + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < PD_K; ++k) {
            float xv[4], yv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { xv[a] = sx[k][ty + 16 * a]; yv[a] = sy[k][tx + 16 * a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float df = xv[a] - yv[b];
                    acc[a][b] += df * df;
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * b;
            if (i < m && j < n) out[(int64_t)i * n + j] = take_sqrt ? sqrtf(acc[a][b]) : acc[a][b];
        }
}

extern "C" int sda_pairwise_dist(const float* x, int m, const float* y, int n, int64_t d, int take_sqrt, float* out,
                                 void* stream) {
    if (!x || !y || !out || m <= 0 || n <= 0 || d <= 0) return SDA_E_BADARG;
    dim3 grid((n + PD_TILE - 1) / PD_TILE, (m + PD_TILE - 1) / PD_TILE);
    if (grid.y > 65535) return SDA_E_UNSUPPORTED;
    hipLaunchKernelGGL(pairwise_dist_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, m, y, n, d, take_sqrt, out);
    return sda_launch_status();
}

__global__ __launch_bounds__(256) void mmd_kernel_sums_kernel(const float* __restrict__ d2, int64_t count,
                                                              double* __restrict__ partial) {
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float e = d2[i];
        float k = 0.f;
        // bandwidths of utils.py:252, smallest first (the reference adds them in this order)
        k += expf(-e / 1e-3f);
        k += expf(-e / 1e-2f);
        k += expf(-e / 1e-1f);
        k += expf(-e / 1e-0f);
        k += expf(-e / 1e1f);
        k += expf(-e / 1e2f);
        k += expf(-e / 1e3f);
        s += (double)k;
    }
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

extern "C" int sda_mmd_kernel_sums(const float* d2, int64_t count, double* partial, int nblocks, void* stream) {
    if (!d2 || !partial || count <= 0 || nblocks <= 0 || nblocks > 65535) return SDA_E_BADARG;
    hipLaunchKernelGGL(mmd_kernel_sums_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, d2, count, partial);
    return sda_launch_status();
}

// min over permutations p of sum_i cost[i][p(i)]  (cost: n x n row-major, HOST memory).  col_of_row may be null.
extern "C" int sda_assignment_cost(const float* cost, int n, double* total, int* col_of_row) {
    if (!cost || !total || n <= 0) return SDA_E_BADARG;
    const double INF = std::numeric_limits<double>::infinity();
    // potentials u (rows), v (columns); match[j] = row assigned to column j; 1-based with a virtual column 0
    std::vector<double> u(n + 1, 0.0), v(n + 1, 0.0), minv(n + 1);
    std::vector<int> match(n + 1, 0), way(n + 1, 0);
    std::vector<char> used(n + 1);
    for (int i = 1; i <= n; ++i) {
        match[0] = i;
        int j0 = 0;
        std::fill(minv.begin(), minv.end(), INF);
        std::fill(used.begin(), used.end(), 0);
        do {                                          // grow the alternating tree until a free column is reached
            used[j0] = 1;
            const int i0 = match[j0];
            const float* row = cost + (int64_t)(i0 - 1) * n;
            double delta = INF;
            int j1 = 0;
            for (int j = 1; j <= n; ++j) {
                if (used[j]) continue;
                const double cur = (double)row[j - 1] - u[i0] - v[j];
                if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                if (minv[j] < delta) { delta = minv[j]; j1 = j; }
            }
            if (j1 == 0) return SDA_E_BADARG;         // NaN / inf costs
            for (int j = 0; j <= n; ++j) {
                if (used[j]) { u[match[j]] += delta; v[j] -= delta; }
                else minv[j] -= delta;
            }
            j0 = j1;
        } while (match[j0] != 0);
        do {                                          // flip the augmenting path
            const int j1 = way[j0];
            match[j0] = match[j1];
            j0 = j1;
        } while (j0);
    }
    double sum = 0.0;
    for (int j = 1; j <= n; ++j) {
        sum += (double)cost[(int64_t)(match[j] - 1) * n + (j - 1)];
        if (col_of_row) col_of_row[match[j] - 1] = j - 1;
    }
    *total = sum;
    return SDA_OK;
}

// min over couplings P >= 0 with row sums 1/m and column sums 1/n of <P, cost>  (cost: m x n row-major, HOST memory): the
// optimal-transport LP behind emd() for UNEQUAL sample counts (sda/utils.py:203-219 passes empty weight vectors to POT =
// uniform marginals).  Scaled by m n the marginals are integers (every row ships n units, every column takes m), so this is
// an integral min-cost flow on the complete bipartite graph: successive shortest paths with node potentials (dense Dijkstra
// from all rows with supply left to the nearest column with demand left, early exit), augmenting by the bottleneck of
// supply, demand and the flows on the path's backward arcs.  Double precision; O((m + n)^2) per augmentation.
extern "C" int sda_transport_cost(const float* cost, int m, int n, double* total) {
    if (!cost || !total || m <= 0 || n <= 0) return SDA_E_BADARG;
    const double INF = std::numeric_limits<double>::infinity();
    const int V = m + n;                                   // nodes: rows 0 .. m-1, columns m .. m+n-1
    std::vector<long long> flow((size_t)m * n, 0), supply(m, n), demand(n, m);
    std::vector<double> pi(V, 0.0), dist(V);
    std::vector<int> prev(V);
    std::vector<char> done(V);
    long long left = (long long)m * n;
    for (int64_t e = 0; e < (int64_t)m * n; ++e)
        if (!(cost[e] >= 0.f) || !(cost[e] < std::numeric_limits<float>::infinity())) return SDA_E_BADARG;   // NaN / inf / negative
    while (left > 0) {
        for (int v = 0; v < V; ++v) { dist[v] = INF; done[v] = 0; prev[v] = -1; }
        for (int i = 0; i < m; ++i) if (supply[i] > 0) dist[i] = 0.0;
        int target = -1;
        for (;;) {
            int v = -1;
            double best = INF;
            for (int w = 0; w < V; ++w) if (!done[w] && dist[w] < best) { best = dist[w]; v = w; }
            if (v < 0) return SDA_E_BADARG;                // (cannot happen: the graph is complete)
            done[v] = 1;
            if (v >= m) {
                const int j = v - m;
                if (demand[j] > 0) { target = v; break; }
                for (int i = 0; i < m; ++i) {              // backward arcs j -> i where flow is routed
                    if (done[i] || flow[(size_t)i * n + j] == 0) continue;
                    double rc = -(double)cost[(size_t)i * n + j] + pi[v] - pi[i];
                    if (rc < 0) rc = 0;                    // (round-off: reduced costs of flow arcs are zero)
                    if (best + rc < dist[i]) { dist[i] = best + rc; prev[i] = v; }
                }
            } else {
                const float* row = cost + (size_t)v * n;
                for (int j = 0; j < n; ++j) {
                    if (done[m + j]) continue;
                    double rc = (double)row[j] + pi[v] - pi[m + j];
                    if (rc < 0) rc = 0;
                    if (best + rc < dist[m + j]) { dist[m + j] = best + rc; prev[m + j] = v; }
                }
            }
        }
        const double dt = dist[target];
        for (int v = 0; v < V; ++v) pi[v] += (done[v] && dist[v] < dt) ? dist[v] : dt;
        // bottleneck along the path
        long long amount = demand[target - m];
        int v = target;
        while (prev[v] >= 0) {
            const int p = prev[v];
            if (p >= m) { const long long f = flow[(size_t)v * n + (p - m)]; if (f < amount) amount = f; }   // backward arc p -> v
            v = p;
        }
        if (supply[v] < amount) amount = supply[v];
        supply[v] -= amount;
        demand[target - m] -= amount;
        left -= amount;
        v = target;
        while (prev[v] >= 0) {
            const int p = prev[v];
            if (p < m) flow[(size_t)p * n + (v - m)] += amount;
            else flow[(size_t)v * n + (p - m)] -= amount;
            v = p;
        }
    }
    double sum = 0.0;
    for (int64_t e = 0; e < (int64_t)m * n; ++e) sum += (double)flow[e] * (double)cost[e];
    *total = sum / ((double)m * (double)n);
    return SDA_OK;
}
