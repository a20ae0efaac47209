// One SGD step of SDAV layer `layer` (src/sdav/network/SDAV.py:126-159 forward, :171-186 loss,
// :223-226 plain gradient descent; `optimizer.minimize(loss_l)` reaches every variable the loss
// depends on, so the encoders of layers 0..layer-1 move too).  fp64 like the reference.
// All matrix products go through the fp64 MFMA GEMM (gemm_dense.hip); the loss gradients are
// row / elementwise HIP kernels.  Restated (and pinned by finite differences) in
// oracle/sdav_train.py.
#include "dlc_internal.h"

namespace dlc_gemm {
int gemm_bias_act(dlc_ctx* ctx, int dtype, int blayout, int act, int64_t M, int64_t N, int64_t K, const void* A,
                  int64_t lda, const void* B, int64_t ldb, const void* bias, void* C, int64_t ldc, hipStream_t st);
int gemm_axpy_dma_f64(dlc_ctx* ctx, int blayout, double alpha, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                      const double* B, int64_t ldb, double* C, int64_t ldc, hipStream_t st);      // gemm_dma_f64.hip
// A zero-padded to lda = Kpad columns, B [K, N] (gemm_dense.hip): the LDS-DMA forms (split-K on 64-row tiles when the context
// has scratch and the tiles are few) where they apply, else the register-staged kernel on the first K columns
int gemm_bias_act_padded_f64(dlc_ctx* ctx, int act, int64_t M, int64_t N, int64_t K, int64_t Kpad, const double* A,
                             const double* B, int64_t ldb, const double* bias, double* C, int64_t ldc, hipStream_t st);
}

namespace {

__device__ __forceinline__ double block_sum(double v, double* red) {   // 256 threads
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ double block_max(double v, double* red) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// x_tilde[r, c] = x[r, c] * mask[r % P, c]     (TensorWrapper.corrupt, TensorflowWrapper.py:34-38)
// out has pitch ldo >= cols; its columns cols .. ldo - 1 are written as zeros (an odd width -- the 1681 pixels of a patch -- is
// padded to an even pitch so that the LDS-DMA GEMM's 16-byte pieces start on 16-byte boundaries: even_pitch below)
__global__ __launch_bounds__(256) void mask_rows_kernel(const double* __restrict__ x, const double* __restrict__ mask,
                                                        long long rows, int P, long long cols, double* __restrict__ out,
                                                        long long ldo) {
    const long long total = rows * ldo;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / ldo, c = e - r * ldo;
        out[e] = c < cols ? x[r * cols + c] * mask[(r % P) * cols + c] : 0.0;
    }
}

// random_mask (TensorflowWrapper.py:148-156): EXACTLY n_zeros zeros among n ones, every subset equally likely (the
// reference concatenates zeros and ones and tf.random.shuffle's them).  Element i draws the 32-bit key
// hash(seed, counter, i); the n_zeros smallest (key, index) pairs become the zeros -- found by one workgroup: a select
// over the keys in a window around the expected threshold (the kernel's short way), or a three-pass radix select over all
// keys (11 + 11 + 10 bits, histograms in LDS), and a last pass that writes the mask; keys are recomputed, never stored;
// equal keys at the threshold go by index (a prefix sum over the threads orders them).  25 us for the 50 430 elements of
// the reference's first layer (rocprofv3), where a device randperm (a sort) took 83 us of a 470 us training step.
__device__ __forceinline__ unsigned mask_key(unsigned s0, unsigned s1, unsigned i) {
    // two rounds of murmur3's 32-bit finaliser over (index, seed, counter): 32-bit multiplies only (the 64-bit mix this
    // replaced cost ~250 cycles a key, and the kernel is one workgroup)
    unsigned h = i * 0x9e3779b1u + s0;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    h += s1;
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
    return h;
}

constexpr int RM_THREADS = 1024;
#define IDX(j) ((long long)tid + (long long)(j) * RM_THREADS)
// (Forms that kept a thread's keys in registers between three histogram passes over ALL keys -- 80, then 52 keys per thread
// -- were 38 and 32 us for the reference's 50 430 elements; this one is 25: what is left is the keys themselves, five 32-bit
// multiplies each at a quarter of the vector rate, generated twice.)
__global__ __launch_bounds__(RM_THREADS) void random_mask_kernel(double* __restrict__ mask, long long n, long long n_zeros,
                                                                 unsigned long long seed, unsigned long long counter) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned sel_bin, sel_below;
    __shared__ unsigned eq_cnt[RM_THREADS];
    const int tid = threadIdx.x;
    const unsigned s0 = (unsigned)seed ^ (unsigned)(counter >> 32) * 0x27d4eb2fu;
    const unsigned s1 = (unsigned)(seed >> 32) + (unsigned)counter * 0x165667b1u;
    // thread t owns the indices t, t + 1024, ...: the mask is written in coalesced rows (a contiguous range per thread wrote
    // 50 000 scattered doubles from one CU: 30 us); a key is recomputed wherever it is wanted
    auto key_at = [&](int j) { return mask_key(s0, s1, (unsigned)IDX(j)); };
    // The short way: the keys are uniform, so the n_zeros-th smallest lies within a few
    // standard deviations of n_zeros / n * 2^32.  One pass over the keys counts those below a window of +- 8 sigma around
    // that value and lists the ~16 sigma keys inside it (1 700 of the reference's 50 430); the radix select then runs over
    // the LIST -- two keys per thread -- and a second pass over the keys writes the mask.  Two key generations and a list
    // instead of one generation and three histogram passes of 50 LDS atomics per thread.  A window that does not hold the
    // threshold (1e-15), a list that overflows, or equal keys AT the threshold (1e-5: which of them are zeros is then a
    // matter of order) leave the decision to the general passes below; without ties both ways give the same mask.
    {
        constexpr int RM_CAP = 4096;
        __shared__ unsigned l_len, l_below, l_eq;
        __shared__ unsigned lkey[RM_CAP];
        const double pz = (double)n_zeros / (double)n, sig = sqrt((double)n * pz * (1.0 - pz));
        const double mid = pz * 4294967296.0, wid = (8.0 * sig + 4.0) * 4294967296.0 / (double)n;
        if (n_zeros > 0 && n_zeros < n && mid - wid > 0.0 && mid + wid < 4294967295.0) {
            const unsigned lo = (unsigned)(mid - wid), span = (unsigned)(mid + wid) - lo;
            if (tid == 0) { l_len = 0; l_below = 0; l_eq = 0; }
            __syncthreads();
            unsigned below = 0;
            for (int j = 0; IDX(j) < n; ++j) {
                const unsigned k = key_at(j);
                below += k < lo ? 1u : 0u;
                if (k - lo <= span) {
                    const unsigned slot = atomicAdd(&l_len, 1u);
                    if (slot < (unsigned)RM_CAP) lkey[slot] = k - lo;
                }
            }
            for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
            if ((tid & 63) == 0) atomicAdd(&l_below, below);
            __syncthreads();
            const unsigned len = l_len, nbelow = l_below;
            if (len <= (unsigned)RM_CAP && (long long)nbelow < n_zeros && n_zeros <= (long long)nbelow + (long long)len) {
                unsigned pre = 0, pm = 0;
                long long wnt = n_zeros - (long long)nbelow;          // the wnt-th smallest offset of the list
                for (int sh = 24; sh >= 0; sh -= 8) {                // four passes of eight bits, 256 bins: four a lane
                    if (tid < 256) hist[tid] = 0;
                    __syncthreads();
                    for (unsigned e = tid; e < len; e += RM_THREADS)
                        if ((lkey[e] & pm) == pre) atomicAdd(&hist[(lkey[e] >> sh) & 255u], 1u);
                    __syncthreads();
                    if (tid < 64) {
                        const unsigned h0 = hist[tid * 4], h1 = hist[tid * 4 + 1], h2 = hist[tid * 4 + 2], h3 = hist[tid * 4 + 3];
                        const unsigned sum = h0 + h1 + h2 + h3;
                        unsigned inc = sum;
                        for (int o = 1; o < 64; o <<= 1) {
                            const unsigned v = __shfl_up(inc, o);
                            if (tid >= o) inc += v;
                        }
                        const unsigned exc = inc - sum;
                        if ((long long)exc < wnt && wnt <= (long long)inc) {
                            unsigned bl = exc, b = 0;
                            if ((long long)(bl + h0) < wnt) { bl += h0; b = 1;
                                if ((long long)(bl + h1) < wnt) { bl += h1; b = 2;
                                    if ((long long)(bl + h2) < wnt) { bl += h2; b = 3; } } }
                            sel_bin = (unsigned)tid * 4 + b; sel_below = bl;
                        }
                    }
                    __syncthreads();
                    pre |= sel_bin << sh;
                    pm |= 255u << sh;
                    wnt -= sel_below;
                    __syncthreads();
                }
                // pre = the threshold's offset; wnt of the keys equal to it are zeros: all of them, unless keys collide there
                for (unsigned e = tid; e < len; e += RM_THREADS)
                    if (lkey[e] == pre) atomicAdd(&l_eq, 1u);
                __syncthreads();
                if ((long long)l_eq == wnt) {
                    const unsigned thr = lo + pre;
                    for (int j = 0; IDX(j) < n; ++j) mask[IDX(j)] = key_at(j) <= thr ? 0.0 : 1.0;
                    return;
                }
            }
            __syncthreads();
        }
    }
    unsigned prefix = 0, pmask = 0;          // the bits of the threshold key found so far, and which bits they are
    long long want = n_zeros;                // how many of the keys matching the prefix are still to be taken
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3 && want > 0 && want < n; ++pass) {
        const int nb = 1 << widths[pass];
        for (int b = tid; b < nb; b += RM_THREADS) hist[b] = 0;
        __syncthreads();
        for (int j = 0; IDX(j) < n; ++j) {
            const unsigned k = key_at(j);
            if ((k & pmask) == prefix) atomicAdd(&hist[(k >> shifts[pass]) & (nb - 1)], 1u);
        }
        __syncthreads();
        if (tid < 64) {                      // the bin that holds the want-th smallest key: lane sums of nb / 64 bins, a
            const int per = nb / 64;         // wave scan over them, then the one lane whose range holds it walks its bins
            unsigned sum = 0;                // (one thread walking 2048 bins paid an LDS round trip per bin: 0.2 ms)
            for (int j = 0; j < per; ++j) sum += hist[tid * per + j];
            unsigned inc = sum;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = __shfl_up(inc, o);
                if (tid >= o) inc += v;
            }
            const unsigned exc = inc - sum;
            if ((long long)exc < want && want <= (long long)inc) {
                unsigned below = exc;
                int b = tid * per;
                for (; b < tid * per + per - 1 && (long long)(below + hist[b]) < want; ++b) below += hist[b];
                sel_bin = (unsigned)b; sel_below = below;
            }
        }
        __syncthreads();
        prefix |= sel_bin << shifts[pass];
        pmask |= (unsigned)(nb - 1) << shifts[pass];
        want -= sel_below;
        __syncthreads();
    }
    // keys < prefix: zeros; keys == prefix: the first `want` of them in (thread, j) order -- any fixed rule does: the keys are random
    const bool all = n_zeros >= n, none = n_zeros <= 0;
    unsigned mine = 0;
    for (int j = 0; IDX(j) < n; ++j) mine += key_at(j) == prefix ? 1u : 0u;
    eq_cnt[tid] = mine;
    __syncthreads();
    // exclusive prefix of the threads' counts (keys equal to the threshold are a handful: 32-bit keys rarely collide)
    if (tid < 64) {
        unsigned sum = 0;
        for (int j = 0; j < RM_THREADS / 64; ++j) sum += eq_cnt[tid * (RM_THREADS / 64) + j];
        unsigned inc = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(inc, o);
            if (tid >= o) inc += v;
        }
        hist[tid] = inc - sum;               // what lies in front of this lane's 16 threads
    }
    __syncthreads();
    long long before = hist[tid / (RM_THREADS / 64)];
    for (int t = tid / (RM_THREADS / 64) * (RM_THREADS / 64); t < tid; ++t) before += eq_cnt[t];
    auto put = [&](int j, unsigned k) {
        bool zero = k < prefix;
        if (k == prefix) { zero = before < want; ++before; }
        mask[IDX(j)] = (all || (zero && !none)) ? 0.0 : 1.0;
    };
    for (int j = 0; IDX(j) < n; ++j) put(j, key_at(j));
}
#undef IDX

// softmax_cross_entropy_with_logits_v2(labels, logits = y) per row (SDAV.py:172), mean over rows.
// dz2 = d(cd)/d(y) * y(1-y);  dlab (optional) = d(cd)/d(labels) = -log_softmax(y)/rows.
// lab has pitch ldl, dz2 pitch ldz >= cols (its columns past cols are written as zeros: the A operand of dh = dz2 W).
__global__ __launch_bounds__(256) void xent_grad_kernel(const double* __restrict__ y, const double* __restrict__ lab, long long ldl,
                                                        long long rows, int cols, double* __restrict__ dz2, long long ldz,
                                                        double* __restrict__ dlab, double* __restrict__ acc) {
    __shared__ double red[4];
    const long long r = blockIdx.x;
    const double* yr = y + r * cols;
    const double* lr = lab + r * ldl;
    double mx = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmax(mx, yr[c]);
    mx = block_max(mx, red);
    double se = 0.0, sl = 0.0;
    for (int c = threadIdx.x; c < cols; c += 256) { se += exp(yr[c] - mx); sl += lr[c]; }
    se = block_sum(se, red);
    sl = block_sum(sl, red);
    const double lse = log(se), inv_rows = 1.0 / (double)rows;
    double cd = 0.0;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const double yv = yr[c], lv = lr[c];
        const double logsm = yv - mx - lse;
        cd -= lv * logsm;
        const double dy = (exp(logsm) * sl - lv) * inv_rows;
        dz2[r * ldz + c] = dy * yv * (1.0 - yv);
        if (dlab) dlab[r * cols + c] = -logsm * inv_rows;
    }
    for (long long c = cols + threadIdx.x; c < ldz; c += 256) dz2[r * ldz + c] = 0.0;
    cd = block_sum(cd, red);
    if (threadIdx.x == 0) acc[r] = cd * inv_rows;             // cd_part[r]: summed in order by update_kernel's last workgroup
}

constexpr int FN_MAX_SLICES = 64;
// n_t = || H_t - H_{t+1} ||_F over a frame's P x N block (SDAV.py:176-183); also cs (:174):
// sum |h - s| / cs_den, where cs_den is the number of entries reduce_mean sees after the axis-1
// norm -- batch*N at layer 0 (h is 3-D [B,P,N] there: axis 1 = patches), batch*P afterwards.
__global__ __launch_bounds__(256) void frame_norm_kernel(const double* __restrict__ h, int batch, long long frame_elems,
                                                         double sparse_level, double cs_den, double* __restrict__ nrm_part,
                                                         double* __restrict__ acc) {
    // Per frame t: its share of the sparsity term and the squared distance to frame t + 1 (the consecutive-frame term).
    // A frame is cut into gridDim.y slices (one workgroup per frame walked 75 000 elements alone: 97 us of a 790 us
    // step); slice y leaves its part in nrm_part[t * 64 + y]; hidden_grad_kernel and update_kernel add the parts IN ORDER and take
    // the roots: nrm feeds the consecutive-frame gradient, so the SGD trajectory is the same bits run after run (an
    // atomicAdd here made it depend on the order the workgroups finished).
    __shared__ double red[4];
    const int t = blockIdx.x;                                   // 0 .. batch-1
    const double* a = h + (long long)t * frame_elems;
    double s2 = 0.0, l1 = 0.0;
    for (long long e = (long long)blockIdx.y * 256 + threadIdx.x; e < frame_elems; e += (long long)gridDim.y * 256) {
        const double v = a[e];
        l1 += fabs(v - sparse_level);
        if (t + 1 < batch) { const double dlt = v - a[frame_elems + e]; s2 += dlt * dlt; }
    }
    s2 = block_sum(s2, red);
    l1 = block_sum(l1, red);
    if (threadIdx.x == 0) {
        acc[(long long)t * FN_MAX_SLICES + blockIdx.y] = l1 / cs_den;        // l1_part, as nrm_part
        nrm_part[(long long)t * FN_MAX_SLICES + blockIdx.y] = s2;
    }
}
// dh += sparse_penalty*sign(h - s)/cs_den + consecutive term;  dz1 = dh * h(1-h)
// nrm[f] = sqrt of frame f's squared distance to frame f + 1: the slices' parts added in slice order (every workgroup adds
// the same parts in the same order: the same bits, and no launch of its own for a few hundred additions)
__global__ __launch_bounds__(256) void hidden_grad_kernel(const double* __restrict__ h, const double* __restrict__ dh_in,
                                                          const double* __restrict__ nrm_part, int slices, int batch,
                                                          long long frame_elems, double cs_den, double sparse_level,
                                                          double sparse_penalty, double consecutive_penalty,
                                                          double* __restrict__ dz1) {
    extern __shared__ double nrm[];                            // [batch - 1]
    for (int t = threadIdx.x; t + 1 < batch; t += 256) {
        double n2 = 0.0;
        for (int y = 0; y < slices; ++y) n2 += nrm_part[(long long)t * FN_MAX_SLICES + y];
        nrm[t] = sqrt(n2);
    }
    __syncthreads();
    const long long total = (long long)batch * frame_elems;
    const double ccs = consecutive_penalty / (double)(batch - 1), sps = sparse_penalty / cs_den;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long f = e / frame_elems;
        const double hv = h[e];
        double g = dh_in[e];
        const double d = hv - sparse_level;
        g += sps * (d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0));
        if (f + 1 < batch) g += ccs * (hv - h[e + frame_elems]) / nrm[f];
        if (f > 0) g -= ccs * (h[e - frame_elems] - hv) / nrm[f - 1];
        dz1[e] = g * hv * (1.0 - hv);
    }
}

// dz1_prev = (dxt [+ dlab]) * mask[r % P] * h_prev(1-h_prev)
__global__ __launch_bounds__(256) void backprop_input_kernel(const double* __restrict__ dxt, const double* __restrict__ dlab,
                                                             const double* __restrict__ mask,
                                                             const double* __restrict__ h_prev, long long rows, int P,
                                                             long long cols, double* __restrict__ dz1_prev) {
    const long long total = rows * cols;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / cols, c = e - r * cols;
        double g = dxt[e];
        if (dlab) g += dlab[e];
        const double hv = h_prev[e];
        dz1_prev[e] = g * mask[(r % P) * cols + c] * hv * (1.0 - hv);
    }
}

// out[c * ldo + r] = in[r, c]  (ldo >= rows: two transposes side by side make the K-stacked operand of the fused
// weight-gradient product)
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ in, long long ldi, long long rows, long long cols,
                                                        double* __restrict__ out, long long ldo,
                                                        const double* __restrict__ in2 = nullptr, long long ldi2 = 0,
                                                        double* __restrict__ out2 = nullptr) {
    __shared__ double tile[32][33];
    if (blockIdx.z == 1) { in = in2; ldi = ldi2; out = out2; } // (a second matrix of the same shape in the same launch)
    const long long r0 = (long long)blockIdx.y * 32, c0 = (long long)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(r0 + i) * ldi + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) out[(c0 + i) * ldo + r0 + tx] = tile[tx][i];
}

// out[c] = sum_r in[r, c]: 32 columns x 8 row groups per workgroup, the groups' partial sums combined in group order (kept
// for the encoder biases of the layers between 1 and the trained one, whose dz1 buffer is reused before the step's end)
__global__ __launch_bounds__(256) void colsum_kernel(const double* __restrict__ in, long long rows, long long cols,
                                                     double* __restrict__ out) {
    __shared__ double part[8][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const long long c = (long long)blockIdx.x * 32 + cl;
    double s = 0.0;
    if (c < cols) {
#pragma unroll 4
        for (long long r = rg; r < rows; r += 8) s += in[r * cols + c];
    }
    part[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < cols) {
        double t = part[0][cl];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += part[g][cl];
        out[c] = t;
    }
}

// Everything behind the last product of a step in ONE launch (r04: two column-sum kernels, five SGD kernels and the loss
// kernel -- eight launches of microseconds each): workgroups [0, w_blocks) walk the weight matrices (p -= lr g), the next
// ones take 32 bias columns each (the column sum of their gradient rows, 8 row groups combined in group order, then the
// update), the last one adds the loss's parts in their fixed order.  Same arithmetic, same order, same bits as the
// separate kernels.
constexpr int UP_MAX = 8;
struct UpdateArgs {
    int n_w;                                    // weight matrices
    double* w[UP_MAX]; const double* gw[UP_MAX]; long long w_n[UP_MAX]; long long w_blk0[UP_MAX + 1];   // block ranges
    int n_b;                                    // bias vectors
    double* b[UP_MAX + 1]; const double* gb_rows[UP_MAX + 1]; long long b_cols[UP_MAX + 1]; long long b_blk0[UP_MAX + 2];
    long long b_rows[UP_MAX + 1];               // gradient rows to add up (1: gb_rows is the gradient itself)
    long long b_ld[UP_MAX + 1];                 // ... and their pitch
    long long rows;
    double lr;
    // loss
    const double* cd_part; const double* nrm_part; const double* l1_part; int batch, slices;
    double sparse_penalty, consecutive_penalty; double* loss_out;
};
__global__ __launch_bounds__(256) void update_kernel(UpdateArgs a) {
    __shared__ double part[8][33];
    __shared__ double red[4];
    const long long blk = blockIdx.x;
    if (blk < a.w_blk0[a.n_w]) {
        int m = 0;
        while (blk >= a.w_blk0[m + 1]) ++m;
        const long long nb = a.w_blk0[m + 1] - a.w_blk0[m];
        double* __restrict__ p = a.w[m];
        const double* __restrict__ g = a.gw[m];
        for (long long e = (blk - a.w_blk0[m]) * 256 + threadIdx.x; e < a.w_n[m]; e += nb * 256) p[e] -= a.lr * g[e];
        return;
    }
    if (blk < a.b_blk0[a.n_b]) {
        int m = 0;
        while (blk >= a.b_blk0[m + 1]) ++m;
        const long long cols = a.b_cols[m];
        const double* __restrict__ in = a.gb_rows[m];
        const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
        const long long c = (blk - a.b_blk0[m]) * 32 + cl;
        double s = 0.0;
        if (c < cols) {
#pragma unroll 4
            for (long long r = rg; r < a.b_rows[m]; r += 8) s += in[r * a.b_ld[m] + c];
        }
        part[rg][cl] = s;
        __syncthreads();
        if (rg == 0 && c < cols) {
            double t = part[0][cl];
#pragma unroll
            for (int g = 1; g < 8; ++g) t += part[g][cl];
            a.b[m][c] -= a.lr * t;
        }
        return;
    }
    // {loss, cd, cs, cc}: the rows' cross-entropies added in row order, the frames' sparsity / consecutive-frame parts in
    // slice then frame order (a fixed tree: the reported loss is the same bits run after run, like the parameters)
    if (!a.loss_out) return;
    double cd = 0.0;
    for (long long r = threadIdx.x; r < a.rows; r += 256) cd += a.cd_part[r];
    cd = block_sum(cd, red);
    // the frames' parts: thread t adds frame t's slices in slice order (every frame at once -- one thread walking all of them
    // was batch x slices dependent round trips, most of this launch's 17 us), thread 0 then adds the frames in frame order:
    // the same additions in the same order as one thread would make
    __shared__ double f_l1[256], f_n2[256];
    double cs = 0.0, cc = 0.0;
    for (int t0 = 0; t0 < a.batch; t0 += 256) {
        const int t = t0 + (int)threadIdx.x;
        if (t < a.batch) {
            double n2 = 0.0, l1 = 0.0;
            for (int y = 0; y < a.slices; ++y) {
                n2 += a.nrm_part[(long long)t * FN_MAX_SLICES + y];
                l1 += a.l1_part[(long long)t * FN_MAX_SLICES + y];
            }
            f_l1[threadIdx.x] = l1; f_n2[threadIdx.x] = n2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int u = 0; u < 256 && t0 + u < a.batch; ++u) {
                cs += f_l1[u];
                if (t0 + u + 1 < a.batch) cc += sqrt(f_n2[u]) / (double)(a.batch - 1);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.loss_out[0] = cd + a.sparse_penalty * cs + a.consecutive_penalty * cc;
        a.loss_out[1] = cd;
        a.loss_out[2] = cs;
        a.loss_out[3] = cc;
    }
}

inline unsigned grid_for(long long n) {
    long long b = dlc::cdiv(n, 256);
    return (unsigned)(b > 256 * 32 ? 256 * 32 : (b < 1 ? 1 : b));
}

struct TrainWs {
    size_t xt[8], h[8], gw[8], gbe[8];     // per layer 0..layer
    size_t y, dz2, dlab, dh, dz1a, dz1b, dxt, tr, gbd, nrm, nrm_part, l1_part, cd_part, acc, total;   // (gw2 is gone: one product makes both uses of the tied weight)
};

inline int64_t even_pitch(int64_t cols) { return cols + (cols & 1); }

TrainWs train_ws(int64_t rows, int batch, const int64_t* dims, int layer) {
    TrainWs w;
    size_t o = 0;
    auto take = [&](size_t elems) { size_t at = o; o += dlc::align_up(elems * 8, 256); return at; };
    int64_t wmax = 0;
    for (int l = 0; l <= layer + 1; ++l) wmax = dims[l] > wmax ? dims[l] : wmax;
    for (int l = 0; l <= layer; ++l) {
        w.xt[l] = take((size_t)rows * even_pitch(dims[l]));
        // behind the trained layer's h: the first dz1 (hidden_grad_kernel's output) -- [h ; dz1] is the K-stacked B operand
        // of the fused weight-gradient product
        w.h[l] = take((size_t)rows * dims[l + 1] * (l == layer ? 2 : 1));
        w.gw[l] = take((size_t)dims[l] * dims[l + 1]);
        w.gbe[l] = take((size_t)dims[l + 1]);
    }
    w.y = take((size_t)rows * dims[layer]);
    w.dz2 = take((size_t)rows * even_pitch(dims[layer]));
    w.dlab = take((size_t)rows * dims[layer]);
    w.dh = take((size_t)rows * wmax);
    w.dz1a = take((size_t)rows * wmax);
    w.dz1b = take((size_t)rows * wmax);
    w.dxt = take((size_t)rows * wmax);
    w.tr = take((size_t)rows * wmax * 2);
    w.gbd = take((size_t)dims[layer]);
    w.nrm = take((size_t)batch);
    w.nrm_part = take((size_t)batch * FN_MAX_SLICES);
    w.l1_part = take((size_t)batch * FN_MAX_SLICES);
    w.cd_part = take((size_t)rows);
    w.acc = take(4);
    w.total = o;
    return w;
}

}  // namespace

extern "C" size_t dlc_sdav_train_workspace_bytes(int64_t batch, int64_t patches, const int64_t* dims, int n_layers,
                                                 int layer) {
    if (batch < 2 || patches < 1 || !dims || n_layers < 1 || n_layers > 8 || layer < 0 || layer >= n_layers) return 0;
    return train_ws(batch * patches, (int)batch, dims, layer).total;
}

extern "C" int dlc_sdav_train_step(dlc_ctx* ctx, int layer, int64_t batch, int64_t patches, int n_layers,
                                   const int64_t* dims, const double* x, const double* const* masks, double* const* W,
                                   double* const* b_enc, double* b_dec, double sparse_level, double sparse_penalty,
                                   double consecutive_penalty, double learning_rate, double* loss_out, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!dims || !x || !masks || !W || !b_enc || !b_dec || n_layers < 1 || n_layers > 8 || layer < 0 || layer >= n_layers ||
        patches < 1)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_train_step: bad argument");
    if (batch < 2)
        return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "sdav_train_step: a batch needs >= 2 frames (consecutive-frame term, SDAV.py:176-183)");
    if (batch * patches > 0x7fffffffll) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "sdav_train_step: batch too large");
    if (dims[1] != dims[layer + 1])
        return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "sdav_train_step: hidden_units[0] != hidden_units[layer] (the slice of SDAV.py:178-181 needs equal widths)");
    for (int l = 0; l <= layer; ++l)
        if (!masks[l] || !W[l] || !b_enc[l]) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "sdav_train_step: null parameter of layer %d", l);
    const long long rows = batch * patches;
    const TrainWs w = train_ws(rows, (int)batch, dims, layer);
    if (!workspace || workspace_bytes < w.total)
        return dlc::fail(ctx, DLC_ERR_WORKSPACE, "sdav_train_step: workspace %zu < %zu bytes", workspace_bytes, w.total);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    auto P = [&](size_t off) { return (double*)(ws + off); };
    const int Pn = (int)patches;
#define GEMM(bl, act, M, N, K, A, lda, B, ldb, bias, C, ldc)                                                     \
    do {                                                                                                        \
        int rc_ = dlc_gemm::gemm_bias_act(ctx, DLC_F64, bl, act, M, N, K, A, lda, B, ldb, bias, C, ldc, st);     \
        if (rc_ != DLC_OK) return rc_;                                                                          \
    } while (0)

    // ---- forward through layers 0..layer (old parameters everywhere)
    const double* cur = x;
    for (int l = 0; l <= layer; ++l) {
        const long long kp = even_pitch(dims[l]);                  // (1681 -> 1682: a zero column the weights have no row for)
        hipLaunchKernelGGL(mask_rows_kernel, dim3(grid_for(rows * kp)), dim3(256), 0, st, cur, masks[l], rows, Pn,
                           (long long)dims[l], P(w.xt[l]), kp);
        {
            const int rc_ = dlc_gemm::gemm_bias_act_padded_f64(ctx, DLC_ACT_SIGMOID, rows, dims[l + 1], dims[l], kp, P(w.xt[l]), W[l],
                                                              dims[l + 1], b_enc[l], P(w.h[l]), dims[l + 1], st);
            if (rc_ != DLC_OK) return rc_;
        }
        cur = P(w.h[l]);
    }
    const long long K = dims[layer], N = dims[layer + 1];
    const double* h = P(w.h[layer]);
    GEMM(DLC_B_NK, DLC_ACT_SIGMOID, rows, K, N, h, N, W[layer], N, b_dec, P(w.y), K);            // y = sigmoid(h W^T + b_d)
    const long long Kp = even_pitch(K);                            // pitch of x~ and dz2 at the trained layer
    const double* labels = layer == 0 ? x : P(w.xt[layer]);       // (x: the caller's, pitch K; x~ of a deeper layer: K is even there
    const long long ld_labels = layer == 0 ? K : Kp;              //  or the pitch is Kp)

    // ---- loss pieces and the gradient at the trained layer
    hipLaunchKernelGGL(xent_grad_kernel, dim3((unsigned)rows), dim3(256), 0, st, P(w.y), labels, ld_labels, rows, (int)K, P(w.dz2), Kp,
                       layer > 0 ? P(w.dlab) : (double*)nullptr, P(w.cd_part));
    // tf.norm(h - s, axis=1, ord=1) + reduce_mean (SDAV.py:174): h is [B,P,N] at layer 0, [B*P,N] afterwards
    const double cs_den = layer == 0 ? (double)batch * (double)N : (double)rows;
    int fn_slices = 1;
    {
        long long slices = dlc::cdiv((long long)patches * N, (long long)256 * 16);      // >= 16 elements per thread
        if (slices > FN_MAX_SLICES) slices = FN_MAX_SLICES;
        hipLaunchKernelGGL(frame_norm_kernel, dim3((unsigned)batch, (unsigned)slices), dim3(256), 0, st, h, (int)batch,
                           (long long)patches * N, sparse_level, cs_den, P(w.nrm_part), P(w.l1_part));
        fn_slices = (int)slices;
    }
    {                                                                                             // dh = dz2 W
        const int rc_ = dlc_gemm::gemm_bias_act_padded_f64(ctx, DLC_ACT_NONE, rows, N, K, Kp, P(w.dz2), W[layer], N, nullptr, P(w.dh), N, st);
        if (rc_ != DLC_OK) return rc_;
    }
    double* dz1 = P(w.h[layer]) + rows * N;                   // right behind h
    if ((size_t)batch * 8 > 48 * 1024) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "sdav_train_step: batch too large");
    hipLaunchKernelGGL(hidden_grad_kernel, dim3(grid_for(rows * N)), dim3(256), (size_t)batch * 8, st, h, P(w.dh), P(w.nrm_part),
                       fn_slices, (int)batch, (long long)patches * N, cs_den, sparse_level, sparse_penalty, consecutive_penalty, dz1);
    // A layer's weight gradient and its SGD step in ONE launch where the LDS-DMA GEMM takes the shape: W += -lr * (A . B)
    // in the product's epilogue (gemm_axpy_dma_f64) -- dW never reaches memory, and the update kernel's pass over W and dW
    // (100 MB at layer 0: 25 us of a 0.5 ms step) is gone.  W[l] must be dead by then: every layer's product runs AFTER the
    // last use of its weights (dz1 W^T on the way down), the same products on the same operands as before, in another
    // order.  Where the kernel does not take the shape (odd widths) the gradient goes to w.gw[l] and update_kernel steps it.
    bool stepped[8] = {};
    auto weight_step = [&](int l, long long Kl, long long Nl, long long kk, const double* Bop) -> int {
        int rc_ = dlc_gemm::gemm_axpy_dma_f64(ctx, DLC_B_KN, -learning_rate, Kl, Nl, kk, P(w.tr), kk, Bop, Nl, W[l], Nl, st);
        if (rc_ == DLC_OK) { stepped[l] = true; return DLC_OK; }
        if (rc_ != 1) return rc_;
        return dlc_gemm::gemm_bias_act(ctx, DLC_F64, DLC_B_KN, DLC_ACT_NONE, Kl, Nl, kk, P(w.tr), kk, Bop, Nl, nullptr, P(w.gw[l]), Nl, st);
    };

    // ---- backward through the encoders layer .. 0
    double* spare[2] = {P(w.dz1a), P(w.dz1b)};
    const double* dz1_of[8] = {};                       // the rows whose column sums are layer l's encoder-bias gradient
    int which = 0;
    for (int l = layer; l >= 0; --l) {
        const long long Kl = dims[l], Nl = dims[l + 1];
        // dz1 of the trained layer lives behind h, the deeper ones alternate between two buffers: at the step's end only the
        // trained layer's and the last two written (layers 1 and 0) are still there; the others are summed now
        if (l == layer || l <= 1) dz1_of[l] = dz1;
        else hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)dlc::cdiv(Nl, 32)), dim3(256), 0, st, dz1, rows, Nl, P(w.gbe[l]));
        double* dz1_next = nullptr;
        if (l > 0) {
            GEMM(DLC_B_NK, DLC_ACT_NONE, rows, Kl, Nl, dz1, Nl, W[l], Nl, nullptr, P(w.dxt), Kl);          // dz1 W^T
            hipLaunchKernelGGL(backprop_input_kernel, dim3(grid_for(rows * Kl)), dim3(256), 0, st, P(w.dxt),
                               l == layer ? P(w.dlab) : (const double*)nullptr, masks[l], P(w.h[l - 1]), rows, Pn, Kl,
                               spare[which]);
            dz1_next = spare[which];
            which ^= 1;
        }
        if (l == layer) {
            // The tied weight's two gradients in ONE product (they were two of 1681 x 300 x 2500, each too short a K loop to
            // run well: 80 us apiece): d/dW = dz2^T h (decoder use) + x~^T dz1 (encoder use) = [dz2^T | x~^T] . [h ; dz1],
            // K = 2 rows (dz1 lies right behind h).
            hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)dlc::cdiv(K, 32), (unsigned)dlc::cdiv(rows, 32), 2), dim3(256), 0, st,
                               P(w.dz2), Kp, rows, K, P(w.tr), 2 * rows, (const double*)P(w.xt[layer]), Kp, P(w.tr) + rows);
            const int rc_ = weight_step(l, K, N, 2 * rows, h);
            if (rc_ != DLC_OK) return rc_;
        } else {
            hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)dlc::cdiv(Kl, 32), (unsigned)dlc::cdiv(rows, 32)), dim3(256), 0,
                               st, P(w.xt[l]), even_pitch(Kl), rows, Kl, P(w.tr), rows);
            const int rc_ = weight_step(l, Kl, Nl, rows, dz1);                                          // x~^T dz1
            if (rc_ != DLC_OK) return rc_;
        }
        dz1 = dz1_next;
    }

    // ---- plain gradient descent on everything the loss reached (SDAV.py:223-226), the bias gradients' column sums and the
    // loss: one launch
    {
        UpdateArgs u{};
        u.n_w = 0;
        long long blk = 0;
        for (int l = 0; l <= layer; ++l) {
            if (stepped[l]) continue;                                // (its product's epilogue took the step)
            const int m = u.n_w++;
            u.w[m] = W[l]; u.gw[m] = P(w.gw[l]); u.w_n[m] = dims[l] * dims[l + 1];
            u.w_blk0[m] = blk;
            blk += grid_for(u.w_n[m]);
        }
        u.w_blk0[u.n_w] = blk;
        u.n_b = layer + 2;
        for (int l = 0; l <= layer; ++l) {
            u.b[l] = b_enc[l]; u.b_cols[l] = dims[l + 1];
            if (dz1_of[l]) { u.gb_rows[l] = dz1_of[l]; u.b_rows[l] = rows; }
            else { u.gb_rows[l] = P(w.gbe[l]); u.b_rows[l] = 1; }
            u.b_ld[l] = dims[l + 1];
            u.b_blk0[l] = blk;
            blk += dlc::cdiv(dims[l + 1], (int64_t)32);
        }
        u.b[layer + 1] = b_dec; u.gb_rows[layer + 1] = P(w.dz2); u.b_rows[layer + 1] = rows; u.b_cols[layer + 1] = K;
        u.b_ld[layer + 1] = Kp;
        u.b_blk0[layer + 1] = blk;
        blk += dlc::cdiv(K, (int64_t)32);
        u.b_blk0[layer + 2] = blk;
        u.rows = rows; u.lr = learning_rate;
        u.cd_part = P(w.cd_part); u.nrm_part = P(w.nrm_part); u.l1_part = P(w.l1_part); u.batch = (int)batch; u.slices = fn_slices;
        u.sparse_penalty = sparse_penalty; u.consecutive_penalty = consecutive_penalty; u.loss_out = loss_out;
        hipLaunchKernelGGL(update_kernel, dim3((unsigned)(blk + 1)), dim3(256), 0, st, u);
    }
#undef GEMM
    DLC_LAUNCH_CHECK(ctx, "sdav_train_step kernels");
    return DLC_OK;
}

extern "C" int dlc_random_mask_f64(dlc_ctx* ctx, double* mask, int64_t n, int64_t n_zeros, uint64_t seed, uint64_t counter,
                                   void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!mask || n < 1 || n_zeros < 0 || n_zeros > n) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "random_mask: bad argument");
    if (n > (1ll << 26)) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "random_mask: n=%lld too large (one workgroup walks the keys)", (long long)n);
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    hipLaunchKernelGGL(random_mask_kernel, dim3(1), dim3(RM_THREADS), 0, (hipStream_t)stream, mask, (long long)n,
                       (long long)n_zeros, (unsigned long long)seed, (unsigned long long)counter);
    DLC_LAUNCH_CHECK(ctx, "random_mask_kernel");
    return DLC_OK;
}
