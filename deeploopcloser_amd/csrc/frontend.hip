// Patch front-end after key-point detection (src/sdav/input/CvInputParser.py:19-33, 49-123):
// grey conversion as cv2.imread(IMREAD_GRAYSCALE) does it, and the key-point-centred
// patch gather with the reference's clamp-inside-the-image rule and /255.0.
// HBM-bound byte gathers; one output element per thread, contiguous along the patch row.
#include "dlc_internal.h"

namespace {

__device__ __forceinline__ unsigned gray_of(unsigned r, unsigned g, unsigned b) {
    return (r * 4899u + g * 9617u + b * 1868u + 8192u) >> 14;                           // OpenCV fixed-point BT.601
}
// four pixels per thread: three dwords in, one out (a byte per lane made 3 strided byte loads per pixel: 2.1 TB/s)
__global__ __launch_bounds__(256) void rgb_to_gray_kernel(const unsigned char* __restrict__ rgb, long long n,
                                                          unsigned char* __restrict__ gray, int vec) {
    const long long quads = vec ? n / 4 : 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < quads; i += (long long)gridDim.x * 256) {
        const unsigned* src = (const unsigned*)rgb + 3 * i;
        const unsigned a = src[0], b = src[1], c = src[2];                              // r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
        const unsigned p0 = gray_of(a & 255, (a >> 8) & 255, (a >> 16) & 255);
        const unsigned p1 = gray_of(a >> 24, b & 255, (b >> 8) & 255);
        const unsigned p2 = gray_of((b >> 16) & 255, b >> 24, c & 255);
        const unsigned p3 = gray_of((c >> 8) & 255, (c >> 16) & 255, c >> 24);
        ((unsigned*)gray)[i] = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
    }
    for (long long i = quads * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        gray[i] = (unsigned char)gray_of(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
}

// CvInputParser.py:74-86 for one axis
__device__ __forceinline__ int window_lo(int c, int dim, int half) {
    const int lo = c - half, hi = c + half;
    const int fwd = lo < 0 ? -lo : 0;
    const int aux = hi - dim + 1;
    const int back = aux > 0 ? aux : 0;
    return lo - back + fwd;
}

constexpr int EP_PATCHES = 1;                              // patches per workgroup (4: 140 us against 111 -- fewer workgroups, each a serial chain)
template <typename T>
__global__ __launch_bounds__(256) void extract_patches_kernel(const unsigned char* __restrict__ gray, int H, int W,
                                                              const int* __restrict__ kp, int P, int ps, long long n_patches,
                                                              T* __restrict__ out) {
    __shared__ T by255[256];                                           // the 256 quotients once, not a division per element
    by255[threadIdx.x] = (T)threadIdx.x / (T)255.0;
    __syncthreads();
    const int step_x = 256 / ps, step_y = 256 - step_x * ps;           // e += 256 without an integer division per element
    const int total = ps * ps;
    for (int pi = 0; pi < EP_PATCHES; ++pi) {
        const long long fp = (long long)blockIdx.x * EP_PATCHES + pi;  // frame * P + patch
        if (fp >= n_patches) return;
        const long long frame = fp / P;
        const int cx = kp[fp * 2], cy = kp[fp * 2 + 1];               // (x, y) = kp.pt rounded; x walks dim 0 (:111-119)
        const int x0 = window_lo(cx, H, ps / 2), y0 = window_lo(cy, W, ps / 2);
        const unsigned char* img = gray + frame * (long long)H * W;
        T* o = out + fp * (long long)total;
        int dx = threadIdx.x / ps, dy = threadIdx.x - dx * ps;
        // eight elements of a thread at a time: all their byte loads first, then the table reads, then the stores (one
        // element after the other was seven dependent round trips -- pixel, table, store -- per thread of a 41 x 41 patch)
        for (int e0 = threadIdx.x; e0 < total; e0 += 8 * 256) {
            unsigned char v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = e0 + u * 256 < total;
                v[u] = ok ? img[(long long)(x0 + dx) * W + (y0 + dy)] : (unsigned char)0;
                dx += step_x; dy += step_y;
                if (dy >= ps) { dy -= ps; ++dx; }
            }
            T q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = by255[v[u]];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u * 256 < total) o[e0 + u * 256] = q[u];
        }
    }
}

// ---- key-point detector (NOT in the reference: it uses OpenCV-contrib's non-free SURF,
// CvInputParser.py:36-46; SURVEY section 8f-1 asks for "a simple GPU detector" instead) ----------------
// Harris corners in exact integer arithmetic: Sobel 3x3 gradients Ix, Iy (int32), structure
// tensor summed over the 5x5 window (Sxx, Syy, Sxy), response R16 = 16*(Sxx*Syy - Sxy^2) -
// (Sxx+Syy)^2 (k = 1/16, int64).  Defined for pixels at least 3 away from the border, 0 elsewhere.
__device__ __forceinline__ int px(const uint8_t* g, int W, int r, int c) { return (int)g[(long long)r * W + c]; }

// One pass from the grey bytes to the lists of corner candidates.  A workgroup owns 32 x 32 pixels: it stages the grey
// tile with a 4-pixel halo in LDS, forms the Sobel gradients of the tile + 3 (packed: ix in the low, iy in the high 16
// bits, |.| <= 1020), the 5 x 5 structure tensor and the response of the tile + 1, suppresses non-maxima over 3 x 3
// (ties: the lower row-major index wins) and writes the survivors -- (response, pixel) -- into the tile's own list of
// HT_CAP entries (survivors never touch, so a tile has at most a quarter of its pixels) and the list's length.  (As three
// kernels the gradients and the suppressed response image made round trips through 196 MB + 392 MB per 1063 frames,
// and the selection below walked the 392 MB again: 0.66 ms of the front-end's 1.03 ms.)  Same integers, same
// responses; the order inside a list is arbitrary, the selection's total order is not.
//   gradients: a thread makes four neighbours in a row from six dwords of the grey tile (not 32 byte reads);
//   tensor:    a thread walks five response rows of one column, keeping the 5-wide row sums of the nine gradient rows
//              it needs in registers; per gradient one v_perm and three v_dot2_i32_i16 on the packed pair give
//              ix^2 + iy^2, ix^2 and 2 ix iy (exact: 25 terms of at most 2 * 1020^2).
constexpr int HT_R = 32, HT_C = 32, HT_GW = 11;            // grey tile pitch in dwords (40 bytes used)
constexpr int HT_CAP = HT_R * HT_C / 4;                    // = the selection's workgroup size
constexpr int HT_SR = 5, HT_STRIPS = (HT_R + 2 + HT_SR - 1) / HT_SR;
static_assert(HT_CAP == 256 && HT_STRIPS * (HT_C + 2) <= 256, "tile shape and workgroup size go together");
typedef short dlc_s2 __attribute__((ext_vector_type(2)));
template <bool ALIGNED>
__global__ __launch_bounds__(256) void harris_candidates_kernel(const uint8_t* __restrict__ gray, int H, int W,
                                                                long long* __restrict__ lv, int* __restrict__ li,
                                                                int* __restrict__ lcount) {
    __shared__ unsigned sg[HT_R + 8][HT_GW];
    __shared__ int sgrad[HT_R + 6][HT_C + 7];
    __shared__ long long tile[HT_R + 2][HT_C + 3];
    __shared__ int nkeep;
    const long long frame = blockIdx.z;
    const long long tile_id = (frame * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int r0 = blockIdx.y * HT_R, c0 = blockIdx.x * HT_C;
    const uint8_t* g = gray + frame * H * W;
    if (threadIdx.x == 0) nkeep = 0;
    if (ALIGNED) {                                         // W and the frames' base are multiples of 4: whole dwords
        for (int e = threadIdx.x; e < (HT_R + 8) * HT_GW; e += 256) {
            const int tr = e / HT_GW, j = e - tr * HT_GW;
            const int r = r0 + tr - 4, c = c0 + 4 * j - 4;
            sg[tr][j] = (r >= 0 && r < H && c >= 0 && c < W) ? *(const unsigned*)(g + (long long)r * W + c) : 0u;
        }
    } else {
        uint8_t* sb = (uint8_t*)&sg[0][0];
        for (int e = threadIdx.x; e < (HT_R + 8) * HT_GW * 4; e += 256) {
            const int tr = e / (HT_GW * 4), tc = e - tr * (HT_GW * 4);
            const int r = r0 + tr - 4, c = c0 + tc - 4;
            sb[e] = (r >= 0 && r < H && c >= 0 && c < W) ? g[(long long)r * W + c] : (uint8_t)0;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < (HT_R + 6) * 10; e += 256) {
        const int tr = e / 10, j = e - tr * 10;               // gradient row, group of four columns
        const int r = r0 + tr - 3;
        // grey rows r - 1, r, r + 1 = tile rows tr, tr + 1, tr + 2; gradient column tc sits at grey byte tc + 1
        const unsigned long long u = sg[tr][j] | ((unsigned long long)sg[tr][j + 1] << 32);
        const unsigned long long m = sg[tr + 1][j] | ((unsigned long long)sg[tr + 1][j + 1] << 32);
        const unsigned long long d = sg[tr + 2][j] | ((unsigned long long)sg[tr + 2][j + 1] << 32);
        int colx[6], dmu[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int ub = (int)((u >> (8 * k)) & 0xff), mb = (int)((m >> (8 * k)) & 0xff), db = (int)((d >> (8 * k)) & 0xff);
            colx[k] = ub + 2 * mb + db;
            dmu[k] = db - ub;
        }
        const bool row_ok = r >= 1 && r < H - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tc = 4 * j + q, c = c0 + tc - 3;
            const int ix = colx[q + 2] - colx[q], iy = dmu[q] + 2 * dmu[q + 1] + dmu[q + 2];
            const int out = (row_ok && c >= 1 && c < W - 1) ? ((ix & 0xffff) | (int)((unsigned)iy << 16)) : 0;
            if (tc < HT_C + 6) sgrad[tr][tc] = out;
        }
    }
    __syncthreads();
    if (threadIdx.x < HT_STRIPS * (HT_C + 2)) {
        const int strip = threadIdx.x / (HT_C + 2), tc = threadIdx.x - strip * (HT_C + 2);
        const int c = c0 + tc - 1;
        int T[HT_SR + 4], X[HT_SR + 4], Y2[HT_SR + 4];        // 5-wide row sums of gradient rows HT_SR strip .. + HT_SR + 3
#pragma unroll
        for (int gr = 0; gr < HT_SR + 4; ++gr) {
            int t = 0, x = 0, y2 = 0;
            if (HT_SR * strip + gr < HT_R + 6) {
#pragma unroll
                for (int dc = 0; dc < 5; ++dc) {
                    const int v = sgrad[HT_SR * strip + gr][tc + dc];
                    const dlc_s2 pv = __builtin_bit_cast(dlc_s2, v);
                    const dlc_s2 sw = __builtin_bit_cast(dlc_s2, (int)__builtin_amdgcn_perm((unsigned)v, (unsigned)v, 0x01000302u));
                    const dlc_s2 lo = __builtin_bit_cast(dlc_s2, v & 0xffff);
                    t = __builtin_amdgcn_sdot2(pv, pv, t, false);
                    x = __builtin_amdgcn_sdot2(pv, lo, x, false);
                    y2 = __builtin_amdgcn_sdot2(pv, sw, y2, false);
                }
            }
            T[gr] = t; X[gr] = x; Y2[gr] = y2;
        }
#pragma unroll
        for (int k = 0; k < HT_SR; ++k) {
            const int tr = HT_SR * strip + k, r = r0 + tr - 1;
            if (tr < HT_R + 2) {
                long long resp = 0;
                if (r >= 3 && r < H - 3 && c >= 3 && c < W - 3) {
                    const int t = T[k] + T[k + 1] + T[k + 2] + T[k + 3] + T[k + 4];
                    const int x = X[k] + X[k + 1] + X[k + 2] + X[k + 3] + X[k + 4];
                    const int y2 = Y2[k] + Y2[k + 1] + Y2[k + 2] + Y2[k + 3] + Y2[k + 4];
                    const long long sxx = x, syy = t - x, sxy = y2 >> 1, tr2 = t;
                    resp = 16 * (sxx * syy - sxy * sxy) - tr2 * tr2;
                }
                tile[tr][tc] = resp;
            }
        }
    }
    __syncthreads();
    long long* tv = lv + tile_id * HT_CAP;
    int* ti = li + tile_id * HT_CAP;
    {
        // four pixels of one column per thread: 18 response reads for four 3 x 3 neighbourhoods
        const int tc = threadIdx.x & 31, tr0 = (threadIdx.x >> 5) * 4;
        long long nb[6][3];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int dc = 0; dc < 3; ++dc) nb[a][dc] = tile[tr0 + a][tc + dc];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + tr0 + k, c = c0 + tc;
            const long long v = nb[k + 1][1];
            bool keep = v > 0 && r < H && c < W;
#pragma unroll
            for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                for (int dc = 0; dc < 3; ++dc) {
                    if (dr == 1 && dc == 1) continue;
                    const long long u = nb[k + dr][dc];
                    const bool before = dr < 1 || (dr == 1 && dc < 1);          // the neighbour's row-major index is lower
                    if (u > v || (u == v && before)) keep = false;
                }
            if (keep) {
                const int slot = atomicAdd(&nkeep, 1);
                if (slot < HT_CAP) { tv[slot] = v; ti[slot] = r * W + c; }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the LDS counter only: the list stores need not have landed
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) lcount[tile_id] = nkeep < HT_CAP ? nkeep : HT_CAP;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_max_step(long long& v) {
    const int lo = (int)(unsigned)v, hi = (int)((unsigned long long)v >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    const long long ov = (long long)(((unsigned long long)ohi << 32) | olo);
    v = ov > v ? ov : v;
}
// The n strongest candidates of a frame, strongest first (ties: the lower pixel index).  One workgroup per frame, one
// thread per list slot: thread t looks at entry t of every tile's list and keeps its three best in registers; a round's
// winner comes from DPP steps inside a wave and the four waves' winners through LDS; the winner's owner moves its next
// one up.  Only an owner whose three are used up while it has more entries goes back to memory (its wave fetches them,
// one tile per lane: those that come after the round's winner in the order, everything before it has been a winner).
// Nothing is stored to memory inside a round -- the winners wait in LDS and go out 256 at a time: hipcc holds a wave
// until a store has read its data registers, a store round trip per round, and with the lists re-read from memory in
// every round the 30 rounds were 0.1 ms for frames that were all resident at once.
__global__ __launch_bounds__(HT_CAP) void harris_select_kernel(const long long* __restrict__ lv, const int* __restrict__ li,
                                                               const int* __restrict__ lcount, int tiles, int W, int n,
                                                               int* __restrict__ pts, long long* __restrict__ resp_out,
                                                               int* __restrict__ count) {
    __shared__ long long sv[2][4];
    __shared__ int si[2][4];
    __shared__ long long win_v[HT_CAP];
    __shared__ int win_i[HT_CAP];
    const long long frame = blockIdx.x;
    const long long* V = lv + frame * tiles * HT_CAP;
    const int* I = li + frame * tiles * HT_CAP;
    const int* C = lcount + frame * tiles;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int found = n;
    auto better = [](long long ov, int oi, long long v, int i) { return ov > v || (ov == v && oi < i); };
    auto flush = [&](int first, int upto) {                  // winners first .. upto - 1 from LDS to the outputs
        __syncthreads();
        const int j = first + tid;
        if (j < upto) {
            const int wi = win_i[tid];
            pts[(frame * n + j) * 2 + 0] = wi % W;               // cv2.KeyPoint.pt = (x = column, y = row)
            pts[(frame * n + j) * 2 + 1] = wi / W;
            resp_out[frame * n + j] = win_v[tid];
        }
        __syncthreads();
    };
    long long v1 = 0, v2 = 0, v3 = 0;                        // response and pixel of the thread's three best, best first
    int i1 = 0x7fffffff, i2 = 0x7fffffff, i3 = 0x7fffffff;
    int left = 0;                                            // entries of this thread that are neither used nor in v1..v3
    // Eight tiles at a time, all their loads in flight together: entries past a list's length are never written, such a
    // lane reads the frame's first entry instead and ignores it (the index is hidden from the compiler, which otherwise
    // turns the select back into a branch and the scan into three dependent round trips per tile: 0.09 ms).
    for (int t0 = 0; t0 < tiles; t0 += 8) {
        int cnt[8], ii[8];
        long long vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cnt[u] = C[t0 + u];         // (the lengths' array ends with eight spare words)
#pragma unroll
        for (int u = 0; u < 8; ++u) cnt[u] = t0 + u < tiles ? cnt[u] : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int e = tid < cnt[u] ? (t0 + u) * HT_CAP + tid : 0;
            asm volatile("" : "+v"(e));
            vv[u] = V[e];
            ii[u] = I[e];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (tid < cnt[u]) {
                const long long v = vv[u];
                const int i = ii[u];
                ++left;
                if (better(v, i, v3, i3)) {
                    if (better(v, i, v2, i2)) {
                        v3 = v2; i3 = i2;
                        if (better(v, i, v1, i1)) { v2 = v1; i2 = i1; v1 = v; i1 = i; }
                        else { v2 = v; i2 = i; }
                    } else { v3 = v; i3 = i; }
                }
            }
    }
    left -= (v1 > 0) + (v2 > 0) + (v3 > 0);
    for (int j = 0; j < n; ++j) {
        // the wave's largest response by DPP steps (no LDS round trip per step as with ds_bpermute): a scan inside the
        // rows of 16 lanes, then rows 0 -> 1, 2 -> 3 and lanes 0..31 -> 32..63; lane 63 holds the maximum.  Lanes
        // without a source take 0, which no response is below.
        long long m = v1;
        dpp_max_step<0x111, 0xf>(m); dpp_max_step<0x112, 0xf>(m); dpp_max_step<0x114, 0xf>(m); dpp_max_step<0x118, 0xf>(m);
        dpp_max_step<0x142, 0xa>(m); dpp_max_step<0x143, 0xc>(m);
        const unsigned mlo = __builtin_amdgcn_readlane((unsigned)m, 63);
        const unsigned mhi = __builtin_amdgcn_readlane((unsigned)((unsigned long long)m >> 32), 63);
        const long long wave_v = (long long)(((unsigned long long)mhi << 32) | mlo);
        int wave_i = 0x7fffffff;                               // the lowest pixel among the lanes that hold it (scalar)
        for (unsigned long long tied = __ballot(v1 == wave_v && wave_v > 0); tied; tied &= tied - 1) {
            const int x = __builtin_amdgcn_readlane(i1, __ffsll((long long)tied) - 1);
            wave_i = x < wave_i ? x : wave_i;
        }
        if (lane == 0) { sv[j & 1][w] = wave_v; si[j & 1][w] = wave_i; }
        __syncthreads();
        long long v = sv[j & 1][0];
        int i = si[j & 1][0];
#pragma unroll
        for (int o = 1; o < 4; ++o) {
            const long long ov = sv[j & 1][o];
            const int oi = si[j & 1][o];
            if (better(ov, oi, v, i)) { v = ov; i = oi; }
        }
        const long long wv = v;
        const int wi = i;
        if (wv <= 0) { found = j; break; }       // uniform: every thread holds the same winner
        if (tid == 0) { win_v[j & (HT_CAP - 1)] = wv; win_i[j & (HT_CAP - 1)] = wi; }
        if ((j & (HT_CAP - 1)) == HT_CAP - 1) flush(j - (HT_CAP - 1), j + 1);
        const bool own = i1 == wi;                               // pixels are unique: one owner in the workgroup
        if (own) { v1 = v2; i1 = i2; v2 = v3; i2 = i3; v3 = 0; i3 = 0x7fffffff; }
        const unsigned long long again = __ballot(own && v1 <= 0 && left > 0);
        if (again) {                                             // the owner's wave: the best of its entries after the winner
            const int oslot = w * 64 + __ffsll((long long)again) - 1;
            long long rv = 0;
            int ri = 0x7fffffff;
            for (int t = lane; t < tiles; t += 64)
                if (oslot < C[t]) {
                    const long long cv = V[t * HT_CAP + oslot];
                    const int ci = I[t * HT_CAP + oslot];
                    if (better(wv, wi, cv, ci) && better(cv, ci, rv, ri)) { rv = cv; ri = ci; }
                }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const long long ov = __shfl_xor(rv, o);
                const int oi = __shfl_xor(ri, o);
                if (better(ov, oi, rv, ri)) { rv = ov; ri = oi; }
            }
            if (tid == oslot) { v1 = rv; i1 = ri; left -= 1; }
        }
    }
    const int done = found < n ? found : n;
    if (done & (HT_CAP - 1)) flush(done & ~(HT_CAP - 1), done);
    for (int j = found + tid; j < n; j += HT_CAP) {
        pts[(frame * n + j) * 2 + 0] = -1;
        pts[(frame * n + j) * 2 + 1] = -1;
        resp_out[frame * n + j] = 0;
    }
    if (tid == 0) count[frame] = found;
}

}  // namespace

namespace {
inline int64_t harris_tiles(int H, int W) { return dlc::cdiv((int64_t)H, (int64_t)HT_R) * dlc::cdiv((int64_t)W, (int64_t)HT_C); }
}

extern "C" size_t dlc_harris_keypoints_workspace_bytes(int64_t frames, int H, int W) {
    if (frames < 1 || H < 7 || W < 7) return 0;
    const size_t lists = (size_t)frames * (size_t)harris_tiles(H, W);       // HT_CAP x (response, pixel) and a length each
    return dlc::align_up(lists * HT_CAP * 8, 256) + dlc::align_up(lists * HT_CAP * 4, 256) + dlc::align_up((lists + 8) * 4, 256);
}

extern "C" int dlc_harris_keypoints_u8(dlc_ctx* ctx, const uint8_t* gray, int64_t frames, int H, int W, int n,
                                       int32_t* points, int64_t* responses, int32_t* counts, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!gray || !points || !responses || !counts || frames < 1 || n < 1)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "harris_keypoints: bad argument");
    if (H < 7 || W < 7 || (long long)H * W > 0x7fffffffll || frames > 65535 || H > 65535 * HT_R)
        return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "harris_keypoints: %lld frames of %dx%d unsupported", (long long)frames, H, W);
    const size_t need = dlc_harris_keypoints_workspace_bytes(frames, H, W);
    if (!workspace || workspace_bytes < need)
        return dlc::fail(ctx, DLC_ERR_WORKSPACE, "harris_keypoints: workspace %zu < %zu bytes", workspace_bytes, need);
    if ((uintptr_t)workspace & 255) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "harris_keypoints: workspace must be 256-byte aligned");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    hipStream_t st = (hipStream_t)stream;
    // the tiles' candidate lists: responses, pixels, lengths (every length is written by its tile's workgroup)
    const size_t lists = (size_t)frames * (size_t)harris_tiles(H, W);
    long long* lv = (long long*)workspace;
    int* li = (int*)((char*)lv + dlc::align_up(lists * HT_CAP * 8, 256));
    int* lcount = (int*)((char*)li + dlc::align_up(lists * HT_CAP * 4, 256));
    dim3 tiles((unsigned)dlc::cdiv((int64_t)W, (int64_t)HT_C), (unsigned)dlc::cdiv((int64_t)H, (int64_t)HT_R), (unsigned)frames);
    if ((W & 3) == 0 && ((uintptr_t)gray & 3) == 0)
        hipLaunchKernelGGL(harris_candidates_kernel<true>, tiles, dim3(256), 0, st, gray, H, W, lv, li, lcount);
    else
        hipLaunchKernelGGL(harris_candidates_kernel<false>, tiles, dim3(256), 0, st, gray, H, W, lv, li, lcount);
    DLC_LAUNCH_CHECK(ctx, "harris_candidates_kernel");
    hipLaunchKernelGGL(harris_select_kernel, dim3((unsigned)frames), dim3(HT_CAP), 0, st, (const long long*)lv, (const int*)li,
                       (const int*)lcount, (int)harris_tiles(H, W), W, n, (int*)points, (long long*)responses, (int*)counts);
    DLC_LAUNCH_CHECK(ctx, "harris_select_kernel");
    return DLC_OK;
}

extern "C" int dlc_rgb_to_gray_u8(dlc_ctx* ctx, const uint8_t* rgb, int64_t n_pixels, uint8_t* gray, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!rgb || !gray || n_pixels < 1) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "rgb_to_gray: bad argument");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    const int vec = (((uintptr_t)rgb | (uintptr_t)gray) & 3) == 0;
    long long blocks = dlc::cdiv(vec ? dlc::cdiv(n_pixels, (int64_t)4) : n_pixels, 256);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(rgb_to_gray_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rgb,
                       (long long)n_pixels, gray, vec);
    DLC_LAUNCH_CHECK(ctx, "rgb_to_gray_kernel");
    return DLC_OK;
}

extern "C" int dlc_extract_patches(dlc_ctx* ctx, const uint8_t* gray, int64_t frames, int H, int W,
                                   const int32_t* key_points, int P, int patch_size, int out_dtype, void* out,
                                   void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!gray || !key_points || !out || frames < 1 || P < 1) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "extract_patches: bad argument");
    if (patch_size < 1 || patch_size % 2 == 0)
        return dlc::fail(ctx, DLC_ERR_BAD_ARG, "Invalid patch size. Patch size must be an odd number");   // CvInputParser.py:61-62
    if (H < patch_size || W < patch_size)
        return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "extract_patches: image %dx%d smaller than the %d patch", H, W, patch_size);
    if (frames * P > 0x7fffffffll) return dlc::fail(ctx, DLC_ERR_BAD_SHAPE, "extract_patches: too many patches");
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    dim3 grid((unsigned)dlc::cdiv(frames * P, (int64_t)EP_PATCHES));
    if (out_dtype == DLC_F64)
        hipLaunchKernelGGL(extract_patches_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, gray, H, W,
                           (const int*)key_points, P, patch_size, (long long)(frames * P), (double*)out);
    else if (out_dtype == DLC_F32)
        hipLaunchKernelGGL(extract_patches_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, gray, H, W,
                           (const int*)key_points, P, patch_size, (long long)(frames * P), (float*)out);
    else
        return dlc::fail(ctx, DLC_ERR_UNSUPPORTED, "extract_patches: out dtype %d", out_dtype);
    DLC_LAUNCH_CHECK(ctx, "extract_patches_kernel");
    return DLC_OK;
}
