// scene.hip — the reference's scene back-end on the GPU: height map with sigmoid "bumps" and
// ball centroids (/root/reference/shaders/pt_cloud.comp), world positions and 8-neighbour edge lengths
// (/root/reference/shaders/pt_cloud_weights.comp), driven as /root/reference/src/scene.rs:147-331 (append_scene)
// drives its two Vulkan dispatches of [80,60,1] x 8x8 over a 640x480 frame (scene.rs:245,:256).
//
// What is computed is the deterministic reading frozen in DESIGN.md §Scene and restated in oracle/orc_scene.c (the
// shaders as written race - store_ball, barrier() used as a grid barrier - and call pow() where GLSL leaves it
// undefined): every stage completes over the whole frame before the next starts (one launch per stage); texel (x, y)
// is read for pixel (x, y); squares are products; a bump whose sigmoid base is not positive adds nothing; ball
// centroids are exact integer means. Integer outputs and every float are bit-identical to the oracle: one IEEE
// operation per operator (-ffp-contract=off; sqrtf and / are correctly rounded in HIP by default, unlike __fsqrt_rn,
// which maps to the native approximate instruction).
// Atomic-bound byte work. The height map is a max over ~150 M bump taps per 640 x 480 frame; the shader (and rounds 2-4 here)
// gives every pixel a lane that walks its 400 / 1 600 taps through imageAtomicMax in global memory (2.0-2.3 ms per frame). Round 5
// (scene_cloud_strips): the map is PRIVATISED in LDS - a workgroup owns a strip of 16 pixel columns x a band of 64 map rows, stamps
// every tap that lands there with ds_max_u32 (a wave per pixel, a lane per tap: uniform tap counts, coalesced table reads,
// consecutive lanes on consecutive LDS words) and merges its image into the global map with one atomicMax per touched cell. A max
// is order-free, so the result is the shader's imageAtomicMax result bit for bit. The other stages keep one lane per pixel in
// 8x8 workgroups, the dispatch the reference uses (scene.rs:245,:256).
#include <hip/hip_runtime.h>
#include <string.h>

#include <string>

#include "yh_internal.h"

using namespace yh;

namespace {

#define SC_MAX_DEPTH 4000.0f
#define SC_TAN_HALF_YFOV 0.55430907f
#define SC_TAN_HALF_XFOV 0.9489646f
#define SC_BOT_AVOID 100.0f
#define SC_BOT_NORM 20
#define SC_TERRAIN_NORM 10
#define SC_BUMP_ERR 0.1f

struct SceneParams {
    const uint16_t* depth;        // [H][W]
    const uint8_t* cls_id;        // [H][W][2] (class, id), or nullptr when `frame` is given
    const uint32_t* frame;        // [H][W] packed pixels as classify leaves them
    int frame_mode;               // with `frame`: 0 = low 16 bits as src/scene.rs:93 reads them, 1 = class bits 31-24, id bits 23-16
    int W, H, mode;
    int band_h;                   // map rows per workgroup of scene_cloud_strips (<= SC_BH; chosen so that the grid is one round of the chip)
    uint32_t* map;                // [H][W]
    float4 *world, *conn0, *conn1;
    long long* ball_acc;          // [3][100]: sum x, sum y, count
    float4* balls;                // [100]
    // the bump profiles, tabulated once per handle: a tap's height depends only on (val, dx, dy), and val is the pixel's
    // ROW for terrain (pt_cloud.comp:116) or the constant 100 for robots (:122)
    const uint32_t* terrain_tab;  // [H][ly 20][lx 20]
    const uint32_t* robot_tab;    // [ly 40][lx 40]
};

// height of one bump tap (pt_cloud.comp:55-73): val / (1 + C_1^(C_2 prox - 1)), truncated; 0 where the shader's pow() is undefined
__device__ __forceinline__ uint32_t bump_tap(float val, int L, int lx, int ly) {
    const float C1 = __fsub_rn(__fdiv_rn(val, SC_BUMP_ERR), 1.0f), C2 = __fdiv_rn(2.0f, (float)L);
    if (!(C1 > 0.0f)) return 0u;
    const float logC1 = spec_logf(C1);   // pow(C_1, e) = exp(e * log(C_1))
    const int dx = L - lx, dy = L - ly;  // pos - loc with loc = pos - L + (lx, ly)
    const float prox = __builtin_sqrtf((float)(dx * dx + dy * dy));
    const float e = __fsub_rn(__fmul_rn(C2, prox), 1.0f);
    const float y_add = __fdiv_rn(val, __fadd_rn(1.0f, spec_expf(__fmul_rn(e, logC1))));
    return y_add >= 1.0f ? (uint32_t)y_add : 0u;
}

// one lane per table entry: terrain [H][ly][lx] with val = row, robot [ly][lx] with val = 100 (lx fastest: consecutive lanes of
// scene_cloud_strips take consecutive lx = consecutive map columns)
__global__ __launch_bounds__(256) void scene_tables(uint32_t* terrain, uint32_t* robot, int H) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int nt = H * 4 * SC_TERRAIN_NORM * SC_TERRAIN_NORM, nr = 4 * SC_BOT_NORM * SC_BOT_NORM;
    if (t < nt) {
        const int y = t / (4 * SC_TERRAIN_NORM * SC_TERRAIN_NORM), r = t % (4 * SC_TERRAIN_NORM * SC_TERRAIN_NORM);
        terrain[t] = bump_tap((float)y, SC_TERRAIN_NORM, r % (2 * SC_TERRAIN_NORM), r / (2 * SC_TERRAIN_NORM));
    } else if (t - nt < nr) {
        const int r = t - nt;
        robot[r] = bump_tap(SC_BOT_AVOID, SC_BOT_NORM, r % (2 * SC_BOT_NORM), r / (2 * SC_BOT_NORM));
    }
}

// pt_cloud.comp main (:84-123) and its bump() (:44-76), privatised. Geometry of one workgroup (512 lanes, 8 waves):
//   strip   pixel columns [c0, c0 + 16): a pixel (x, y) stamps around (nx, ny) = (x, H - dic(depth)), i.e. map columns
//           x - L .. x + L - 1 with L <= 20: the strip's taps land in map columns [c0 - 20, c0 + 36) - the LDS image's 56 columns;
//   band    map rows [r0, r0 + band_h): every workgroup of a strip walks ALL of the strip's pixels (16 x H: cheap) and stamps the taps
//           of each bump that fall into its own band; band_h = 64 (320 workgroups at 640 x 480; a grid of ONE round - 80-row bands - measured slower);
//           the image is 21 KB.
// A wave takes four pixel rows at a time: its 64 lanes compute the 4 x 16 pixels (the shader's arithmetic, one IEEE
// operation per operator), then the wave stamps the bump pixels one at a time (ballot + readlane: wave-uniform target),
// lane t owning taps t, t + 64, ... of the bump. The tap HEIGHTS sit in registers: a terrain tap depends on the pixel's row only
// (pt_cloud.comp:116), so the row's 400-entry table is loaded once per row (7 coalesced loads, only if some pixel of the row hits
// the band) and serves its 16 pixels; the robot table (1 600 entries, a constant) is loaded once per workgroup (25 registers). The
// stamping loop therefore has no memory read: per tap a range test and one ds_max_u32 (row pitch 84 words). A tap of height 0
// changes nothing and is not issued. (First version of this kernel, a wave per pixel with the table read from global memory inside
// the tap loop: 0.56-0.65 ms per frame against 2.0-2.3 for the global-atomic form; it waited for one L2 round trip per 64 taps.)
// Ball pixels add their position to 64-bit sums in LDS (band 0 only: once per pixel), flushed once per workgroup.
#define SC_CW 16
#define SC_BH 96   // most map rows a workgroup's LDS image holds; the launch picks band_h <= SC_BH (yh_scene_create)
#define SC_HALO 20
#define SC_LDW 84   // 56 columns used; 84 = 64 + 20: a wave's 3.2 consecutive tap rows of a terrain bump fall on 64 different banks
#define SC_TT (4 * SC_TERRAIN_NORM * SC_TERRAIN_NORM)   // 400 taps
#define SC_RT (4 * SC_BOT_NORM * SC_BOT_NORM)           // 1600 taps
#define SC_TK ((SC_TT + 63) / 64)                       // 7 taps per lane
#define SC_RK (SC_RT / 64)                              // 25 taps per lane
__global__ __launch_bounds__(512) void scene_cloud_strips(const SceneParams p) {
    __shared__ uint32_t img[SC_BH * SC_LDW];
    __shared__ unsigned long long ball[300];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.x * SC_CW, r0 = blockIdx.y * p.band_h;
    const bool do_balls = blockIdx.y == 0;
    const int ncell = p.band_h * SC_LDW;
    for (int i = tid; i < ncell; i += 512) img[i] = 0u;
    if (do_balls) for (int i = tid; i < 300; i += 512) ball[i] = 0ull;
    uint32_t rt[SC_RK];   // the robot bump, this lane's taps
#pragma unroll
    for (int k = 0; k < SC_RK; ++k) rt[k] = p.robot_tab[lane + 64 * k];
    __syncthreads();
    const int cw = min(SC_CW, p.W - c0);
    const int band_lo = max(r0, 1), band_hi = min(r0 + p.band_h, p.H - 1);   // rows y with 0 < y < H - 1 inside the band
    const int ibase = -r0 * SC_LDW - (c0 - SC_HALO);                       // img[ibase + y * SC_LDW + x] = the cell of map (x, y)
    // FOUR pixel rows per wave and iteration (lane = 16 (row in the group) + column): the depth / class loads of 64 pixels are one
    // latency, not four, and the four rows' terrain tables are requested together before the first stamp (the first version of this loop
    // took one row at a time: 60 exposed round trips per wave on a grid of ~1.25 workgroups per CU)
    const int lr = lane >> 4, lc = lane & 15;
    for (int y4 = 4 * wave; y4 < p.H; y4 += 32) {
        const int y = y4 + lr;
        int nx = 0, ny = 0, L = 0;
        if (lc < cw && y < p.H) {
            const int x = c0 + lc;
            const size_t i = (size_t)y * p.W + x;
            const float ty = __fdiv_rn(__fmul_rn(__fmul_rn(SC_TAN_HALF_YFOV, (float)y), 2.0f), (float)p.H);
            const float tx = __fdiv_rn(__fmul_rn(__fmul_rn(SC_TAN_HALF_XFOV, (float)x), 2.0f), (float)p.W);
            const float cy = __fdiv_rn(1.0f, __builtin_sqrtf(__fadd_rn(1.0f, __fmul_rn(ty, ty))));
            const float cx = __fdiv_rn(1.0f, __builtin_sqrtf(__fadd_rn(1.0f, __fmul_rn(tx, tx))));
            const float d = __fmul_rn(__fmul_rn((float)p.depth[i], cy), cx);
            const int dic = (int)__fdiv_rn(__fmul_rn((float)p.H, d), SC_MAX_DEPTH);
            int cls, id;
            if (p.cls_id) { cls = p.cls_id[2 * i]; id = p.cls_id[2 * i + 1]; }
            else {
                const uint32_t px = p.frame[i];
                if (p.frame_mode == 0) { cls = (int)(px & 0xFFu); id = (int)((px >> 8) & 0xFFu); }   // `as u16` then R8G8 (scene.rs:93, :198)
                else { cls = (int)(px >> 24); id = (int)((px >> 16) & 0xFFu); }
            }
            int action = cls;
            if (action > 1) action = action - 1;
            nx = x; ny = p.H - dic;
            if (action == 0) L = SC_TERRAIN_NORM;
            else if (action == 2) {
                if (do_balls && id < 100) {
                    __hip_atomic_fetch_add(&ball[id], (unsigned long long)(long long)nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&ball[100 + id], (unsigned long long)(long long)ny, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&ball[200 + id], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else L = SC_BOT_NORM;
        }
        const bool hit = L > 0 && max(ny - L, band_lo) < min(ny + L, band_hi);   // this lane's bump meets the band
        const unsigned long long hit_t = __ballot(hit && L == SC_TERRAIN_NORM);
        unsigned long long todo_r = __ballot(hit && L == SC_BOT_NORM);
        if (hit_t) {
            uint32_t tt[4][SC_TK];   // the terrain bumps of the four rows, this lane's taps (a row without a hit is not fetched)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if ((hit_t >> (16 * r)) & 0xFFFFull) {
                    const uint32_t* trow = p.terrain_tab + (size_t)(y4 + r) * SC_TT;
#pragma unroll
                    for (int k = 0; k < SC_TK; ++k) tt[r][k] = lane + 64 * k < SC_TT ? trow[lane + 64 * k] : 0u;
                } else {
#pragma unroll
                    for (int k = 0; k < SC_TK; ++k) tt[r][k] = 0u;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned long long todo_t = hit_t & (0xFFFFull << (16 * r));
                while (todo_t) {
                    const int j = __ffsll((long long)todo_t) - 1;
                    todo_t &= todo_t - 1;
                    const int x0 = __builtin_amdgcn_readlane(nx, j) - SC_TERRAIN_NORM, y0 = __builtin_amdgcn_readlane(ny, j) - SC_TERRAIN_NORM;
#pragma unroll
                    for (int k = 0; k < SC_TK; ++k) {
                        const int t = lane + 64 * k, ly = t / (2 * SC_TERRAIN_NORM), lx = t - ly * (2 * SC_TERRAIN_NORM);
                        const int yy = y0 + ly, xx = x0 + lx;
                        if (tt[r][k] && yy >= band_lo && yy < band_hi && xx > 0 && xx < p.W - 1)
                            __hip_atomic_fetch_max(&img[ibase + yy * SC_LDW + xx], tt[r][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        while (todo_r) {
            const int j = __ffsll((long long)todo_r) - 1;
            todo_r &= todo_r - 1;
            const int x0 = __builtin_amdgcn_readlane(nx, j) - SC_BOT_NORM, y0 = __builtin_amdgcn_readlane(ny, j) - SC_BOT_NORM;
#pragma unroll
            for (int k = 0; k < SC_RK; ++k) {
                const int t = lane + 64 * k, ly = t / (2 * SC_BOT_NORM), lx = t - ly * (2 * SC_BOT_NORM);
                const int yy = y0 + ly, xx = x0 + lx;
                if (rt[k] && yy >= band_lo && yy < band_hi && xx > 0 && xx < p.W - 1)
                    __hip_atomic_fetch_max(&img[ibase + yy * SC_LDW + xx], rt[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < ncell; i += 512) {
        const uint32_t v = img[i];
        if (v) {
            const int row = i / SC_LDW, col = i - row * SC_LDW;
            atomicMax(p.map + (size_t)(r0 + row) * p.W + (c0 - SC_HALO + col), v);
        }
    }
    if (do_balls)
        for (int i = tid; i < 300; i += 512)
            if (ball[i]) atomicAdd((unsigned long long*)p.ball_acc + i, ball[i]);
}

__global__ void scene_balls(const SceneParams p) {
    const int k = threadIdx.x;
    if (k >= 100) return;
    const long long sx = p.ball_acc[k], sy = p.ball_acc[100 + k], n = p.ball_acc[200 + k];
    p.balls[k] = make_float4(n ? (float)((double)sx / (double)n) : 0.0f, n ? (float)((double)sy / (double)n) : 0.0f, (float)n, 0.0f);
}

// pt_cloud_weights.comp stage 1 (:57-87)
__global__ __launch_bounds__(64) void scene_world(const SceneParams p) {
    const int x = blockIdx.x * 8 + threadIdx.x, y = blockIdx.y * 8 + threadIdx.y;
    if (x >= p.W || y >= p.H) return;
    const size_t i = (size_t)y * p.W + x;
    p.world[i] = make_float4((float)x, (float)p.map[i], (float)y, 0.0f);
}

// stage 2 (:91-111): r (x, y+1), g (x-1, y+1), b (x-1, y), a (x-1, y-1)
__global__ __launch_bounds__(64) void scene_conn1(const SceneParams p) {
    const int x = blockIdx.x * 8 + threadIdx.x, y = blockIdx.y * 8 + threadIdx.y;
    if (x >= p.W || y >= p.H) return;
    const size_t i = (size_t)y * p.W + x;
    const float4 me = p.world[i];
    const int ox[4] = { 0, -1, -1, -1 }, oy[4] = { 1, 1, 0, -1 };
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int qx = x + ox[k], qy = y + oy[k];
        v[k] = -1.0f;
        if (qx >= 0 && qx < p.W && qy >= 0 && qy < p.H) {
            // STRICT: pack(x, y) = float((x << 16) & y) is 0 for every pixel (pt_cloud_weights.comp:32), so unpack()
            // returns world(0, 0) whoever the neighbour is; SANE: the neighbour's own position
            const float4 o = p.world[p.mode == 0 ? 0 : (size_t)qy * p.W + qx];
            const float dx = __fsub_rn(me.x, o.x), dy = __fsub_rn(me.y, o.y), dz = __fsub_rn(me.z, o.z);
            v[k] = __builtin_sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        }
    }
    p.conn1[i] = make_float4(v[0], v[1], v[2], v[3]);
}

// stage 3 (:115-123): r (x, y-1), g (x+1, y-1), b (x+1, y), a (x+1, y+1)
__global__ __launch_bounds__(64) void scene_conn0(const SceneParams p) {
    const int x = blockIdx.x * 8 + threadIdx.x, y = blockIdx.y * 8 + threadIdx.y;
    if (x >= p.W || y >= p.H) return;
    const size_t i = (size_t)y * p.W + x;
    const int ox[4] = { 0, 1, 1, 1 }, oy[4] = { -1, -1, 0, 1 };
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int qx = x + ox[k], qy = y + oy[k];
        v[k] = -1.0f;
        if (qx >= 0 && qx < p.W && qy >= 0 && qy < p.H) {
            const float4 o = p.conn1[(size_t)qy * p.W + qx];
            v[k] = k == 0 ? o.x : (k == 1 ? o.y : (k == 2 ? o.z : o.w));
        }
    }
    p.conn0[i] = make_float4(v[0], v[1], v[2], v[3]);
}

thread_local std::string g_scene_create_error;

}  // namespace

struct yh_scene {
    int dev = 0, W = 0, H = 0, band_h = 64;
    hipStream_t stream = nullptr;
    hipEvent_t copied = nullptr;
    std::string err;
    uint16_t* depth = nullptr;
    uint8_t* cls_id = nullptr;
    uint32_t* frame = nullptr;
    uint32_t* map = nullptr;
    float4 *world = nullptr, *conn0 = nullptr, *conn1 = nullptr, *balls = nullptr;
    long long* ball_acc = nullptr;
    uint32_t *terrain_tab = nullptr, *robot_tab = nullptr;
    bool ran = false;
    // what the last append ran on (yh_scene_time replays exactly this)
    const uint8_t* last_cls = nullptr;
    const uint32_t* last_frame = nullptr;
    int last_frame_mode = 0, last_mode = 0;
    int fail(int code, const std::string& m) { err = m; return code; }
};

#define SCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (h)->fail(YH_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

namespace {
int run_scene(yh_scene* h, const uint16_t* depth_dev, const uint8_t* cls_dev, const uint32_t* frame_dev, int frame_mode, int mode) {
    SceneParams p;
    p.depth = depth_dev; p.cls_id = cls_dev; p.frame = frame_dev; p.frame_mode = frame_mode;
    p.W = h->W; p.H = h->H; p.mode = mode; p.band_h = h->band_h;
    p.terrain_tab = h->terrain_tab; p.robot_tab = h->robot_tab;
    p.map = h->map; p.world = h->world; p.conn0 = h->conn0; p.conn1 = h->conn1; p.ball_acc = h->ball_acc; p.balls = h->balls;
    const size_t npx = (size_t)h->W * h->H;
    SCHK(h, hipMemsetAsync(h->map, 0, npx * 4, h->stream));
    SCHK(h, hipMemsetAsync(h->ball_acc, 0, 300 * sizeof(long long), h->stream));
    const dim3 grid((unsigned)((h->W + 7) / 8), (unsigned)((h->H + 7) / 8)), block(8, 8);   // [80,60,1] x 8x8 at 640x480 (scene.rs:245,:256)
    hipLaunchKernelGGL(scene_cloud_strips, dim3((unsigned)((h->W + SC_CW - 1) / SC_CW), (unsigned)((h->H + h->band_h - 1) / h->band_h)), dim3(512), 0, h->stream, p);
    hipLaunchKernelGGL(scene_balls, dim3(1), dim3(128), 0, h->stream, p);
    hipLaunchKernelGGL(scene_world, grid, block, 0, h->stream, p);
    hipLaunchKernelGGL(scene_conn1, grid, block, 0, h->stream, p);
    hipLaunchKernelGGL(scene_conn0, grid, block, 0, h->stream, p);
    SCHK(h, hipGetLastError());
    h->ran = true;
    h->last_cls = cls_dev; h->last_frame = frame_dev; h->last_frame_mode = frame_mode; h->last_mode = mode;
    return YH_OK;
}

// copy_from_slice semantics for host inputs (as yh_set_input_u8): the caller's buffers are free again when the call
// returns. The runtime has staged a copy from PAGEABLE memory by then; from pinned / registered memory the DMA is still
// reading, so wait for the copies (not for the kernels behind them: the event sits between the two).
int host_sources_done(yh_scene* h, const void* a, const void* b) {
    bool pinned = false;
    for (const void* p : { a, b }) {
        if (!p) continue;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost) pinned = true;
        else (void)hipGetLastError();   // (an unregistered pointer is reported as an error: not one)
    }
    if (!pinned) return YH_OK;
    SCHK(h, hipEventRecord(h->copied, h->stream));
    SCHK(h, hipEventSynchronize(h->copied));
    return YH_OK;
}
}  // namespace

extern "C" {

const char* yh_scene_last_error(const yh_scene* h) { return h ? h->err.c_str() : g_scene_create_error.c_str(); }

int yh_scene_create(int32_t device, int32_t width, int32_t height, yh_scene** out) {
    if (!out) { g_scene_create_error = "null argument"; return YH_EINVAL; }
    *out = nullptr;
    if (width < 3 || height < 3 || width > 8192 || height > 8192) { g_scene_create_error = "frame size out of range"; return YH_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_scene_create_error = "no such HIP device (no CPU fallback)"; return YH_EHIP; }
    yh_scene* h = new yh_scene();
    h->dev = device; h->W = width; h->H = height;
    h->band_h = 64;   // map rows per workgroup of the stamping kernel. Measured at 640 x 480 (ms per frame, terrain only / robots + balls): 32 rows
                      // 0.238 / 0.270, 64 rows 0.229 / 0.262, 80 rows (one round of the chip: 240 workgroups) 0.250 / 0.304
    const size_t npx = (size_t)width * height;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->copied, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&h->depth, npx * 2);
    if (e == hipSuccess) e = hipMalloc((void**)&h->cls_id, npx * 2);
    if (e == hipSuccess) e = hipMalloc((void**)&h->frame, npx * 4);
    // (input images start defined: nothing the handle can be asked to run ever reads uninitialised device memory)
    if (e == hipSuccess) e = hipMemsetAsync(h->depth, 0, npx * 2, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->cls_id, 0, npx * 2, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->frame, 0, npx * 4, h->stream);
    if (e == hipSuccess) e = hipMalloc((void**)&h->map, npx * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&h->world, npx * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&h->conn0, npx * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&h->conn1, npx * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&h->balls, 100 * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&h->ball_acc, 300 * sizeof(long long));
    const size_t nt = (size_t)height * 4 * SC_TERRAIN_NORM * SC_TERRAIN_NORM, nr = 4 * SC_BOT_NORM * SC_BOT_NORM;
    if (e == hipSuccess) e = hipMalloc((void**)&h->terrain_tab, nt * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&h->robot_tab, nr * 4);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(scene_tables, dim3((unsigned)((nt + nr + 255) / 256)), dim3(256), 0, h->stream, h->terrain_tab, h->robot_tab, height);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { g_scene_create_error = std::string("scene setup: ") + hipGetErrorString(e); yh_scene_destroy(h); return YH_EHIP; }
    *out = h;
    return YH_OK;
}

void yh_scene_destroy(yh_scene* h) {
    if (!h) return;
    hipSetDevice(h->dev);
    if (h->stream) hipStreamSynchronize(h->stream);
    void* bufs[] = { h->depth, h->cls_id, h->frame, h->map, h->world, h->conn0, h->conn1, h->balls, h->ball_acc, h->terrain_tab, h->robot_tab };
    for (void* b : bufs) if (b) hipFree(b);
    if (h->copied) hipEventDestroy(h->copied);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int yh_scene_append(yh_scene* h, const uint16_t* depth_host, const uint8_t* class_id_host, int32_t mode) {
    if (!h || !depth_host || !class_id_host) return YH_EINVAL;
    if (mode != YH_COMPAT_STRICT && mode != YH_COMPAT_SANE) return h->fail(YH_EINVAL, "bad compat mode");
    SCHK(h, hipSetDevice(h->dev));
    const size_t npx = (size_t)h->W * h->H;
    SCHK(h, hipMemcpyAsync(h->depth, depth_host, npx * 2, hipMemcpyHostToDevice, h->stream));
    SCHK(h, hipMemcpyAsync(h->cls_id, class_id_host, npx * 2, hipMemcpyHostToDevice, h->stream));
    const int rc = host_sources_done(h, depth_host, class_id_host);
    if (rc) return rc;
    return run_scene(h, h->depth, h->cls_id, nullptr, 0, mode);
}

int yh_scene_append_classified(yh_scene* h, const uint16_t* depth_host, const uint32_t* frame, int32_t frame_on_device, int32_t mode) {
    if (!h || !depth_host || !frame) return YH_EINVAL;
    if (mode != YH_COMPAT_STRICT && mode != YH_COMPAT_SANE) return h->fail(YH_EINVAL, "bad compat mode");
    SCHK(h, hipSetDevice(h->dev));
    const size_t npx = (size_t)h->W * h->H;
    SCHK(h, hipMemcpyAsync(h->depth, depth_host, npx * 2, hipMemcpyHostToDevice, h->stream));
    const uint32_t* fdev = frame;
    if (!frame_on_device) { SCHK(h, hipMemcpyAsync(h->frame, frame, npx * 4, hipMemcpyHostToDevice, h->stream)); fdev = h->frame; }
    const int rc = host_sources_done(h, depth_host, frame_on_device ? nullptr : frame);
    if (rc) return rc;
    return run_scene(h, h->depth, nullptr, fdev, mode == YH_COMPAT_STRICT ? 0 : 1, mode);
}

int yh_scene_read(yh_scene* h, uint32_t* map, float* world, float* conn0, float* conn1, float* balls) {
    if (!h) return YH_EINVAL;
    if (!h->ran) return h->fail(YH_ESTATE, "no frame has been appended");
    SCHK(h, hipSetDevice(h->dev));
    const size_t npx = (size_t)h->W * h->H;
    if (map) SCHK(h, hipMemcpyAsync(map, h->map, npx * 4, hipMemcpyDeviceToHost, h->stream));
    if (world) SCHK(h, hipMemcpyAsync(world, h->world, npx * 16, hipMemcpyDeviceToHost, h->stream));
    if (conn0) SCHK(h, hipMemcpyAsync(conn0, h->conn0, npx * 16, hipMemcpyDeviceToHost, h->stream));
    if (conn1) SCHK(h, hipMemcpyAsync(conn1, h->conn1, npx * 16, hipMemcpyDeviceToHost, h->stream));
    if (balls) SCHK(h, hipMemcpyAsync(balls, h->balls, 100 * 16, hipMemcpyDeviceToHost, h->stream));
    SCHK(h, hipStreamSynchronize(h->stream));
    return YH_OK;
}

int yh_scene_time(yh_scene* h, int32_t reps, float* ms_per_frame) {
    if (!h || reps < 1 || !ms_per_frame) return YH_EINVAL;
    if (!h->ran) return h->fail(YH_ESTATE, "no frame has been appended");
    SCHK(h, hipSetDevice(h->dev));
    hipEvent_t a, b;
    SCHK(h, hipEventCreate(&a)); SCHK(h, hipEventCreate(&b));
    SCHK(h, hipEventRecord(a, h->stream));
    // the last append's own inputs and mode (a frame appended through yh_scene_append_classified is replayed from the
    // frame it was given - a device frame must still be valid -, not from the class image buffer it never wrote)
    const uint8_t* cls = h->last_cls; const uint32_t* fr = h->last_frame; const int fm = h->last_frame_mode, md = h->last_mode;
    for (int r = 0; r < reps; ++r) { const int rc = run_scene(h, h->depth, cls, fr, fm, md); if (rc) { hipEventDestroy(a); hipEventDestroy(b); return rc; } }
    SCHK(h, hipEventRecord(b, h->stream));
    SCHK(h, hipEventSynchronize(b));
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    *ms_per_frame = ms / reps;
    return YH_OK;
}

}  // extern "C"
