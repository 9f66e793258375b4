// render.hip -- EventRenderer.render_events_to_frames on a device-resident event packet.
//
// What the reference computes (v2ecore/renderer.py:161-400): a packet of n events is cut into exposure windows -- by time
// (DURATION), by event count (COUNT), by the first area cell that collects `area_count` events (AREA_COUNT), or not at all
// (SOURCE) -- every complete window becomes a frame clip(hist_ON - hist_OFF, +-full_scale) normalised to 0..1, the window the
// packet ends in stays behind as `currentFrame`; the packet's last event is never accumulated (renderer.py:303-306).
//
// Here the windows are not walked: one launch (k_render_hist) puts every event into the frame(s) its window rule assigns
// it to -- a binary search over the packet's window boundaries (DURATION: time stamps, an event exactly on a boundary
// belongs to both neighbours, as searchsorted left / right give it; AREA_COUNT: the segment ends found by k_area_segments),
// an integer division (COUNT) -- and adds +-1 to that frame's bin; one launch (k_render_finish) clips and normalises all
// frames of the packet.  k_area_segments is the one sequential rule (a cell counter that resets at every trigger): a
// single wave walks the packet 64 events at a time, ranks the events of a chunk within their cells by ballot, and restarts
// at the trigger event, which the reference counts again in the next window (renderer.py:253-266).
#include "common.h"

namespace {

constexpr int MODE_DURATION = 1, MODE_COUNT = 2, MODE_AREA_COUNT = 3, MODE_SOURCE = 4; // renderer.py:19-23

struct RenderArgs {
    const float4 *ev;
    long long n_used;          // events that take part: n - 1
    int mode;
    const double *bounds;      // DURATION: T_0 .. T_m (frame k = [T_k, T_k+1], the open frame m = [T_m, ...))
    int n_bounds;
    long long count_per_frame; // COUNT
    const int *seg_end;        // AREA_COUNT: end index (exclusive) of complete frame k
    int n_seg;
    int n_frames;              // complete frames + the open one
    int *diff;                 // [n_frames][bins_y * bins_x]
    int bins_y, bins_x;
    double y_lo, x_lo, delta_y, delta_x;
};

// number of entries of the sorted array a[0..n) that are <= v
template <typename T> __device__ __forceinline__ int count_le(const T *__restrict__ a, int n, T v)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_render_hist(RenderArgs r)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= r.n_used) return;
    const float4 e = r.ev[i];
    // hist2d_numba_seq (v2e_utils.py:474-486): tracks[0] = y, tracks[1] = x
    const double fi = ((double)e.z - r.y_lo) * r.delta_y;
    const double fj = ((double)e.y - r.x_lo) * r.delta_x;
    if (!(fi >= 0.0 && fi < (double)r.bins_y && fj >= 0.0 && fj < (double)r.bins_x)) return;
    const int bin = (int)fi * r.bins_x + (int)fj;
    const int v = e.w == 1.0f ? 1 : -1; // pol_on = (p == 1), off = not on (renderer.py:379-380)
    long long k0 = 0, k1 = -1;
    if (r.mode == MODE_DURATION) {
        const double t = (double)e.x;
        const int c = count_le<double>(r.bounds, r.n_bounds, t);
        k0 = c - 1; // the frame whose start is the last boundary <= t; before T_0: none
        if (k0 >= 1 && r.bounds[k0] == t) k1 = k0 - 1; // searchsorted(next_start, side="right") keeps it in the frame before, too
    } else if (r.mode == MODE_COUNT) {
        k0 = i / r.count_per_frame;
    } else if (r.mode == MODE_AREA_COUNT) {
        k0 = count_le<int>(r.seg_end, r.n_seg, (int)i);
    }
    const size_t npx = (size_t)r.bins_y * r.bins_x;
    if (k0 >= 0 && k0 < r.n_frames) atomicAdd(&r.diff[(size_t)k0 * npx + bin], v);
    if (k1 >= 0 && k1 < r.n_frames) atomicAdd(&r.diff[(size_t)k1 * npx + bin], v);
}

// frames 0 .. n_complete-1: (clip(diff) + fs) / (2 fs) (renderer.py:245-247, 396-400); frame n_complete (if present): clip(diff)
__global__ __launch_bounds__(256) void k_render_finish(const int *__restrict__ diff, long long total, long long npx, int n_complete,
                                                      double *__restrict__ frames, double *__restrict__ cur, double fs)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    double v = (double)diff[i];
    v = v < -fs ? -fs : (v > fs ? fs : v);
    const long long k = i / npx;
    if (k < n_complete) frames[i] = (v + fs) / (fs * 2);
    else if (cur) cur[i - k * npx] = v;
}

__global__ void k_zero_i32(int *p, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0;
}

// AREA_COUNT windows (renderer.py:253-266, 292-298): walk the events in order, count per area cell, a window ends at the
// first event whose cell count reaches `area_count` -- that event opens the next window and is counted again there.
// One wave; counters in LDS (n_cells ints).  out[0] = complete windows found, out[1] = 1 if the walk ended on a trigger
// (the reference then leaves zeroed counters behind), seg_end[k] = end (exclusive) of window k.
__global__ __launch_bounds__(64) void k_area_segments(const float4 *__restrict__ ev, long long n, int *__restrict__ area_counts, int nw,
                                                     int nh, float area_dim, int area_count, int *__restrict__ seg_end, long long cap,
                                                     int *__restrict__ out)
{
    extern __shared__ int s_cnt[];
    const int lane = threadIdx.x;
    const int n_cells = nw * nh;
    for (int c = lane; c < n_cells; c += 64) s_cnt[c] = area_counts[c];
    __syncthreads();
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull), gt = ~le;
    long long pos = 0;
    int nseg = 0, ended_on_trigger = 0;
    while (pos < n) {
        const long long i = pos + lane;
        const bool act = i < n;
        int cell = -1;
        if (act) {
            const float4 e = ev[i];
            int cx = (int)floorf(e.y / area_dim), cy = (int)floorf(e.z / area_dim); // int(events[i, 1] // area_dimension)
            cx = min(max(cx, 0), nw - 1);
            cy = min(max(cy, 0), nh - 1);
            cell = cx * nh + cy; // area_counts[x, y], shape (nw, nh)
        }
        // rank of every event among the chunk's events of its cell, in index order (1-based), and whether it is the last one
        int rank = 0;
        bool last = false, todo = act;
        unsigned long long tb;
        while ((tb = __ballot(todo)) != 0ull) {
            const int leader = (int)__builtin_ctzll(tb);
            const int v = __builtin_amdgcn_readlane(cell, leader);
            const unsigned long long m = __ballot(act && cell == v);
            if (act && cell == v) {
                rank = (int)__popcll(m & le);
                last = (m & gt) == 0ull;
                todo = false;
            }
        }
        const int cnt = act ? s_cnt[cell] + rank : 0;
        const unsigned long long trig = __ballot(act && cnt >= area_count);
        if (trig == 0ull) {
            if (act && last) s_cnt[cell] = cnt;
            __syncthreads();
            pos += 64;
            continue;
        }
        const long long end = pos + (int)__builtin_ctzll(trig);
        for (int c = lane; c < n_cells; c += 64) s_cnt[c] = 0; // area_counts = np.zeros_like(area_counts)
        __syncthreads();
        if (end >= n - 1) { ended_on_trigger = 1; break; } // doneWithTheseEvents: the open window takes the rest
        if (nseg < cap) { if (lane == 0) seg_end[nseg] = (int)end; }
        ++nseg;
        if (end == pos && __builtin_ctzll(trig) == 0 && area_count <= 1) break; // a window that cannot advance (the reference would not return)
        pos = end; // the trigger event is the first event of the next window
    }
    for (int c = lane; c < n_cells; c += 64) area_counts[c] = s_cnt[c];
    if (lane == 0) { out[0] = nseg; out[1] = ended_on_trigger; }
}

} // namespace

extern "C" {

int v2e_render_area_segments(const float *events, int64_t n, int32_t *area_counts, int nw, int nh, double area_dimension, int area_count,
                             int32_t *seg_end, int64_t cap, int32_t *out2, void *stream)
{
    V2E_REQUIRE(events && area_counts && seg_end && out2 && n > 0 && nw > 0 && nh > 0 && area_count >= 2 && area_dimension > 0, "bad args");
    V2E_REQUIRE((size_t)nw * nh * sizeof(int) <= 64 * 1024, "too many area cells for the LDS counter table (64 KB)");
    k_area_segments<<<1, 64, (size_t)nw * nh * sizeof(int), (hipStream_t)stream>>>((const float4 *)events, n, area_counts, nw, nh,
                                                                                (float)area_dimension, area_count, seg_end, cap, out2);
    V2E_HIP(hipGetLastError());
    return 0;
}

int v2e_render_packet(const float *events, int64_t n_used, int mode, const double *bounds, int n_bounds, int64_t count_per_frame,
                      const int32_t *seg_end, int n_seg, int n_complete, int has_open, int32_t *diff, double *frames_out,
                      double *cur_out, int bins_y, int bins_x, double y_lo, double y_hi, double x_lo, double x_hi, double full_scale,
                      void *stream)
{
    V2E_REQUIRE(diff && bins_y > 0 && bins_x > 0 && n_complete >= 0 && (events || n_used == 0) && full_scale > 0, "bad args");
    V2E_REQUIRE(mode >= MODE_DURATION && mode <= MODE_SOURCE, "bad exposure mode");
    V2E_REQUIRE(mode != MODE_DURATION || (bounds && n_bounds >= 1), "DURATION needs the frame boundaries");
    V2E_REQUIRE(mode != MODE_COUNT || count_per_frame >= 1, "COUNT needs a positive event count");
    V2E_REQUIRE(mode != MODE_AREA_COUNT || seg_end || n_seg == 0, "AREA_COUNT needs the segment ends");
    V2E_REQUIRE(n_complete == 0 || frames_out, "null frames_out");
    const int n_frames = n_complete + (has_open ? 1 : 0);
    if (n_frames == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const long long npx = (long long)bins_y * bins_x, total = npx * n_frames;
    k_zero_i32<<<v2e_cdiv(total, 256), 256, 0, s>>>(diff, total);
    if (n_used > 0) {
        RenderArgs r;
        r.ev = (const float4 *)events; r.n_used = n_used; r.mode = mode; r.bounds = bounds; r.n_bounds = n_bounds;
        r.count_per_frame = count_per_frame; r.seg_end = seg_end; r.n_seg = n_seg; r.n_frames = n_frames; r.diff = diff;
        r.bins_y = bins_y; r.bins_x = bins_x; r.y_lo = y_lo; r.x_lo = x_lo;
        r.delta_y = 1 / ((y_hi - y_lo) / bins_y); r.delta_x = 1 / ((x_hi - x_lo) / bins_x); // v2e_utils.py:478
        k_render_hist<<<v2e_cdiv(n_used, 256), 256, 0, s>>>(r);
    }
    k_render_finish<<<v2e_cdiv(total, 256), 256, 0, s>>>(diff, total, npx, n_complete, frames_out, has_open ? cur_out : nullptr, full_scale);
    V2E_HIP(hipGetLastError());
    return 0;
}

} // extern "C"
