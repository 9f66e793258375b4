// dev tool: what does an in-kernel grid-wide rendezvous cost on MI355X (352 co-resident workgroups),
// compared with a dependent kernel launch (3.5 us, ubench_latency)?  All spins are bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

constexpr unsigned SPIN_MAX = 400000u;

// V1: one counter per iteration, fetch_add + poll (what clip_barrier in emu_chain.h does)
__global__ __launch_bounds__(256) void k_counter(unsigned *ctr, int iters, float4 *ev, int dirty, int *fail)
{
    const int g = blockIdx.x, tid = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (dirty) ev[(size_t)((it & 7) * gridDim.x + g) * 256 + tid] = make_float4((float)it, 1.f, 2.f, 3.f);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(ctr + it, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(ctr + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                if (++spins > SPIN_MAX) { *fail = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

// V2: flag array, no read-modify-write: workgroup g stores the iteration tag into flags[g]; wave 0 polls
// all flags (one dword per lane per 64 workgroups).  FENCE = 1: release/acquire fences around it.
template <int FENCE>
__global__ __launch_bounds__(256) void k_flags(unsigned *flags, int iters, float4 *ev, int dirty, int *fail, unsigned *sink)
{
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int ng = gridDim.x;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned tag = (unsigned)it + 1u;
        if (dirty) ev[(size_t)((it & 7) * gridDim.x + g) * 256 + tid] = make_float4((float)it, 1.f, 2.f, 3.f);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < 64) {
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (tid == 0) __hip_atomic_store(flags + g, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            for (;;) {
                unsigned mn = 0xFFFFFFFFu;
                for (int k = lane; k < ng; k += 64) {
                    const unsigned v = __hip_atomic_load(flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    mn = v < mn ? v : mn;
                }
                const bool ok = mn >= tag;
                if (__ballot(!ok) == 0ull) break;
                if (++spins > SPIN_MAX) { *fail = 1; break; }
            }
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            acc += spins;
        }
        __syncthreads();
    }
    if (tid == 0 && sink) sink[g] = acc;
}

// V3: like V2 but the flag carries a payload (the max) and 12 "key rows" of u16 are published before it and
// read after it with agent-scope loads: the real per-frame exchange of the emulator.
__global__ __launch_bounds__(256) void k_exchange(unsigned *flags, unsigned *rows /*[2][12][512] u32 = 2 x u16*/, int iters, float4 *ev,
                                                   int dirty, int *fail, unsigned *sink)
{
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ng = gridDim.x;
    __shared__ unsigned s_M;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned tag = (unsigned)it + 1u;
        unsigned *rw = rows + (size_t)(it & 1) * 12 * 512;
        if (dirty) ev[(size_t)((it & 7) * gridDim.x + g) * 256 + tid] = make_float4((float)it, 1.f, 2.f, 3.f);
        // publish 12 key totals of this workgroup (u16 each, stored as relaxed agent-scope shorts)
        if (tid < 12) __hip_atomic_store((unsigned short *)rw + (size_t)tid * 1024 + g, (unsigned short)(tid + it), __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < 64) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (tid == 0) __hip_atomic_store(flags + g, (tag << 8) | (unsigned)(g & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0, mx = 0;
            for (;;) {
                unsigned mn = 0xFFFFFFFFu;
                mx = 0;
                for (int k = lane; k < ng; k += 64) {
                    const unsigned v = __hip_atomic_load(flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    mn = (v >> 8) < mn ? (v >> 8) : mn;
                    mx = (v & 255u) > mx ? (v & 255u) : mx;
                }
                if (__ballot(mn < tag) == 0ull) break;
                if (++spins > SPIN_MAX) { *fail = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int o = 32; o; o >>= 1) { unsigned t = __shfl_xor(mx, o); mx = t > mx ? t : mx; }
            if (tid == 0) s_M = mx;
            acc += spins;
        }
        __syncthreads();
        // read 3 key rows per wave (16 B per lane = 8 workgroups' u16 counts)
        unsigned t = s_M;
        for (int j = 0; j < 3; ++j) {
            const uint4 v = *(const uint4 *)((const unsigned short *)rw + (size_t)(wave + 4 * j) * 1024 + lane * 8);
            t += v.x + v.y + v.z + v.w;
        }
        acc += t;
    }
    if (tid == 0 && sink) sink[g] = acc;
}

int main()
{
    const int NB = 352;
    unsigned *ctr, *flags, *rows, *sink; float4 *ev; int *fail;
    const int ITERS = 2000;
    CK(hipMalloc(&ctr, ITERS * 4)); CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&rows, 2 * 12 * 512 * 4)); CK(hipMalloc(&sink, NB * 4));
    CK(hipMalloc(&ev, (size_t)8 * NB * 256 * 16)); CK(hipMalloc(&fail, 4));
    CK(hipMemset(fail, 0, 4)); CK(hipMemset(rows, 0, 2 * 12 * 512 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_exchange, 256, 0));
    printf("occupancy k_exchange: %d blocks/CU\n", occ);
    for (int dirty = 0; dirty < 2; ++dirty) {
        float ms; int hf = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(ctr, 0, ITERS * 4, s));
            hipEventRecord(e0, s); k_counter<<<NB, 256, 0, s>>>(ctr, ITERS, ev, dirty, fail); hipEventRecord(e1, s); CK(hipStreamSynchronize(s));
        }
        hipEventElapsedTime(&ms, e0, e1); printf("dirty=%d counter+fences     %.3f us/iter\n", dirty, ms * 1e3 / ITERS);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(flags, 0, 4096, s));
            hipEventRecord(e0, s); k_flags<1><<<NB, 256, 0, s>>>(flags, ITERS, ev, dirty, fail, sink); hipEventRecord(e1, s); CK(hipStreamSynchronize(s));
        }
        hipEventElapsedTime(&ms, e0, e1); printf("dirty=%d flags+fences       %.3f us/iter\n", dirty, ms * 1e3 / ITERS);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(flags, 0, 4096, s));
            hipEventRecord(e0, s); k_flags<0><<<NB, 256, 0, s>>>(flags, ITERS, ev, dirty, fail, sink); hipEventRecord(e1, s); CK(hipStreamSynchronize(s));
        }
        hipEventElapsedTime(&ms, e0, e1); printf("dirty=%d flags, no fences   %.3f us/iter\n", dirty, ms * 1e3 / ITERS);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(flags, 0, 4096, s));
            hipEventRecord(e0, s); k_exchange<<<NB, 256, 0, s>>>(flags, rows, ITERS, ev, dirty, fail, sink); hipEventRecord(e1, s); CK(hipStreamSynchronize(s));
        }
        hipEventElapsedTime(&ms, e0, e1); printf("dirty=%d full exchange      %.3f us/iter\n", dirty, ms * 1e3 / ITERS);
        CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
        if (hf) printf("  (a spin bound was hit!)\n");
    }
    return 0;
}
