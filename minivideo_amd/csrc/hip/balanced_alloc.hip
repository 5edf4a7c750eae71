// balanced_alloc.hip -- device buffers spread over both halves of the MI355X memory system.
//
// Measured on MI355X (NPS1 / SPX; tools/placement_map.py, profiles/r02d_placement_map.log): device memory comes in regions
// of tens of GB that alternate between two halves of the memory system, and a kernel whose concurrent streams all live in
// one half sees half the bandwidth -- the 1080p Baseline launch takes 9.6-10.0 ms with planes and RGB in regions of the
// same kind and 8.2-8.4 ms with them in different kinds, whatever the offsets inside a region.  hipMalloc hands out one
// region after the other, so which case a caller gets is chance.  This file (a) tells two addresses' halves apart with a
// timing probe and (b) builds buffers out of 256-MB physical chunks (HIP virtual memory management) taken alternately
// from both halves, so that every stream of a batch is spread over the whole memory system.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "minivideo_hotpath.h"

namespace {

// even workgroups stream 16-byte stores over window a, odd ones over window b
__global__ __launch_bounds__(256) void pair_write_kernel(uint4 *a, uint4 *b, size_t n16, uint32_t tag)
{
    uint4 *w = (blockIdx.x & 1) ? b : a;
    const size_t stride = (size_t)(gridDim.x >> 1) * blockDim.x;
    const uint4 v = make_uint4(tag, tag, tag, tag);
    for (size_t i = (size_t)(blockIdx.x >> 1) * blockDim.x + threadIdx.x; i < n16; i += stride) w[i] = v;
}

} // namespace

// Time `reps` passes of concurrent streaming writes over two device windows of `bytes` each (clobbers both).
MVHP_EXPORT int mvhp_probe_pair(int device, void *a, void *b, size_t bytes, int reps, float *ms)
{
    if (!a || !b || bytes < 4096 || reps <= 0 || !ms) return MVHP_FAILURE;
    if (hipSetDevice(device) != hipSuccess) return MVHP_FAILURE;
    static thread_local hipEvent_t e0 = nullptr, e1 = nullptr;
    static thread_local int ev_device = -1;
    if (ev_device != device) {
        if (e0) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); e0 = e1 = nullptr; }
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return MVHP_FAILURE;
        ev_device = device;
    }
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(pair_write_kernel, dim3(2048), dim3(256), 0, 0, (uint4 *)a, (uint4 *)b, n16, 0u);   // warm
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(pair_write_kernel, dim3(2048), dim3(256), 0, 0, (uint4 *)a, (uint4 *)b, n16, (uint32_t)r);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(ms, e0, e1) != hipSuccess) return MVHP_FAILURE;
    *ms /= (float)reps;
    return MVHP_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------------------
// The allocator.  Physical chunks of kChunk bytes (hipMemCreate) are classified into groups by probing them against
// one reference chunk per group -- a pair in the same group writes >= 0.955 x the calibrated same-group time, a pair in
// different groups about 0.90 x -- and a buffer is a virtual range (hipMemAddressReserve) to which chunks are mapped
// round robin over the groups.  hipMalloc'ing driver hands out one region of the device memory after the other (64 GB
// of one group, then 64 GB of the next on the boxes measured), so finding chunks of every group means creating -- and
// afterwards releasing -- up to ~200 GB of them: fine for what this is for (a few long-lived batch buffers on a GPU
// that decodes video), not a general-purpose allocator.  Every failure on the way degrades to "fewer groups"
// (a plain allocation at worst), never to an error, as long as the memory itself can be had.
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr size_t kChunkWanted = (size_t)1 << 30;   // also the probe window: 512 MB gave +-5 % noise against a 10 % contrast
constexpr int kMaxGroups = 4;

struct Phys {
    hipMemGenericAllocationHandle_t h;
    int group;
};
struct Buffer {
    size_t bytes;                 // reserved = mapped size
    std::vector<Phys> chunks;     // in address order
    int per_group[kMaxGroups];
};
struct DevState {
    bool init = false, usable = false;
    size_t chunk = 0;
    float t_same = 0.f;
    int n_groups = 0;
    void *ref_va[kMaxGroups] = {nullptr, nullptr, nullptr, nullptr};   // one mapped reference chunk per group
    hipMemGenericAllocationHandle_t ref_h[kMaxGroups];
    void *scratch_va = nullptr;                                          // where a new chunk is mapped to be classified
    std::vector<Phys> pool;                                              // classified, unmapped, free
    std::map<void *, Buffer> live;
    size_t created = 0;
};
std::mutex g_mu;
std::map<int, DevState> g_dev;

hipMemAllocationProp chunk_prop(int device)
{
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    p.location.id = device;
    return p;
}

bool map_rw(int device, void *va, size_t size, hipMemGenericAllocationHandle_t h)
{
    if (hipMemMap(va, size, 0, h, 0) != hipSuccess) return false;
    hipMemAccessDesc d = {};
    d.location.type = hipMemLocationTypeDevice;
    d.location.id = device;
    d.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, size, &d, 1) != hipSuccess) { (void)hipMemUnmap(va, size); return false; }
    return true;
}

bool create_chunk(int device, DevState &S, hipMemGenericAllocationHandle_t *h)
{
    const hipMemAllocationProp p = chunk_prop(device);
    if (hipMemCreate(h, S.chunk, &p, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    S.created++;
    return true;
}

float pair_ms(int device, void *a, void *b, size_t bytes)
{
    float ms = 0.f;
    if (mvhp_probe_pair(device, a, b, bytes, 3, &ms) != MVHP_SUCCESS) return -1.f;
    return ms;
}

// first use on a device: granularity, scratch range, calibration on three consecutive chunks (the first becomes group 0's reference)
void init_state(int device, DevState &S)
{
    S.init = true;
    if (hipSetDevice(device) != hipSuccess) return;
    const hipMemAllocationProp p = chunk_prop(device);
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &p, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) { (void)hipGetLastError(); return; }
    S.chunk = (kChunkWanted + gran - 1) / gran * gran;
    constexpr int NC = 6;
    void *va[NC] = {nullptr};
    hipMemGenericAllocationHandle_t h[NC];
    int got = 0;
    for (; got < NC; got++) {
        if (hipMemAddressReserve(&va[got], S.chunk, 0, nullptr, 0) != hipSuccess) break;
        if (!create_chunk(device, S, &h[got])) { (void)hipMemAddressFree(va[got], S.chunk); break; }
        if (!map_rw(device, va[got], S.chunk, h[got])) { (void)hipMemRelease(h[got]); (void)hipMemAddressFree(va[got], S.chunk); break; }
    }
    if (got == NC) {
        // consecutive chunks share a region unless they straddle a boundary: the median of five neighbour pairs is "same group"
        float t[NC - 1];
        for (int i = 0; i + 1 < NC; i++) t[i] = pair_ms(device, va[i], va[i + 1], S.chunk);
        if (getenv("MVHP_BALANCED_TRACE")) fprintf(stderr, "balanced: chunk %zu bytes, calibration pairs %.3f %.3f %.3f %.3f %.3f ms\n", S.chunk, t[0], t[1], t[2], t[3], t[4]);
        std::sort(t, t + NC - 1);
        S.t_same = t[(NC - 1) / 2];
        if (t[0] > 0.f) {
            S.usable = true;
            S.n_groups = 1;
            S.ref_va[0] = va[0];
            S.ref_h[0] = h[0];
            S.scratch_va = va[1];        // keeps its address range; the chunks go to the pool, to be classified when drawn
            for (int i = 1; i < NC; i++) {
                (void)hipMemUnmap(va[i], S.chunk);
                if (i > 1) (void)hipMemAddressFree(va[i], S.chunk);
                S.pool.push_back(Phys{h[i], -1});
            }
            return;
        }
    }
    (void)hipGetLastError();
    for (int i = 0; i < got; i++) { (void)hipMemUnmap(va[i], S.chunk); (void)hipMemRelease(h[i]); (void)hipMemAddressFree(va[i], S.chunk); }
}

// the group of a chunk (mapped at the scratch range for the probes); a chunk unlike every reference founds a new group
int classify(int device, DevState &S, hipMemGenericAllocationHandle_t h)
{
    if (!map_rw(device, S.scratch_va, S.chunk, h)) return 0;
    int best = -1;
    float tbest = 0.f;
    for (int g = 0; g < S.n_groups; g++) {
        const float t = pair_ms(device, S.ref_va[g], S.scratch_va, S.chunk);
        if (t > tbest) { tbest = t; best = g; }
    }
    int group = best;
    if (getenv("MVHP_BALANCED_TRACE")) fprintf(stderr, "balanced: chunk %zu: slowest pair %.3f ms against group %d (same-group time %.3f, %d groups)\n", S.created, tbest, best, S.t_same, S.n_groups);
    if (best < 0 || tbest < 0.95f * S.t_same) {
        if (S.n_groups < kMaxGroups) {
            void *va = nullptr;
            hipMemGenericAllocationHandle_t rh;
            // the new group's reference is a chunk of its own (kept mapped); this chunk is the first member
            group = S.n_groups;
            (void)hipMemUnmap(S.scratch_va, S.chunk);
            if (hipMemAddressReserve(&va, S.chunk, 0, nullptr, 0) == hipSuccess && create_chunk(device, S, &rh)) {
                if (map_rw(device, va, S.chunk, rh)) {
                    // is the fresh chunk in the same group as the one that founded it?  (consecutive chunks: almost always)
                    (void)map_rw(device, S.scratch_va, S.chunk, h);
                    const float t = pair_ms(device, va, S.scratch_va, S.chunk);
                    (void)hipMemUnmap(S.scratch_va, S.chunk);
                    if (t >= 0.95f * S.t_same) {
                        S.ref_va[group] = va;
                        S.ref_h[group] = rh;
                        S.n_groups++;
                        return group;
                    }
                    (void)hipMemUnmap(va, S.chunk);
                }
                (void)hipMemRelease(rh);
                (void)hipMemAddressFree(va, S.chunk);
            }
            (void)hipGetLastError();
            return best < 0 ? 0 : best;   // no reference could be set up: count it with its nearest group
        }
        group = best < 0 ? 0 : best;
    }
    (void)hipMemUnmap(S.scratch_va, S.chunk);
    return group;
}

} // namespace

// `count` device buffers of at least bytes[i], each built from chunks of every part of the memory system in turn, found in ONE
// pass over the device memory (the pass is what costs: ~5 ms per GB walked, up to ~200 GB).  *groups_found (may be NULL) = the
// number of parts used (1 = nothing to balance, or the probe saw no structure).  Free each with mvhp_balanced_free().
// MVHP_FAILURE when the memory cannot be had or the device lacks virtual memory management (callers fall back to hipMalloc).
MVHP_EXPORT int mvhp_balanced_alloc_many(int device, int count, const size_t *bytes, void **out, int *groups_found)
{
    if (!out || !bytes || count <= 0 || count > 16) return MVHP_FAILURE;
    std::lock_guard<std::mutex> l(g_mu);
    DevState &S = g_dev[device];
    if (hipSetDevice(device) != hipSuccess) return MVHP_FAILURE;
    if (!S.init) init_state(device, S);
    if (!S.usable) return MVHP_FAILURE;
    std::vector<size_t> nch((size_t)count);
    size_t n = 0;
    for (int i = 0; i < count; i++) {
        if (bytes[i] == 0) return MVHP_FAILURE;
        nch[(size_t)i] = (bytes[i] + S.chunk - 1) / S.chunk;
        n += nch[(size_t)i];
    }
    const size_t want_each = (n + 2) / 3;                 // three groups were seen on MI355X; fewer found = fewer used
    const size_t create_limit = n + ((size_t)216 << 30) / S.chunk;
    std::vector<std::vector<Phys>> by(kMaxGroups);
    std::vector<Phys> raw;
    for (const Phys &p : S.pool) (p.group >= 0 ? by[(size_t)p.group] : raw).push_back(p);
    S.pool.clear();
    size_t made = 0;
    auto enough = [&]() {
        size_t full = 0, total = 0;
        for (auto &v : by) { full += v.size() >= want_each; total += v.size(); }
        return total >= n && (full >= 3 || made >= create_limit);
    };
    // The driver hands out one region after the other, so consecutive chunks share a group except at a region boundary: every
    // eighth chunk is probed; when its group differs from the previous probe's, the chunks in between are probed one by one.
    std::vector<Phys> pending;
    int last_group = -1;
    bool out_of_memory = false;
    auto settle = [&](int g_now) {
        if (g_now == last_group || last_group < 0) {
            for (Phys &q : pending) { q.group = g_now; by[(size_t)g_now].push_back(q); }
        } else {
            for (Phys &q : pending) { q.group = classify(device, S, q.h); by[(size_t)q.group].push_back(q); }
        }
        pending.clear();
        last_group = g_now;
    };
    while (!enough() || !pending.empty()) {
        if (enough() && !pending.empty()) { settle(classify(device, S, pending.back().h)); continue; }
        Phys p;
        p.group = -1;
        if (!raw.empty()) { p = raw.back(); raw.pop_back(); }
        else {
            if (out_of_memory || !create_chunk(device, S, &p.h)) { out_of_memory = true; if (pending.empty()) break; settle(classify(device, S, pending.back().h)); continue; }
            made++;
        }
        // (a chunk only takes physical memory when it is first mapped: map it now so that the walk advances)
        if (map_rw(device, S.scratch_va, S.chunk, p.h)) (void)hipMemUnmap(S.scratch_va, S.chunk);
        pending.push_back(p);
        if (pending.size() >= 8) {
            Phys last = pending.back();
            pending.pop_back();
            const int g = classify(device, S, last.h);
            settle(g);
            last.group = g;
            by[(size_t)g].push_back(last);
        }
        if (out_of_memory) break;
    }
    size_t total = 0;
    for (auto &v : by) total += v.size();
    bool ok = total >= n;
    std::vector<void *> vas((size_t)count, nullptr);
    std::vector<Buffer> bufs((size_t)count);
    int turn = 0;
    for (int bi = 0; bi < count && ok; bi++) {
        Buffer &B = bufs[(size_t)bi];
        B.bytes = nch[(size_t)bi] * S.chunk;
        memset(B.per_group, 0, sizeof(B.per_group));
        void *va = nullptr;
        if (hipMemAddressReserve(&va, B.bytes, 0, nullptr, 0) != hipSuccess) { ok = false; break; }
        vas[(size_t)bi] = va;
        for (size_t i = 0; i < nch[(size_t)bi] && ok; i++) {
            int g = -1;
            for (int k = 1; k <= kMaxGroups; k++) {   // the next group after the last one used that still has a chunk
                const int cand = (turn + k) % kMaxGroups;
                if (!by[(size_t)cand].empty()) { g = cand; break; }
            }
            if (g < 0) { ok = false; break; }
            turn = g;
            Phys p = by[(size_t)g].back();
            by[(size_t)g].pop_back();
            if (!map_rw(device, (uint8_t *)va + i * S.chunk, S.chunk, p.h)) { by[(size_t)g].push_back(p); ok = false; break; }
            B.chunks.push_back(p);
            B.per_group[g]++;
        }
    }
    if (!ok) {
        for (int bi = 0; bi < count; bi++) {
            if (!vas[(size_t)bi]) continue;
            Buffer &B = bufs[(size_t)bi];
            for (size_t i = 0; i < B.chunks.size(); i++) {
                (void)hipMemUnmap((uint8_t *)vas[(size_t)bi] + i * S.chunk, S.chunk);
                (void)hipMemRelease(B.chunks[i].h);
            }
            (void)hipMemAddressFree(vas[(size_t)bi], B.bytes);
        }
    }
    // what was not used goes back to the driver
    for (auto &v : by)
        for (const Phys &p : v) (void)hipMemRelease(p.h);
    for (const Phys &p : raw) (void)hipMemRelease(p.h);
    for (const Phys &p : pending) (void)hipMemRelease(p.h);
    (void)hipGetLastError();
    if (!ok) return MVHP_FAILURE;
    int used_mask = 0;
    for (int bi = 0; bi < count; bi++) {
        for (int g = 0; g < kMaxGroups; g++) used_mask |= (bufs[(size_t)bi].per_group[g] > 0) << g;
        S.live[vas[(size_t)bi]] = bufs[(size_t)bi];
        out[bi] = vas[(size_t)bi];
    }
    if (groups_found) *groups_found = __builtin_popcount((unsigned)used_mask);
    return MVHP_SUCCESS;
}

MVHP_EXPORT int mvhp_balanced_alloc(int device, size_t bytes, void **out, int *groups_found)
{
    return mvhp_balanced_alloc_many(device, 1, &bytes, out, groups_found);
}

MVHP_EXPORT int mvhp_balanced_free(int device, void *ptr)
{
    std::lock_guard<std::mutex> l(g_mu);
    auto di = g_dev.find(device);
    if (di == g_dev.end()) return MVHP_FAILURE;
    DevState &S = di->second;
    auto it = S.live.find(ptr);
    if (it == S.live.end()) return MVHP_FAILURE;
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    const Buffer &B = it->second;
    for (size_t i = 0; i < B.chunks.size(); i++) {
        (void)hipMemUnmap((uint8_t *)ptr + i * S.chunk, S.chunk);
        (void)hipMemRelease(B.chunks[i].h);
    }
    (void)hipMemAddressFree(ptr, B.bytes);
    S.live.erase(it);
    return MVHP_SUCCESS;
}

// chunks per group of a live buffer (diagnostics): out[0..3]
MVHP_EXPORT int mvhp_balanced_info(int device, void *ptr, int *per_group4, size_t *chunk_bytes)
{
    std::lock_guard<std::mutex> l(g_mu);
    auto di = g_dev.find(device);
    if (di == g_dev.end()) return MVHP_FAILURE;
    auto it = di->second.live.find(ptr);
    if (it == di->second.live.end()) return MVHP_FAILURE;
    if (per_group4) memcpy(per_group4, it->second.per_group, sizeof(int) * kMaxGroups);
    if (chunk_bytes) *chunk_bytes = di->second.chunk;
    return MVHP_SUCCESS;
}
