// placement.hip -- where in device memory the buffers of a batch should live (MI355X).
//
// Measured on MI355X (NPS1 / SPX; tools/placement/placement_map.py, profiles/r02d_placement_map.log): device memory comes in regions
// of tens of GB that alternate between two halves of the memory system, and a kernel whose concurrent streams all live in
// one half sees half the bandwidth -- the 1080p Baseline launch takes 9.6-10.0 ms with planes and RGB in regions of the
// same kind and 8.2-8.4 ms with them in different kinds, whatever the offsets inside a region.  hipMalloc hands out one
// region after the other, so which case a caller gets is chance.  This file (a) tells the groups of two addresses apart with
// a timing probe and (b) places the buffers of a batch inside one large allocation so that each stream has a group of its own
// (tools/placement/placement_predict.py: planes, RGB and records in three different groups = the fastest case, every time).
// (An earlier attempt built buffers from 1-GB chunks of HIP virtual memory management taken in turn from every group:
// commit 2951923; separately created chunks did not classify reliably, see DESIGN.md.)
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
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess) return MVHP_FAILURE;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return MVHP_FAILURE; }
    const size_t n16 = bytes / 16;
    bool ok = true;
    hipLaunchKernelGGL(pair_write_kernel, dim3(2048), dim3(256), 0, 0, (uint4 *)a, (uint4 *)b, n16, 0u);   // warm
    ok = ok && hipGetLastError() == hipSuccess;
    ok = ok && hipEventRecord(e0, 0) == hipSuccess;
    for (int r = 0; r < reps && ok; r++) {
        hipLaunchKernelGGL(pair_write_kernel, dim3(2048), dim3(256), 0, 0, (uint4 *)a, (uint4 *)b, n16, (uint32_t)r);
        ok = hipGetLastError() == hipSuccess;   // a refused launch would leave the two events microseconds apart: a "time" all the same
    }
    ok = ok && hipEventRecord(e1, 0) == hipSuccess;
    ok = ok && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(ms, e0, e1) == hipSuccess;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!ok) { (void)hipGetLastError(); return MVHP_FAILURE; }
    *ms /= (float)reps;
    return MVHP_SUCCESS;
}


// ---------------------------------------------------------------------------------------------------------------
// Placed buffers: one arena, classified in 4-GB blocks, each buffer in a run of blocks of a group of its own.
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct Arena {
    int device;
    void *base;
    size_t bytes;
};
constexpr size_t kBlock = (size_t)4 << 30, kWindow = (size_t)1 << 30;

// The arena a caller gets when it names no size: 200 GB -- more blocks to choose from is a better placement -- unless
// MVHP_PLACED_ARENA_GB says otherwise (memory a process has held is wiped when it is released, in the background, about 3 s for 200 GB,
// and device allocations made meanwhile -- by this or the next process -- wait for it).
static size_t default_arena_cap()
{
    if (const char *e = getenv("MVHP_PLACED_ARENA_GB")) {
        const long gb = atol(e);
        if (gb >= 8 && gb <= 256) return (size_t)gb << 30;
    }
    return (size_t)200 << 30;
}

// a probe that fails makes every later comparison meaningless: remembered in `failed`, checked before anything is placed
struct Probe {
    int device;
    bool failed = false;
    float operator()(void *a, void *b)
    {
        float ms = 0.f;
        if (failed || mvhp_probe_pair(device, a, b, kWindow, 2, &ms) != MVHP_SUCCESS || !(ms > 0.f)) { failed = true; return 0.f; }
        return ms;
    }
};

} // namespace

// `count` (<= 8) device buffers of at least bytes[i] inside ONE allocation of `arena_bytes` (0: as much as is free, less 24 GB,
// at most 200 GB), placed so that -- as far as the groups found allow -- every buffer lies in a different group of the memory
// system, the largest buffers choosing first.  *arena receives a handle for mvhp_placed_free(); groups_of[i] (may be NULL) the
// group index of buffer i (-1: it straddles groups), *groups_found (may be NULL) how many groups the arena showed.
// Costs one large hipMalloc (seconds: the driver clears the memory) + ~0.2 s of probing: for long-lived batch buffers.
MVHP_EXPORT int mvhp_placed_alloc(int device, int count, const size_t *bytes, size_t arena_bytes, void **out, void **arena,
                                  int *groups_of, int *groups_found)
{
    if (!out || !arena || !bytes || count <= 0 || count > 8) return MVHP_FAILURE;
    if (hipSetDevice(device) != hipSuccess) return MVHP_FAILURE;
    size_t fr = 0, tot = 0, need = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return MVHP_FAILURE;
    std::vector<size_t> nblk((size_t)count);
    for (int i = 0; i < count; i++) {
        if (bytes[i] == 0) return MVHP_FAILURE;
        nblk[(size_t)i] = (bytes[i] + kBlock - 1) / kBlock;
        need += nblk[(size_t)i] * kBlock;
    }
    if (arena_bytes == 0) {
        const size_t reserve = (size_t)24 << 30;
        arena_bytes = fr > reserve ? std::min(fr - reserve, default_arena_cap()) : 0;
    }
    arena_bytes = arena_bytes / kBlock * kBlock;
    if (arena_bytes < need) return MVHP_FAILURE;
    void *base = nullptr;
    if (hipMalloc(&base, arena_bytes) != hipSuccess) { (void)hipGetLastError(); return MVHP_FAILURE; }
    const size_t nb = arena_bytes / kBlock;
    auto at = [&](size_t blk, size_t off) { return (void *)((uint8_t *)base + blk * kBlock + off); };
    Probe probe{device};
    auto pair_ms = [&](int, void *a, void *b) { return probe(a, b); };
    // "same group" = a window against its neighbour inside one block (median over up to seven blocks)
    std::vector<float> cal;
    for (size_t b = 0; b < nb && cal.size() < 7; b += std::max<size_t>(1, nb / 7)) cal.push_back(pair_ms(device, at(b, 0), at(b, kWindow)));
    std::sort(cal.begin(), cal.end());
    const float t_same = cal[cal.size() / 2];
    std::vector<int> group(nb, 0);
    std::vector<size_t> reps;   // the first block of every group
    for (size_t b = 0; b < nb; b++) {
        int best = -1;
        float tbest = 0.f;
        for (size_t g = 0; g < reps.size(); g++) {
            const float t = reps[g] == b ? t_same : pair_ms(device, at(reps[g], 0), at(b, 0));
            if (t > tbest) { tbest = t; best = (int)g; }
        }
        if (best >= 0 && tbest >= 0.955f * t_same) group[b] = best;
        else if (reps.size() < 6) { group[b] = (int)reps.size(); reps.push_back(b); }
        else group[b] = best < 0 ? 0 : best;
    }
    // a failed probe (launch refused, event error) must not be read as a time -- the classification and the measured
    // assignment below would be made on meaningless numbers and still be reported as a placement.  Callers fall back to
    // ordinary allocations on MVHP_FAILURE.
    if (probe.failed) { (void)hipFree(base); return MVHP_FAILURE; }
    if (getenv("MVHP_PLACEMENT_TRACE")) {
        fprintf(stderr, "placement: arena %.0f GB, same-group pair %.3f ms, groups per 4 GB:", arena_bytes / 1073741824.0, t_same);
        for (size_t b = 0; b < nb; b++) fprintf(stderr, " %c", 'A' + group[b]);
        fprintf(stderr, "\n");
    }
    // Placement.  The labels above come from single probes against one representative per group and can be off by a block or
    // two (some groups differ by less than others: runs like B B B C B B C B are seen), so the choice is MEASURED: one
    // candidate run per group and buffer size, every pair of candidate windows probed, and the assignment of distinct
    // groups to the buffers with the smallest sum of pair times (= the least shared memory system) wins.  With fewer groups
    // than buffers, or no room, the greedy rule decides: the largest buffers choose first -- a free run of a group nobody has
    // taken yet, else of any one group, else any run.
    const int G = (int)reps.size();
    auto run_start = [&](int g, size_t want, const std::vector<char> &taken_) -> long {
        for (size_t s0 = 0; s0 + want <= nb; s0++) {
            bool okrun = true;
            for (size_t k = 0; k < want && okrun; k++) okrun = !taken_[s0 + k] && group[s0 + k] == g;
            if (okrun) return (long)s0;
        }
        return -1;
    };
    std::vector<char> taken(nb, 0);
    bool placed_all = false;
    if (G >= count && count <= 4) {
        // candidate start of buffer i in group g
        std::vector<std::vector<long>> cand((size_t)count, std::vector<long>((size_t)G, -1));
        for (int i = 0; i < count; i++)
            for (int g = 0; g < G; g++) cand[(size_t)i][(size_t)g] = run_start(g, nblk[(size_t)i], taken);
        // pair times between the groups' windows (the window of a group = the start of its longest candidate)
        std::vector<long> win((size_t)G, -1);
        for (int g = 0; g < G; g++)
            for (int i = 0; i < count; i++)
                if (cand[(size_t)i][(size_t)g] >= 0 && win[(size_t)g] < 0) win[(size_t)g] = cand[(size_t)i][(size_t)g];
        std::vector<std::vector<float>> pt((size_t)G, std::vector<float>((size_t)G, 0.f));
        for (int x = 0; x < G; x++)
            for (int y = x + 1; y < G; y++)
                if (win[(size_t)x] >= 0 && win[(size_t)y] >= 0) {
                    const float t = std::min(pair_ms(device, at((size_t)win[(size_t)x], 0), at((size_t)win[(size_t)y], 0)),
                                             pair_ms(device, at((size_t)win[(size_t)x], kWindow), at((size_t)win[(size_t)y], kWindow)));
                    pt[(size_t)x][(size_t)y] = pt[(size_t)y][(size_t)x] = t;
                }
        std::vector<int> pick((size_t)count, -1), best_pick;
        float best_score = 1e30f;
        std::vector<char> used((size_t)G, 0);
        // depth-first over the buffers: a different group for each
        auto rec = [&](auto &&self, int i, float score) -> void {
            if (score >= best_score) return;
            if (i == count) { best_score = score; best_pick = pick; return; }
            for (int g = 0; g < G; g++) {
                if (used[(size_t)g] || cand[(size_t)i][(size_t)g] < 0) continue;
                float add = 0.f;
                for (int k = 0; k < i; k++) add += pt[(size_t)pick[(size_t)k]][(size_t)g];
                used[(size_t)g] = 1;
                pick[(size_t)i] = g;
                self(self, i + 1, score + add);
                used[(size_t)g] = 0;
            }
        };
        rec(rec, 0, 0.f);
        if (probe.failed) { (void)hipFree(base); return MVHP_FAILURE; }
        if (!best_pick.empty()) {
            for (int i = 0; i < count; i++) {
                const int g = best_pick[(size_t)i];
                const long s0 = cand[(size_t)i][(size_t)g];
                for (size_t k = 0; k < nblk[(size_t)i]; k++) taken[(size_t)s0 + k] = 1;
                out[i] = at((size_t)s0, 0);
                if (groups_of) groups_of[i] = g;
            }
            placed_all = true;
            if (getenv("MVHP_PLACEMENT_TRACE")) {
                fprintf(stderr, "placement: pair times between the groups' windows (ms):");
                for (int x = 0; x < G; x++)
                    for (int y = x + 1; y < G; y++) fprintf(stderr, " %c%c %.3f", 'A' + x, 'A' + y, pt[(size_t)x][(size_t)y]);
                fprintf(stderr, "; chosen:");
                for (int i = 0; i < count; i++) fprintf(stderr, " %c", 'A' + best_pick[(size_t)i]);
                fprintf(stderr, " (sum %.3f)\n", best_score);
            }
        }
    }
    bool ok = true;
    if (!placed_all) {
        std::vector<int> order((size_t)count);
        for (int i = 0; i < count; i++) order[(size_t)i] = i;
        std::sort(order.begin(), order.end(), [&](int x, int y) { return bytes[x] > bytes[y]; });
        std::vector<char> group_used(8, 0);
        for (int oi = 0; oi < count && ok; oi++) {
            const int i = order[(size_t)oi];
            const size_t want = nblk[(size_t)i];
            long best_start = -1;
            int best_rank = 99, best_group = -1;
            for (size_t s0 = 0; s0 + want <= nb; s0++) {
                bool free_run = true, one_group = true;
                for (size_t k = 0; k < want; k++) {
                    if (taken[s0 + k]) { free_run = false; break; }
                    if (group[s0 + k] != group[s0]) one_group = false;
                }
                if (!free_run) continue;
                const int rank = one_group ? (group_used[(size_t)group[s0]] ? 1 : 0) : 2;
                if (rank < best_rank) { best_rank = rank; best_start = (long)s0; best_group = one_group ? group[s0] : -1; }
                if (rank == 0) break;
            }
            if (best_start < 0) { ok = false; break; }
            for (size_t k = 0; k < want; k++) taken[(size_t)best_start + k] = 1;
            if (best_group >= 0) group_used[(size_t)best_group] = 1;
            out[i] = at((size_t)best_start, 0);
            if (groups_of) groups_of[i] = best_group;
        }
    }
    if (!ok) { (void)hipFree(base); return MVHP_FAILURE; }
    if (groups_found) *groups_found = (int)reps.size();
    *arena = new Arena{device, base, arena_bytes};
    return MVHP_SUCCESS;
}

// The same for a PIPELINE's batch buffers (the decode engine: three batches in flight per context): `sets` copies of `count`
// buffers, out[s * count + i] = buffer i of set s.  Buffer i of every set lies in the group chosen for i -- what matters to a
// launch is that ITS records, planes and RGB sit in three different groups, and every set is one launch's buffers; buffers with
// any_group[i] != 0 (a staging area no kernel streams from at speed) go wherever room is left.  At most four buffers may ask
// for a group of their own.  MVHP_FAILURE: not enough memory, or the arena does not show enough groups with room -- use
// ordinary allocations.
MVHP_EXPORT int mvhp_placed_alloc_sets(int device, int sets, int count, const size_t *bytes, const uint8_t *any_group,
                                       size_t arena_bytes, void **out, void **arena, int *groups_of, int *groups_found)
{
    if (!out || !arena || !bytes || sets <= 0 || sets > 8 || count <= 0 || count > 8) return MVHP_FAILURE;
    if (hipSetDevice(device) != hipSuccess) return MVHP_FAILURE;
    size_t fr = 0, tot = 0, need = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return MVHP_FAILURE;
    std::vector<size_t> nblk((size_t)count);
    std::vector<int> own;   // the buffers that want a group of their own
    for (int i = 0; i < count; i++) {
        if (bytes[i] == 0) return MVHP_FAILURE;
        nblk[(size_t)i] = (bytes[i] + kBlock - 1) / kBlock;
        need += nblk[(size_t)i] * kBlock * (size_t)sets;
        if (!any_group || !any_group[i]) own.push_back(i);
    }
    if (own.size() > 4) return MVHP_FAILURE;
    if (arena_bytes == 0) {
        const size_t reserve = (size_t)24 << 30;
        arena_bytes = fr > reserve ? std::min(fr - reserve, default_arena_cap()) : 0;
    }
    arena_bytes = arena_bytes / kBlock * kBlock;
    if (arena_bytes < need) return MVHP_FAILURE;
    void *base = nullptr;
    if (hipMalloc(&base, arena_bytes) != hipSuccess) { (void)hipGetLastError(); return MVHP_FAILURE; }
    const size_t nb = arena_bytes / kBlock;
    auto at = [&](size_t blk, size_t off) { return (void *)((uint8_t *)base + blk * kBlock + off); };
    Probe probe{device};
    // groups of the 4-GB blocks, as in mvhp_placed_alloc: against one representative per group
    std::vector<float> cal;
    for (size_t b = 0; b < nb && cal.size() < 7; b += std::max<size_t>(1, nb / 7)) cal.push_back(probe(at(b, 0), at(b, kWindow)));
    std::sort(cal.begin(), cal.end());
    const float t_same = cal[cal.size() / 2];
    std::vector<int> group(nb, 0);
    std::vector<size_t> reps;
    for (size_t b = 0; b < nb; b++) {
        int best = -1;
        float tbest = 0.f;
        for (size_t g = 0; g < reps.size(); g++) {
            const float t = reps[g] == b ? t_same : probe(at(reps[g], 0), at(b, 0));
            if (t > tbest) { tbest = t; best = (int)g; }
        }
        if (best >= 0 && tbest >= 0.955f * t_same) group[b] = best;
        else if (reps.size() < 6) { group[b] = (int)reps.size(); reps.push_back(b); }
        else group[b] = best < 0 ? 0 : best;
    }
    if (probe.failed) { (void)hipFree(base); return MVHP_FAILURE; }
    const int G = (int)reps.size();
    // `sets` disjoint runs of `want` blocks inside group g (first fit); empty = no room
    auto runs_in = [&](int g, size_t want, const std::vector<char> &taken_) {
        std::vector<size_t> starts;
        std::vector<char> t = taken_;
        for (size_t s0 = 0; s0 + want <= nb && (int)starts.size() < sets; s0++) {
            bool okrun = true;
            for (size_t k = 0; k < want && okrun; k++) okrun = !t[s0 + k] && (g < 0 || group[s0 + k] == g);
            if (!okrun) continue;
            starts.push_back(s0);
            for (size_t k = 0; k < want; k++) t[s0 + k] = 1;
            s0 += want - 1;
        }
        if ((int)starts.size() < sets) starts.clear();
        return starts;
    };
    std::vector<char> taken(nb, 0);
    const int K = (int)own.size();
    bool ok = G >= K;
    std::vector<int> pick((size_t)K, -1), best_pick;
    if (ok && K > 0) {
        // which groups go together is MEASURED (labels alone misplaced a batch once in a dozen runs): pair times between the
        // windows where buffer i would start in group g
        std::vector<std::vector<long>> win((size_t)K, std::vector<long>((size_t)G, -1));
        for (int a = 0; a < K; a++)
            for (int g = 0; g < G; g++) {
                const std::vector<size_t> r = runs_in(g, nblk[(size_t)own[(size_t)a]], taken);
                if (!r.empty()) win[(size_t)a][(size_t)g] = (long)r[0];
            }
        std::vector<long> gw((size_t)G, -1);
        for (int g = 0; g < G; g++)
            for (int a = 0; a < K && gw[(size_t)g] < 0; a++) gw[(size_t)g] = win[(size_t)a][(size_t)g];
        std::vector<std::vector<float>> pt((size_t)G, std::vector<float>((size_t)G, 0.f));
        for (int x = 0; x < G; x++)
            for (int y = x + 1; y < G; y++)
                if (gw[(size_t)x] >= 0 && gw[(size_t)y] >= 0)
                    pt[(size_t)x][(size_t)y] = pt[(size_t)y][(size_t)x] =
                        std::min(probe(at((size_t)gw[(size_t)x], 0), at((size_t)gw[(size_t)y], 0)),
                                 probe(at((size_t)gw[(size_t)x], kWindow), at((size_t)gw[(size_t)y], kWindow)));
        if (probe.failed) { (void)hipFree(base); return MVHP_FAILURE; }
        float best_score = 1e30f;
        std::vector<char> used((size_t)G, 0);
        auto rec = [&](auto &&self, int a, float score) -> void {
            if (score >= best_score) return;
            if (a == K) { best_score = score; best_pick = pick; return; }
            for (int g = 0; g < G; g++) {
                if (used[(size_t)g] || win[(size_t)a][(size_t)g] < 0) continue;
                float add = 0.f;
                for (int k = 0; k < a; k++) add += pt[(size_t)pick[(size_t)k]][(size_t)g];
                used[(size_t)g] = 1;
                pick[(size_t)a] = g;
                self(self, a + 1, score + add);
                used[(size_t)g] = 0;
            }
        };
        rec(rec, 0, 0.f);
        ok = !best_pick.empty();
    }
    if (ok) {
        for (int a = 0; a < K && ok; a++) {
            const int i = own[(size_t)a], g = best_pick[(size_t)a];
            const std::vector<size_t> r = runs_in(g, nblk[(size_t)i], taken);
            if (r.empty()) { ok = false; break; }   // (two buffers' runs competed for the same group: cannot happen with distinct groups)
            for (int st = 0; st < sets; st++) {
                for (size_t k = 0; k < nblk[(size_t)i]; k++) taken[r[(size_t)st] + k] = 1;
                out[(size_t)st * (size_t)count + (size_t)i] = at(r[(size_t)st], 0);
            }
            if (groups_of) groups_of[i] = g;
        }
        for (int i = 0; i < count && ok; i++) {
            if (!any_group || !any_group[i]) continue;
            const std::vector<size_t> r = runs_in(-1, nblk[(size_t)i], taken);
            if (r.empty()) { ok = false; break; }
            for (int st = 0; st < sets; st++) {
                for (size_t k = 0; k < nblk[(size_t)i]; k++) taken[r[(size_t)st] + k] = 1;
                out[(size_t)st * (size_t)count + (size_t)i] = at(r[(size_t)st], 0);
            }
            if (groups_of) groups_of[i] = -1;
        }
    }
    if (getenv("MVHP_PLACEMENT_TRACE")) {
        fprintf(stderr, "placement (sets): arena %.0f GB, %d groups, groups per 4 GB:", arena_bytes / 1073741824.0, G);
        for (size_t b = 0; b < nb; b++) fprintf(stderr, " %c", 'A' + group[b]);
        fprintf(stderr, "; %s\n", ok ? "placed" : "NO ROOM");
    }
    if (!ok) { (void)hipFree(base); return MVHP_FAILURE; }
    if (groups_found) *groups_found = G;
    *arena = new Arena{device, base, arena_bytes};
    return MVHP_SUCCESS;
}

MVHP_EXPORT void mvhp_placed_free(void *arena)
{
    Arena *a = (Arena *)arena;
    if (!a) return;
    (void)hipSetDevice(a->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(a->base);
    delete a;
}
