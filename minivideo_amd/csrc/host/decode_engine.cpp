// decode_engine.cpp -- see decode_engine.h.
//
// Threads of one decode call (all joined before it returns):
//   feeder              hands out pictures in `order`, never more than `wanted` minus what is already accepted or in
//                       flight (the reference stops after picture_number IDRs, h264.c:173-179), grouped into chunks
//                       (one H2D transfer) and batches (one kernel launch, one set of stream parameters);
//   T entropy workers   mvhp_stream::decode_compact() straight into a page-locked chunk slot (the compact transfer
//                       format: only non-zero levels cross PCIe; the GPU expands it into packed records);
//   per context:        uploader (claims whole batches from the shared queue: pictures are independent, so this is
//                       the frame-level work queue of SURVEY 8e -- no collective), launcher, downloader;
//   caller thread       calls the sink once per picture, in order.
// One mutex + one condition variable guard all queues; the heavy work (entropy decode, copies, kernels, sink) runs
// outside the lock.
#include "decode_engine.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "h264_frontend.h"
#include "stream_internal.h"

namespace mvengine {

namespace {

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    return atoi(e);
}

} // namespace

// Host cores this process may actually use: the hardware thread count, cut down to the CPU-time quota of the
// container (cgroup v2 cpu.max / v1 cfs quota) -- a box of 256 hardware threads often grants 16 CPUs per GPU, and
// 256 entropy threads sharing 16 CPUs only thrash.
int effective_cores()
{
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    long quota = -1, period = -1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = "";
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atol(q);
        fclose(f);
    } else {
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%ld", &period) != 1) period = -1; fclose(g); }
    }
    if (quota > 0 && period > 0) n = std::min<long>(n, std::max<long>(1, (quota + period - 1) / period));
    return n;
}

namespace {

bool same_params(const mvhp_stream_params_t &a, const mvhp_stream_params_t &b)
{
    return a.width_mbs == b.width_mbs && a.height_mbs == b.height_mbs &&
           a.chroma_qp_index_offset == b.chroma_qp_index_offset &&
           a.second_chroma_qp_index_offset == b.second_chroma_qp_index_offset && a.flags == b.flags &&
           (!(a.flags & MVHP_PARAM_SCALING) ||
            (memcmp(a.scaling4, b.scaling4, sizeof(a.scaling4)) == 0 && memcmp(a.scaling8, b.scaling8, sizeof(a.scaling8)) == 0));
}

struct Pinned {
    uint8_t *p = nullptr;
    size_t cap = 0;
};

struct InChunk {
    Pinned buf;              // n slots of pic_bytes: one compact picture each (include/minivideo_hotpath.h)
    int batch = -1;
    int first_slot = 0;      // picture offset inside the batch
    int n = 0;               // pictures in this chunk
    size_t pic_bytes = 0;    // slot size = the most a compact picture of these parameters can take
    std::vector<size_t> used;// bytes of each slot actually written
    int remaining = 0;       // pictures not yet entropy-decoded (guarded by the engine mutex)
};

struct OutChunk {
    Pinned yuv, rgb;
    int refs = 0;            // pictures handed to the sink queue and not yet consumed
};

struct PicResult {
    int idr = -1;
    int rc = MVHP_FAILURE;
    std::string err;
    mvhp_stream_params_t params{};
    OutChunk *oc = nullptr;
    const uint8_t *yuv = nullptr, *rgb = nullptr;
    bool ready = false;      // final: the sink may take it
    bool parsed_ok = false;
    bool kept = false;       // the sink answered 2: its output chunk stays referenced until release_picture()
};

struct DevBuf {
    void *compact = nullptr; // uploaded compact pictures, slot stride as in the chunks
    void *packed = nullptr;  // the records they expand to
    uint8_t *yuv = nullptr, *rgb = nullptr;
    size_t compact_cap = 0, packed_cap = 0, yuv_cap = 0, rgb_cap = 0;
    bool busy = false;
    bool arena_piece = false;   // the four buffers are pieces of the context's placed arena: never freed one by one
};

// slot size of a compact picture: the format's upper bound, 16-byte aligned
size_t compact_slot_bytes(const mvhp_stream_params_t &p)
{
    const size_t mbs = (size_t)p.width_mbs * p.height_mbs;
    return (mbs * MVHP_COMPACT_MB_BYTES_MAX + MVHP_COMPACT_SLACK_BYTES + 15) & ~(size_t)15;
}

// a valid compact picture without a single level (every offset points at one empty Intra4x4 record): what a slot holds
// when its picture failed to parse, so that the batch can still be expanded
size_t write_empty_compact(uint8_t *buf, size_t mbs)
{
    memset(buf, 0, mbs * 4 + MVHP_MB_HEADER_BYTES);
    return mbs * 4 + MVHP_MB_HEADER_BYTES;
}

struct Batch {
    int id = 0;
    mvhp_stream_params_t params{};
    std::vector<int> seqs;           // slot -> position in `order`
    int capacity = 0;                // planned pictures (device buffers are sized for it)
    int total = -1;                  // pictures, known once the closing chunk has been issued
    int uploaded = 0;
    int ctx = -1;                    // claimed by this context
    int exclude_ctx = -1;            // a re-queued batch does not go back to the context it failed on
    bool retry = false;
    bool dead = false;
    std::string dead_why;
    DevBuf *buf = nullptr;
    std::deque<InChunk *> ready;     // entropy-decoded chunks waiting for upload
    double t_open = 0, t_claim = 0, t_closed = 0, t_uploaded = 0, t_kernel = 0;   // MINIVIDEO_ENGINE_TRACE (seconds into the call)
};

struct Item {
    InChunk *chunk;
    int slot;   // inside the chunk
    int seq;
};

struct RetryGroup {
    std::vector<int> seqs;
    int exclude_ctx = -1;
    size_t pos = 0;
};

constexpr int kMaxDevBufs = 8;

struct Ctx {
    DevCtx *dev = nullptr;
    int device = 0;
    DevBuf bufs[kMaxDevBufs];   // n_bufs of them in use.  Three batches per context is the minimum: one filling / uploading, one in
                      // the kernel, one downloading -- with two, the batch after next could not be claimed until a download finished
                      // and the entropy threads ran out of chunk slots (170 pictures took 90 ms instead of 44 to fill in the taper of
                      // a 2048-picture job).  More (MINIVIDEO_DEVBUFS = 4...8) were measured in round 3: no gain (tools/e2e_devbufs_sweep.sh)
    int n_bufs = 3;
    int open_batch = -1;
    std::deque<int> to_launch, to_download;
    bool fail_next = false;
    bool launched_once = false;   // (over the engine's life: the code objects stay loaded)
    void *arena = nullptr;        // MINIVIDEO_PLACED=1: the three batch buffers live in one placed arena ...
    int arena_pictures = 0;       // ... sized for this many pictures per batch ...
    mvhp_stream_params_t arena_params{};   // ... of this shape
    bool arena_tried = false;
    size_t mem_budget = 0;   // bytes one batch may occupy on the device
};

} // namespace

class Engine {
public:
    Engine(const DeviceApi &api) : api_(api) {}
    ~Engine();
    bool init(const mvhp_engine_opts_t *opts, std::string &err);
    int decode(const mvhp_stream &s, const int *order, int n_order, int wanted, int out_mask, mvhp_picture_sink_t sink,
               void *user, mvhp_decode_stats_t *stats, std::string &err);

private:
    // ---- threads ----
    void feeder();
    void worker(int t);
    void uploader(int k);
    void launcher(int k);
    void downloader(int k);
    // ---- helpers (mutex held unless noted) ----
    int allowance() const { return wanted_ - ok_ - (issued_ - consumed_); }
    bool pick_chunk(int k, InChunk **c, Batch **b);
    void fail_batch(Batch *b, const std::string &why);
    void close_batch(Batch *b);
    void release_batch(Batch *b);
    void put_out(OutChunk *oc);
public:
    void release_picture(int seq);   // any thread
private:
    bool grow(Pinned &p, size_t need);   // no lock needed
    int batch_capacity(const mvhp_stream_params_t &p, int remaining) const;
    int planned_batch(int cap, int remaining, int batch_id) const;
    int chunk_pictures(const mvhp_stream_params_t &p) const;
    bool ensure_devbuf(Ctx &c, DevBuf &b, const Batch &bt, std::string &err);   // no lock needed

    const DeviceApi &api_;
    mvhp_engine_opts_t opts_{};
    std::vector<Ctx> ctx_;
    int host_threads_ = 1;
    std::vector<std::unique_ptr<InChunk>> all_in_;
    std::vector<std::unique_ptr<OutChunk>> all_out_;
    size_t in_limit_ = 4, out_limit_ = 4;

    // ---- state of the running decode call ----
    std::mutex mu_;
    std::condition_variable cv_;
    const mvhp_stream *s_ = nullptr;
    const int *order_ = nullptr;
    int n_order_ = 0, wanted_ = 0;
    bool want_rgb_ = false, want_yuv_ = true;   // which outputs are downloaded (the planes are always reconstructed)
    bool stop_ = false;
    bool sink_waiting_ = false;
    int pos_ = 0;                 // next position of `order` the feeder has not issued yet
    int issued_ = 0, consumed_ = 0, ok_ = 0, failed_ = 0;
    int kept_ = 0;                // pictures a sink kept (verdict 2) and has not released yet
    int in_sink_ = -1;            // the picture whose sink callback is running
    bool released_early_ = false; // ... and was given back by another thread meanwhile
    int next_batch_id_ = 0;
    std::vector<PicResult> results_;
    std::map<int, std::unique_ptr<Batch>> batches_;
    std::deque<Item> work_q_;
    std::deque<RetryGroup> retry_q_;
    std::deque<InChunk *> free_in_;
    std::deque<OutChunk *> free_out_;
    mvhp_decode_stats_t st_{};
    std::vector<double> worker_busy_, worker_wait_;   // per entropy thread: inside decode_compact / waiting for work
    double feeder_wait_chunk_ = 0, t_last_entropy_ = 0, t_last_download_ = 0, t_start_ = 0;   // MINIVIDEO_ENGINE_TRACE
    std::atomic<uint64_t> stream_bytes_{0};
    // allocations of the running call (cold-start accounting; their own lock: grow() and ensure_devbuf() run unlocked)
    std::mutex alloc_mu_;
    double alloc_host_s_ = 0, alloc_dev_s_ = 0, first_launch_s_ = 0, first_picture_s_ = 0;
    uint64_t alloc_host_bytes_ = 0, alloc_dev_bytes_ = 0;
    bool placed_ = false;    // MINIVIDEO_PLACED=1 / opts.reserved[0] & 1
    int job_cap_ = 1;        // the largest batch the running call can form (per context)
    mvhp_stream_params_t job_params_{};   // ... of pictures of this size (the first picture's)
};

Engine::~Engine()
{
    for (auto &c : all_in_) api_.host_free(c->buf.p);
    for (auto &c : all_out_) { api_.host_free(c->yuv.p); api_.host_free(c->rgb.p); }
    for (Ctx &c : ctx_) {
        if (c.arena) {   // pieces of the arena go with it; a buffer that left the arena is freed below like any other
            for (DevBuf &b : c.bufs) if (b.arena_piece) b = DevBuf();
            api_.placed_free(c.dev, c.arena);
        }
        for (DevBuf &b : c.bufs) {
            if (b.compact) api_.dev_free(c.dev, b.compact);
            if (b.packed) api_.dev_free(c.dev, b.packed);
            if (b.yuv) api_.dev_free(c.dev, b.yuv);
            if (b.rgb) api_.dev_free(c.dev, b.rgb);
        }
        if (c.dev) api_.ctx_destroy(c.dev);
    }
}

bool Engine::init(const mvhp_engine_opts_t *opts, std::string &err)
{
    if (opts) opts_ = *opts;
    else opts_.fail_context = -1;
    const int n_dev = api_.device_count();
    if (n_dev <= 0) { err = "no HIP device available: this build has no CPU reconstruction path"; return false; }
    int n_ctx = opts_.contexts;
    if (n_ctx <= 0) {
        n_ctx = n_dev;
        const int cap = env_int("MINIVIDEO_GPUS", 0);
        if (cap > 0 && cap < n_ctx) n_ctx = cap;
        const int fake = env_int("MINIVIDEO_FAKE_GPUS", 0);   // several contexts on the devices there are (tests)
        if (fake > 0) n_ctx = fake;
    }
    if (n_ctx > 64) n_ctx = 64;
    host_threads_ = opts_.host_threads > 0 ? opts_.host_threads : env_int("MINIVIDEO_HOST_THREADS", 0);
    if (host_threads_ <= 0) host_threads_ = effective_cores();
    if (host_threads_ < 1) host_threads_ = 1;
    if (host_threads_ > 256) host_threads_ = 256;
    if (opts_.batch_pictures <= 0) opts_.batch_pictures = env_int("MINIVIDEO_BATCH", 0);
    if (opts_.fail_context < 0) opts_.fail_context = env_int("MINIVIDEO_TEST_FAIL_CONTEXT", -1);
    placed_ = (opts_.reserved[0] & 1) != 0 || env_int("MINIVIDEO_PLACED", 0) != 0;
    ctx_.resize((size_t)n_ctx);
    for (int k = 0; k < n_ctx; k++) {
        ctx_[k].device = (std::max(0, opts_.first_device) + k) % n_dev;
        ctx_[k].dev = api_.ctx_create(ctx_[k].device, err);
        if (!ctx_[k].dev) return false;
    }
    // device memory one batch may take: half of what is free now, split over the batch buffers (filling / in the kernel / downloading / waiting for the download) of every context
    // that shares the device
    for (int k = 0; k < n_ctx; k++) {
        int sharers = 0;
        for (int j = 0; j < n_ctx; j++) sharers += ctx_[j].device == ctx_[k].device;
        const size_t free_b = api_.dev_free_bytes(ctx_[k].dev);
        ctx_[k].n_bufs = std::max(3, std::min(kMaxDevBufs, env_int("MINIVIDEO_DEVBUFS", 3)));
        ctx_[k].mem_budget = free_b / 2 / (size_t)ctx_[k].n_bufs / (size_t)std::max(1, sharers);   // half of what is free, over the batch buffers
    }
    return true;
}

bool Engine::grow(Pinned &p, size_t need)
{
    if (need <= p.cap) return true;
    api_.host_free(p.p);
    const double t0 = now_s();
    p.p = (uint8_t *)api_.host_alloc(need);
    {
        std::lock_guard<std::mutex> l(alloc_mu_);
        alloc_host_s_ += now_s() - t0;
        alloc_host_bytes_ += need;
    }
    p.cap = p.p ? need : 0;
    return p.p != nullptr;
}

// pictures per transfer: ~64 MiB of records, so that a transfer runs at the link rate and a short job is not held up
int Engine::chunk_pictures(const mvhp_stream_params_t &p) const
{
    if (opts_.chunk_pictures > 0) return opts_.chunk_pictures;
    const size_t pb = std::max<size_t>(1, mvhp_packed_frame_bytes(&p));
    return (int)std::min<size_t>(64, std::max<size_t>(1, ((size_t)64 << 20) / pb));
}

// pictures per launch.  Speed only.  The batch kernels want 4 x CUs (four pictures per workgroup) or 8 x CUs (eight)
// pictures, but the host entropy stage is the slow side and the download of a batch (9.4 MB per 1080p picture) is
// next: what matters is that downloads start early and that the job's last download is short.  So batches RAMP UP from 64
// pictures, doubling, to the cap (the download of batch k hides behind the entropy work of batch k+1 as long as batches
// do not shrink faster than the link is quicker than the entropy stage), and TAPER at the end (each takes at most 35 %
// of what is left per context, down to one picture per entropy thread): 16 32 64 128 256 512 364 236 154 100 65 42 27 18 16 18
// for 2048 pictures on one context and 16 threads (round 2 started and ended on 64);
// long jobs run most of their pictures in 1024-picture launches.  Modelled wall for 2048 x 1080p: 0.56 s against 0.63 s
// with 1024 512 256 128 64 64 and 0.535 s of pure entropy work.
int Engine::batch_capacity(const mvhp_stream_params_t &p, int remaining) const
{
    const int n_ctx = (int)ctx_.size();
    // the cap: four pictures per CU (the four-picture kernel's full round); a long job -- from 4096 pictures per context on --
    // runs its steady state in launches of 2048, eight per CU, which is what the eight-picture kernel wants (round 3: the
    // product path reaches the kernel the bench times; the ramp and the taper stay as they are)
    int cap = opts_.batch_pictures > 0 ? opts_.batch_pictures : ((n_order_ >= 4096 * n_ctx) ? 2048 : 1024);
    const size_t per_pic = compact_slot_bytes(p) + mvhp_packed_frame_bytes(&p) + mvhp_yuv_frame_bytes(&p) +
                           (want_rgb_ ? mvhp_rgb_frame_bytes(&p) : 0);
    size_t budget = ctx_[0].mem_budget;
    for (const Ctx &c : ctx_) budget = std::min(budget, c.mem_budget);
    const int mem_cap = (int)std::min<size_t>(1 << 20, std::max<size_t>(1, budget / std::max<size_t>(1, per_pic)));
    cap = std::min(cap, mem_cap);
    for (const Ctx &c : ctx_)
        if (c.arena && c.arena_pictures > 0 && same_params(c.arena_params, p)) cap = std::min(cap, c.arena_pictures);
    return planned_batch(cap, remaining, next_batch_id_);
}

// batches ramp up (the first pictures should not wait for a full-size batch to fill), run at the cap, and taper off (the
// last batch's upload, kernel and download are the tail nobody overlaps with)
int Engine::planned_batch(int cap, int remaining, int batch_id) const
{
    const int n_ctx = (int)ctx_.size();
    const int share = (remaining + n_ctx - 1) / n_ctx;
    if (opts_.batch_pictures > 0) return std::max(1, std::min(cap, share));   // an explicit batch size is taken as given
    const int round = batch_id / n_ctx;                                         // batches each context has been given so far
    // the ramp starts at, and the taper ends on, one picture per entropy thread: the threads finish such a batch together, and
    // the last batch's upload + kernel + download (the tail nothing overlaps) is 6 ms for 16 full-HD pictures against 15 ms
    // for 64.  Round 3: both were 64 -- a 64-picture job was ONE launch (36 ms; 26 ms in batches of 16), tools/e2e_ab.py.
    const int unit = std::max(8, std::min(64, host_threads_ / n_ctx));
    const int ramp = round < 10 ? std::min((long)unit << round, (long)cap) : cap;
    const int taper = std::max(unit, (int)((remaining * 0.35 + n_ctx - 1) / n_ctx));
    int b = std::max(1, std::min(std::min(cap, ramp), std::min(share, taper)));
    if (share - b > 0 && share - b < unit / 2 && share <= cap) b = share;   // (no two-picture batch behind the last full one)
    return b;
}

bool Engine::ensure_devbuf(Ctx &c, DevBuf &b, const Batch &bt, std::string &err)
{
    // MINIVIDEO_PLACED=1 (opt-in; for engines that live long: the arena is one allocation of up to 200 GB, seconds to get):
    // the context's three batch buffers come from mvhp_placed_alloc_sets -- records, planes and RGB of a batch each in a group
    // of the device's memory regions of its own, which is where the bench's kernel timings are taken (DESIGN 3 "Placement").
    // Once, for the first batch shape the context sees, sized for the largest batch the job can form; a batch that does not fit
    // (another shape, a larger explicit batch) uses ordinary allocations as before.
    if (placed_ && api_.placed_alloc && !c.arena_tried) {
        c.arena_tried = true;
        const int n = std::max(bt.capacity, job_cap_);
        const size_t bytes[4] = {(size_t)n * compact_slot_bytes(bt.params), (size_t)n * mvhp_packed_frame_bytes(&bt.params),
                                 (size_t)n * mvhp_yuv_frame_bytes(&bt.params), (size_t)n * mvhp_rgb_frame_bytes(&bt.params)};
        void *ptrs[12];
        const double t0 = now_s();
        c.arena = api_.placed_alloc(c.dev, 3, bytes, ptrs);
        {
            std::lock_guard<std::mutex> l(alloc_mu_);
            alloc_dev_s_ += now_s() - t0;
            if (c.arena) alloc_dev_bytes_ += 3 * (bytes[0] + bytes[1] + bytes[2] + bytes[3]);
        }
        if (c.arena) {
            c.arena_pictures = n;
            c.arena_params = bt.params;
            {   // what is left of the device now bounds every later ordinary allocation (a batch of another shape, another
                // context on this device): the budgets of all contexts on it are taken again
                const size_t free_b = api_.dev_free_bytes(c.dev);
                for (Ctx &o : ctx_) {
                    if (o.device != c.device) continue;
                    int sharers = 0;
                    for (const Ctx &j : ctx_) sharers += j.device == c.device;
                    o.mem_budget = std::min(o.mem_budget, free_b / 2 / (size_t)o.n_bufs / (size_t)std::max(1, sharers));
                }
            }
            for (int k = 0; k < 3; k++) {
                DevBuf &d = c.bufs[k];
                // (ordinary buffers from an earlier call of another shape are released; none is in use: the first batch)
                if (d.compact) api_.dev_free(c.dev, d.compact);
                if (d.packed) api_.dev_free(c.dev, d.packed);
                if (d.yuv) api_.dev_free(c.dev, d.yuv);
                if (d.rgb) api_.dev_free(c.dev, d.rgb);
                d.compact = ptrs[k * 4 + 0]; d.compact_cap = bytes[0];
                d.packed = ptrs[k * 4 + 1]; d.packed_cap = bytes[1];
                d.yuv = (uint8_t *)ptrs[k * 4 + 2]; d.yuv_cap = bytes[2];
                d.rgb = (uint8_t *)ptrs[k * 4 + 3]; d.rgb_cap = bytes[3];
                d.arena_piece = true;
            }
        }
    }
    if (b.arena_piece) {
        const size_t n = (size_t)bt.capacity;
        if (n * compact_slot_bytes(bt.params) <= b.compact_cap && n * mvhp_packed_frame_bytes(&bt.params) <= b.packed_cap &&
            n * mvhp_yuv_frame_bytes(&bt.params) <= b.yuv_cap && n * mvhp_rgb_frame_bytes(&bt.params) <= b.rgb_cap)
            return true;
        // a batch the arena was not sized for (another picture size, a later and longer job): this batch buffer leaves the
        // arena for ordinary allocations -- its pieces stay where they are until the engine goes (batch_capacity() keeps
        // batches of the arena's own shape inside it)
        b = DevBuf();
        b.busy = true;
    }
    auto need = [&](void **ptr, size_t *cap, size_t bytes) {
        if (*cap >= bytes) return true;
        const double t0 = now_s();
        if (*ptr) api_.dev_free(c.dev, *ptr);
        *ptr = api_.dev_alloc(c.dev, bytes);
        *cap = *ptr ? bytes : 0;
        {
            std::lock_guard<std::mutex> l(alloc_mu_);
            alloc_dev_s_ += now_s() - t0;
            alloc_dev_bytes_ += bytes;
        }
        return *ptr != nullptr;
    };
    // sized for the largest batch this call can form on a context, not for the batch at hand: the ramp and the taper hand a
    // buffer batches of changing sizes, and every growth is a hipFree + hipMalloc in the middle of the pipeline (hipFree
    // waits for the device) -- the second and third call of an engine still paid for that (0.55 s, then 0.51 s, same job)
    const size_t n = (size_t)std::max(bt.capacity, (bt.params.width_mbs == job_params_.width_mbs && bt.params.height_mbs == job_params_.height_mbs) ? job_cap_ : 0);
    if (!need(&b.compact, &b.compact_cap, n * compact_slot_bytes(bt.params)) ||
        !need(&b.packed, &b.packed_cap, n * mvhp_packed_frame_bytes(&bt.params)) ||
        !need((void **)&b.yuv, &b.yuv_cap, n * mvhp_yuv_frame_bytes(&bt.params)) ||
        (want_rgb_ && !need((void **)&b.rgb, &b.rgb_cap, n * mvhp_rgb_frame_bytes(&bt.params)))) {
        err = "out of device memory for a batch of " + std::to_string(bt.capacity) + " pictures";
        return false;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// feeder
// ---------------------------------------------------------------------------------------------------------------
// The batch gets no more pictures.  Usually the uploader sees the total when it finishes the closing chunk; a batch
// closed after its last chunk went up is handed on here.
void Engine::close_batch(Batch *b)
{
    b->total = (int)b->seqs.size();
    b->t_closed = now_s() - t_start_;
    st_.max_batch_pictures = std::max(st_.max_batch_pictures, (uint32_t)b->total);
    if (b->ctx >= 0 && b->uploaded == b->total) {
        Ctx &cx = ctx_[(size_t)b->ctx];
        if (cx.open_batch == b->id) cx.open_batch = -1;
        if (b->dead) fail_batch(b, b->dead_why);
        else cx.to_launch.push_back(b->id);
    }
}

void Engine::feeder()
{
    std::unique_lock<std::mutex> l(mu_);
    Batch *cur = nullptr;   // the open batch
    for (;;) {
        RetryGroup *rg = retry_q_.empty() ? nullptr : &retry_q_.front();
        const bool have = rg || (pos_ < n_order_ && allowance() > 0);
        // an open batch is closed as soon as it cannot take the next picture: nothing may be issued right now, or the
        // next pictures are a re-queued group (they go into a batch of their own, kept off the context they failed on)
        if (cur && (stop_ || !have || cur->retry != (rg != nullptr) || (rg && rg->exclude_ctx != cur->exclude_ctx))) {
            close_batch(cur);
            cur = nullptr;
            cv_.notify_all();
        }
        if (stop_) return;
        if (!have) { cv_.wait(l); continue; }
        const int first_seq = rg ? rg->seqs[rg->pos] : pos_;
        mvhp_stream_params_t p0{};
        if (mvhp_stream_params(s_, order_[first_seq], &p0) != MVHP_SUCCESS) {
            // a picture whose parameter sets never arrived fails without occupying a slot (never a re-queued one)
            PicResult &r = results_[(size_t)first_seq];
            const mvhp_stream::Idr &idr = s_->idrs[(size_t)order_[first_seq]];
            r.rc = MVHP_FAILURE;
            r.err = idr.why.empty() ? "parameter sets missing" : idr.why;
            r.ready = true;
            if (rg) { if (++rg->pos >= rg->seqs.size()) retry_q_.pop_front(); }
            else { pos_++; issued_++; }
            cv_.notify_all();
            continue;
        }
        if (cur && !same_params(cur->params, p0)) {   // a batch holds one set of stream parameters
            close_batch(cur);
            cur = nullptr;
        }
        const int avail = rg ? (int)(rg->seqs.size() - rg->pos) : std::min(allowance(), n_order_ - pos_);
        if (!cur) {
            auto nb = std::make_unique<Batch>();
            nb->id = next_batch_id_++;
            nb->params = p0;
            nb->capacity = batch_capacity(p0, avail);
            nb->retry = rg != nullptr;
            nb->exclude_ctx = rg ? rg->exclude_ctx : -1;
            nb->seqs.reserve((size_t)nb->capacity);
            nb->t_open = now_s() - t_start_;
            cur = nb.get();
            batches_[cur->id] = std::move(nb);
        }
        const int C = std::min(chunk_pictures(p0), cur->capacity);
        int n = std::min(std::min(C, avail), cur->capacity - (int)cur->seqs.size());
        auto seq_at = [&](int i) { return rg ? rg->seqs[rg->pos + (size_t)i] : pos_ + i; };
        int same = 1;
        while (same < n) {
            mvhp_stream_params_t pi{};
            if (mvhp_stream_params(s_, order_[seq_at(same)], &pi) != MVHP_SUCCESS || !same_params(pi, p0)) break;
            same++;
        }
        n = same;
        // a free page-locked chunk (the pool grows up to its limit; page-locking happens outside the lock)
        InChunk *c = nullptr;
        const size_t pic_bytes = compact_slot_bytes(p0);
        for (;;) {
            if (stop_) break;
            if (!free_in_.empty()) { c = free_in_.front(); free_in_.pop_front(); break; }
            if (all_in_.size() < in_limit_) {
                all_in_.push_back(std::make_unique<InChunk>());
                c = all_in_.back().get();
                break;
            }
            const double w0 = now_s();
            cv_.wait(l);
            feeder_wait_chunk_ += now_s() - w0;
        }
        if (!c) continue;   // stopping: the head of the loop closes the batch
        bool have_mem = true;
        if (c->buf.cap < (size_t)C * pic_bytes) {
            l.unlock();
            have_mem = grow(c->buf, (size_t)C * pic_bytes);
            l.lock();
            // `rg` is NOT fetched again: a group another context pushed meanwhile waits for the next iteration (`avail` and
            // `n` were sized without it), and push_back keeps references to a deque's elements valid -- only this
            // thread pops
        }
        if (!have_mem) {   // no page-locked memory: these pictures fail
            free_in_.push_back(c);
            for (int i = 0; i < n; i++) {
                PicResult &r = results_[(size_t)seq_at(i)];
                r.rc = MVHP_FAILURE;
                r.err = "out of page-locked host memory";
                r.ready = true;
            }
        } else {
            c->batch = cur->id;
            c->first_slot = (int)cur->seqs.size();
            c->n = n;
            c->pic_bytes = pic_bytes;
            c->used.assign((size_t)n, 0);
            c->remaining = n;
            for (int i = 0; i < n; i++) {
                const int seq = seq_at(i);
                cur->seqs.push_back(seq);
                PicResult &r = results_[(size_t)seq];
                r.params = p0;
                r.parsed_ok = false;
                work_q_.push_back(Item{c, i, seq});
            }
            st_.pictures_issued += (uint32_t)n;
        }
        if (rg) {
            rg->pos += (size_t)n;
            if (rg->pos >= rg->seqs.size()) {
                retry_q_.pop_front();
                if (cur) { close_batch(cur); cur = nullptr; }   // one batch per re-queued group
            }
        } else {
            pos_ += n;
            issued_ += n;
        }
        if (cur && (int)cur->seqs.size() >= cur->capacity) {
            close_batch(cur);
            cur = nullptr;
        }
        if (cur && cur->seqs.empty()) {   // (only after an allocation failure on a batch's first chunk)
            batches_.erase(cur->id);
            cur = nullptr;
        }
        cv_.notify_all();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// entropy workers
// ---------------------------------------------------------------------------------------------------------------
void Engine::worker(int t)
{
    for (;;) {
        Item it;
        {
            std::unique_lock<std::mutex> l(mu_);
            const double w0 = now_s();
            cv_.wait(l, [&] { return stop_ || !work_q_.empty(); });
            if (stop_) return;
            worker_wait_[(size_t)t] += now_s() - w0;
            it = work_q_.front();
            work_q_.pop_front();
        }
        const int idr = order_[it.seq];
        std::string err;
        const double t0 = now_s();
        uint8_t *slot = it.chunk->buf.p + (size_t)it.slot * it.chunk->pic_bytes;
        size_t used = 0;
        const int rc = s_->decode_compact(idr, slot, it.chunk->pic_bytes, &used, err);
        if (rc != h264::RC_SUCCESS) {   // the slot still has to expand to something: an empty picture
            const mvhp_stream::Idr &d = s_->idrs[(size_t)idr];
            used = write_empty_compact(slot, (size_t)d.sps.width_mbs * (size_t)d.sps.height_map_units);
        }
        it.chunk->used[(size_t)it.slot] = used;   // (each slot has one writer; read by the uploader behind the engine mutex)
        worker_busy_[(size_t)t] += now_s() - t0;
        bool wake = false;
        if (idr >= 0 && (size_t)idr < s_->idrs.size()) stream_bytes_ += s_->samples[s_->idrs[(size_t)idr].sample].nal_size;
        {
            std::lock_guard<std::mutex> l(mu_);
            PicResult &r = results_[(size_t)it.seq];
            t_last_entropy_ = now_s() - t_start_;
            if (rc == h264::RC_SUCCESS) {
                r.parsed_ok = true;
            } else {   // final: entropy decoding is deterministic, a second try would fail the same way
                r.parsed_ok = false;
                r.rc = rc;
                r.err = err;
                r.ready = true;
            }
            wake = rc != h264::RC_SUCCESS;   // a final result: the sink may be waiting for exactly this picture
            if (--it.chunk->remaining == 0) {
                auto bi = batches_.find(it.chunk->batch);
                if (bi != batches_.end()) bi->second->ready.push_back(it.chunk);
                else free_in_.push_back(it.chunk);
                wake = true;                 // a chunk for the uploader (or back in the pool for the feeder)
            }
        }
        // (every waiter shares one condition variable: waking them per picture cost the sixteen entropy threads' CPU quota --
        //  a decoded picture inside an unfinished chunk changes nothing anyone waits for)
        if (wake) cv_.notify_all();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// per-context stages
// ---------------------------------------------------------------------------------------------------------------
bool Engine::pick_chunk(int k, InChunk **c, Batch **b)
{
    Ctx &cx = ctx_[(size_t)k];
    if (cx.open_batch >= 0) {
        Batch *ob = batches_[cx.open_batch].get();
        if (ob->ready.empty()) return false;
        *c = ob->ready.front();
        ob->ready.pop_front();
        *b = ob;
        return true;
    }
    DevBuf *fb = nullptr;
    for (int i = 0; i < cx.n_bufs; i++)
        if (!cx.bufs[i].busy) { fb = &cx.bufs[i]; break; }
    if (!fb) return false;
    for (auto &kv : batches_) {   // lowest id first: batches are claimed in the order they were opened
        Batch *nb = kv.second.get();
        if (nb->ctx >= 0 || nb->ready.empty()) continue;
        if (nb->exclude_ctx == k) continue;
        nb->ctx = k;
        nb->t_claim = now_s() - t_start_;
        nb->buf = fb;
        fb->busy = true;
        cx.open_batch = nb->id;
        *c = nb->ready.front();
        nb->ready.pop_front();
        *b = nb;
        return true;
    }
    return false;
}

void Engine::release_batch(Batch *b)
{
    if (b->buf) b->buf->busy = false;
    for (InChunk *c : b->ready) free_in_.push_back(c);
    batches_.erase(b->id);
}

// A batch that failed on its context: entropy-decode its pictures again and queue them for another context, once.
void Engine::fail_batch(Batch *b, const std::string &why)
{
    std::vector<int> open;
    const int n = b->total >= 0 ? b->total : (int)b->seqs.size();
    for (int i = 0; i < n; i++) {
        const PicResult &r = results_[(size_t)b->seqs[(size_t)i]];
        if (r.parsed_ok && !r.ready) open.push_back(b->seqs[(size_t)i]);
    }
    if (!b->retry && ctx_.size() > 1 && !open.empty()) {
        RetryGroup g;
        g.seqs = open;
        g.exclude_ctx = b->ctx;
        retry_q_.push_back(std::move(g));
        st_.batches_requeued++;
    } else {
        for (int seq : open) {
            PicResult &r = results_[(size_t)seq];
            r.rc = MVHP_FAILURE;
            r.err = why;
            r.ready = true;
        }
    }
    release_batch(b);
}

void Engine::uploader(int k)
{
    Ctx &cx = ctx_[(size_t)k];
    for (;;) {
        InChunk *c = nullptr;
        Batch *b = nullptr;
        {
            std::unique_lock<std::mutex> l(mu_);
            cv_.wait(l, [&] { return stop_ || pick_chunk(k, &c, &b); });
            if (stop_) return;
        }
        // b stays alive: only this context's stages release a batch it has claimed
        std::string err;
        float ms = 0.f;
        bool ok = !b->dead;
        if (ok) ok = ensure_devbuf(cx, *b->buf, *b, err);   // sized once per batch (capacity is fixed when it opens)
        if (ok) {
            std::vector<void *> dst((size_t)c->n);
            std::vector<const void *> src((size_t)c->n);
            size_t bytes = 0;
            for (int i = 0; i < c->n; i++) {   // only what the entropy stage wrote crosses the link
                dst[(size_t)i] = (uint8_t *)b->buf->compact + (size_t)(c->first_slot + i) * c->pic_bytes;
                src[(size_t)i] = c->buf.p + (size_t)i * c->pic_bytes;
                bytes += c->used[(size_t)i];
            }
            ok = api_.h2d(cx.dev, c->n, dst.data(), src.data(), c->used.data(), &ms, err) == MVHP_SUCCESS;
            if (ok) {
                std::lock_guard<std::mutex> l(mu_);
                st_.h2d_s += ms * 1e-3;
                st_.h2d_bytes += bytes;
            }
        }
        {
            std::lock_guard<std::mutex> l(mu_);
            if (!ok && !b->dead) { b->dead = true; b->dead_why = err; }
            b->uploaded += c->n;
            free_in_.push_back(c);
            if (b->total >= 0 && b->uploaded == b->total) {
                cx.open_batch = -1;
                b->t_uploaded = now_s() - t_start_;
                if (b->dead) fail_batch(b, b->dead_why);
                else cx.to_launch.push_back(b->id);
            }
        }
        cv_.notify_all();
    }
}

void Engine::launcher(int k)
{
    Ctx &cx = ctx_[(size_t)k];
    for (;;) {
        Batch *b = nullptr;
        bool inject = false;
        {
            std::unique_lock<std::mutex> l(mu_);
            cv_.wait(l, [&] { return stop_ || !cx.to_launch.empty(); });
            if (stop_) return;
            b = batches_[cx.to_launch.front()].get();
            cx.to_launch.pop_front();
            if (cx.fail_next) { cx.fail_next = false; inject = true; }
        }
        std::string err;
        float ms = 0.f;
        int layout = 0, waves = 0;
        int rc = MVHP_SUCCESS;
        const double t_call = now_s();
        if (inject) { rc = MVHP_FAILURE; err = "injected failure (test hook)"; }
        else rc = api_.recon(cx.dev, &b->params, b->buf->compact, compact_slot_bytes(b->params), b->buf->packed, b->total,
                             b->buf->yuv, want_rgb_ ? b->buf->rgb : nullptr, &ms, &layout, &waves, err);
        if (!cx.launched_once) {   // host time of the first call beyond the device time: code-object load, first-launch setup
            cx.launched_once = true;
            std::lock_guard<std::mutex> la(alloc_mu_);
            first_launch_s_ += std::max(0.0, (now_s() - t_call) - ms * 1e-3);
        }
        {
            std::lock_guard<std::mutex> l(mu_);
            if (rc == MVHP_SUCCESS) {
                st_.batches++;
                st_.kernel_s += ms * 1e-3;
                if (layout >= 0 && layout < 4) st_.launches_by_layout[layout]++;
                else if (layout >= MVHP_LAYOUT_WIDE && layout <= MVHP_LAYOUT_PIPE1) st_.launches_wide[layout - MVHP_LAYOUT_WIDE]++;
                b->t_kernel = now_s() - t_start_;
                cx.to_download.push_back(b->id);
            } else {
                fail_batch(b, err);
            }
        }
        cv_.notify_all();
    }
}

void Engine::release_picture(int seq)
{
    {
        std::lock_guard<std::mutex> l(mu_);
        if (seq >= 0 && seq == in_sink_) { released_early_ = true; return; }   // its callback is still running: see the sink loop
        if (seq < 0 || (size_t)seq >= results_.size() || !results_[(size_t)seq].kept) return;
        PicResult &r = results_[(size_t)seq];
        r.kept = false;
        if (r.oc && --r.oc->refs == 0) put_out(r.oc);
        r.oc = nullptr;
        kept_--;
    }
    cv_.notify_all();
}

void Engine::put_out(OutChunk *oc)
{
    free_out_.push_back(oc);
}

void Engine::downloader(int k)
{
    Ctx &cx = ctx_[(size_t)k];
    for (;;) {
        Batch *b = nullptr;
        {
            std::unique_lock<std::mutex> l(mu_);
            cv_.wait(l, [&] { return stop_ || !cx.to_download.empty(); });
            if (stop_) return;
            b = batches_[cx.to_download.front()].get();
            cx.to_download.pop_front();
        }
        const size_t yb = mvhp_yuv_frame_bytes(&b->params), rb = want_rgb_ ? mvhp_rgb_frame_bytes(&b->params) : 0;
        const int C = std::min(chunk_pictures(b->params), b->capacity);
        std::string fail;
        for (int g = 0; g < b->total && fail.empty(); g += C) {
            const int n = std::min(C, b->total - g);
            OutChunk *oc = nullptr;
            {
                // A free output chunk.  The pool is bounded, except that a starved sink lets it grow: every parked
                // chunk is then ahead of the picture the sink waits for, and that picture may need a chunk itself
                // (re-queued batches run behind later ones).
                std::unique_lock<std::mutex> l(mu_);
                for (;;) {
                    if (stop_) return;
                    if (!free_out_.empty()) { oc = free_out_.front(); free_out_.pop_front(); break; }
                    if (all_out_.size() < out_limit_ || sink_waiting_) {
                        all_out_.push_back(std::make_unique<OutChunk>());
                        oc = all_out_.back().get();
                        break;
                    }
                    cv_.wait(l);
                }
            }
            std::string err;
            float ms = 0.f, ms2 = 0.f;
            bool ok = (!want_yuv_ || grow(oc->yuv, (size_t)C * yb)) && (!want_rgb_ || grow(oc->rgb, (size_t)C * rb));
            if (!ok) err = "out of page-locked host memory";
            if (ok) {
                void *dst[2];
                const void *src[2];
                size_t nb[2];
                int np = 0;
                if (want_yuv_) { dst[np] = oc->yuv.p; src[np] = b->buf->yuv + (size_t)g * yb; nb[np++] = (size_t)n * yb; }
                if (want_rgb_) { dst[np] = oc->rgb.p; src[np] = b->buf->rgb + (size_t)g * rb; nb[np++] = (size_t)n * rb; }
                ok = api_.d2h(cx.dev, np, dst, src, nb, &ms, err) == MVHP_SUCCESS;
            }
            {
                std::lock_guard<std::mutex> l(mu_);
                if (ok) {
                    st_.d2h_s += (ms + ms2) * 1e-3;
                    st_.d2h_bytes += (uint64_t)n * ((want_yuv_ ? yb : 0) + rb);
                    oc->refs = 0;
                    for (int i = 0; i < n; i++) {
                        PicResult &r = results_[(size_t)b->seqs[(size_t)(g + i)]];
                        if (!r.parsed_ok || r.ready) continue;
                        r.rc = MVHP_SUCCESS;
                        r.oc = oc;
                        r.yuv = want_yuv_ ? oc->yuv.p + (size_t)i * yb : nullptr;
                        r.rgb = want_rgb_ ? oc->rgb.p + (size_t)i * rb : nullptr;
                        r.ready = true;
                        oc->refs++;
                    }
                    if (oc->refs == 0) put_out(oc);
                } else {
                    put_out(oc);
                    fail = err;
                }
            }
            cv_.notify_all();
        }
        {
            std::lock_guard<std::mutex> l(mu_);
            t_last_download_ = now_s() - t_start_;
            if (getenv("MINIVIDEO_ENGINE_TRACE"))
                fprintf(stderr, "engine trace: batch %d: %d pictures, opened %.4f claimed %.4f closed %.4f uploaded %.4f kernel done %.4f downloaded %.4f\n",
                        b->id, b->total, b->t_open, b->t_claim, b->t_closed, b->t_uploaded, b->t_kernel, t_last_download_);
            if (!fail.empty()) fail_batch(b, fail);   // pictures already delivered stay delivered
            else release_batch(b);
        }
        cv_.notify_all();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// one decode call
// ---------------------------------------------------------------------------------------------------------------
int Engine::decode(const mvhp_stream &s, const int *order, int n_order, int wanted, int out_mask, mvhp_picture_sink_t sink,
                   void *user, mvhp_decode_stats_t *stats, std::string &err)
{
    if (!order || n_order <= 0 || wanted <= 0) { err = "nothing to decode"; return MVHP_FAILURE; }
    for (int i = 0; i < n_order; i++)
        if (order[i] < 0 || (size_t)order[i] >= s.idrs.size()) { err = "IDR index out of range"; return MVHP_FAILURE; }
    const double t_start = now_s();
    const int n_ctx = (int)ctx_.size();
    {
        std::lock_guard<std::mutex> l(mu_);
        s_ = &s; order_ = order; n_order_ = n_order; wanted_ = std::min(wanted, n_order);
        want_rgb_ = (out_mask & 1) != 0;
        want_yuv_ = !want_rgb_ || (out_mask & 2) == 0;
        stop_ = false; sink_waiting_ = false;
        pos_ = issued_ = consumed_ = ok_ = failed_ = 0;
        next_batch_id_ = 0;
        results_.assign((size_t)n_order, PicResult());
        for (int i = 0; i < n_order; i++) results_[(size_t)i].idr = order[i];
        batches_.clear(); work_q_.clear(); retry_q_.clear();
        free_in_.clear(); free_out_.clear();
        for (auto &c : all_in_) free_in_.push_back(c.get());
        for (auto &c : all_out_) { c->refs = 0; free_out_.push_back(c.get()); }
        for (Ctx &c : ctx_) { c.open_batch = -1; c.to_launch.clear(); c.to_download.clear(); for (DevBuf &d : c.bufs) d.busy = false; c.fail_next = false; }
        if (opts_.fail_context >= 0 && opts_.fail_context < n_ctx) ctx_[(size_t)opts_.fail_context].fail_next = true;
        memset(&st_, 0, sizeof(st_));
        worker_busy_.assign((size_t)host_threads_, 0.0);
        worker_wait_.assign((size_t)host_threads_, 0.0);
        feeder_wait_chunk_ = t_last_entropy_ = t_last_download_ = 0;
        t_start_ = t_start;
        stream_bytes_ = 0;
        {
            std::lock_guard<std::mutex> la(alloc_mu_);
            alloc_host_s_ = alloc_dev_s_ = first_launch_s_ = first_picture_s_ = 0;
            alloc_host_bytes_ = alloc_dev_bytes_ = 0;
        }
        // pools: enough input chunks that every entropy thread has a slot to write while the earlier chunks upload (three
        // pictures per thread were measured too: no gain, and page-locking the extra chunks costs the first call 0.1 s), and
        // a few output chunks per context
        mvhp_stream_params_t p0{};
        int C = 8;
        for (int i = 0; i < n_order; i++)
            if (mvhp_stream_params(&s, order[i], &p0) == MVHP_SUCCESS) { C = chunk_pictures(p0); break; }
        in_limit_ = (size_t)std::max(3, (host_threads_ + C - 1) / C + 1 + n_ctx);
        if (const int e = env_int("MINIVIDEO_IN_CHUNKS", 0)) in_limit_ = (size_t)std::max(2, std::min(64, e));   // developer aid
        {   // the largest batch this call can form on a context: what a placed arena is sized for
            int cap = opts_.batch_pictures > 0 ? opts_.batch_pictures : ((n_order_ >= 4096 * n_ctx) ? 2048 : 1024);
            const size_t per_pic = compact_slot_bytes(p0) + mvhp_packed_frame_bytes(&p0) + mvhp_yuv_frame_bytes(&p0) + mvhp_rgb_frame_bytes(&p0);
            size_t budget = ctx_[0].mem_budget;
            for (const Ctx &c : ctx_) budget = std::min(budget, c.mem_budget);
            cap = std::min<int>(cap, (int)std::min<size_t>(1 << 20, std::max<size_t>(1, budget / std::max<size_t>(1, per_pic))));
            cap = std::max(1, std::min(cap, (wanted_ + n_ctx - 1) / n_ctx));
            job_cap_ = 1;   // the largest batch the ramp / cap / taper will actually form when nothing fails
            for (int rem = wanted_, id = 0; rem > 0; id++) {
                const int b = planned_batch(cap, rem, id);
                job_cap_ = std::max(job_cap_, b);
                rem -= b;
            }
            job_params_ = p0;
        }
        out_limit_ = (size_t)(2 * n_ctx + 2);
    }
    const int threads = std::min(host_threads_, std::max(1, wanted_));
    std::vector<std::thread> th;
    th.emplace_back([this] { feeder(); });
    for (int t = 0; t < threads; t++) th.emplace_back([this, t] { worker(t); });
    for (int k = 0; k < n_ctx; k++) {
        th.emplace_back([this, k] { uploader(k); });
        th.emplace_back([this, k] { launcher(k); });
        th.emplace_back([this, k] { downloader(k); });
    }

    // ---- sink loop (this thread) ----
    bool aborted = false;
    double sink_s = 0;
    for (int next = 0; next < n_order;) {
        PicResult r;
        {
            std::unique_lock<std::mutex> l(mu_);
            if (ok_ >= wanted_) break;
            if (!results_[(size_t)next].ready) {
                sink_waiting_ = true;   // (lets the downloader grow its pool rather than deadlock behind a re-queued batch)
                cv_.notify_all();
                cv_.wait(l, [&] { return results_[(size_t)next].ready; });
                sink_waiting_ = false;
            }
            r = results_[(size_t)next];
            in_sink_ = next;
            released_early_ = false;
        }
        if (next == 0) first_picture_s_ = now_s() - t_start;
        int verdict = (r.rc == MVHP_SUCCESS) ? 1 : 0;
        if (sink) {
            const double t0 = now_s();
            verdict = sink(user, next, r.idr, r.rc, r.err.c_str(), &r.params, r.yuv, r.rgb);
            sink_s += now_s() - t0;
        }
        bool wake = false;
        {
            std::lock_guard<std::mutex> l(mu_);
            in_sink_ = -1;
            if (verdict == 2 && r.rc == MVHP_SUCCESS && r.oc && !released_early_) {   // kept by the sink: the chunk stays out until
                results_[(size_t)next].kept = true;                                    // release_picture(next)
                kept_++;
                verdict = 1;
            } else {   // (released_early_: the other thread was done with it before the callback had returned)
                if (verdict == 2) verdict = (r.rc == MVHP_SUCCESS) ? 1 : 0;
                if (r.oc && --r.oc->refs == 0) { put_out(r.oc); wake = true; }   // an output chunk for the downloader
                results_[(size_t)next].oc = nullptr;
            }
            consumed_++;
            if (r.rc == MVHP_SUCCESS && verdict == 1) ok_++;
            else { failed_++; wake = true; }                                  // the allowance grew: the feeder may issue another picture
        }
        if (wake) cv_.notify_all();
        next++;
        if (verdict < 0) { aborted = true; break; }
    }
    {
        std::lock_guard<std::mutex> l(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : th) t.join();
    {   // pictures the sink kept (verdict 2) live in this engine's output chunks: the call ends when the last one is back
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return kept_ == 0; });
    }
    {
        std::lock_guard<std::mutex> l(mu_);
        batches_.clear();
        work_q_.clear();
        st_.pictures_ok = (uint32_t)ok_;
        st_.pictures_failed = (uint32_t)failed_;
        st_.contexts = (uint32_t)n_ctx;
        st_.host_threads = (uint32_t)threads;
        st_.sink_s = sink_s;
        st_.stream_bytes = stream_bytes_;
        for (double b : worker_busy_) st_.entropy_busy_s += b;
        st_.wall_s = now_s() - t_start;
        {
            std::lock_guard<std::mutex> la(alloc_mu_);
            st_.host_alloc_s = alloc_host_s_; st_.dev_alloc_s = alloc_dev_s_; st_.first_launch_s = first_launch_s_;
            st_.first_picture_s = first_picture_s_;
            st_.host_alloc_bytes = alloc_host_bytes_; st_.dev_alloc_bytes = alloc_dev_bytes_;
        }
        for (const Ctx &c : ctx_) if (c.arena) st_.placed_buffers = 1;
        if (getenv("MINIVIDEO_ENGINE_TRACE")) {   // where the wall time went (developer aid)
            double wait = 0, wmax = 0, bmin = 1e30, bmax = 0;
            for (double w : worker_wait_) { wait += w; wmax = std::max(wmax, w); }
            for (double b : worker_busy_) { bmin = std::min(bmin, b); bmax = std::max(bmax, b); }
            fprintf(stderr, "engine trace: wall %.4f s, threads %d: busy min %.4f max %.4f, waiting for work sum %.4f max %.4f; feeder waited %.4f s for a free "
                            "input chunk; last entropy result at %.4f s, last download at %.4f s, sink %.4f s, batches %u\n",
                    st_.wall_s, threads, bmin, bmax, wait, wmax, feeder_wait_chunk_, t_last_entropy_, t_last_download_, sink_s, st_.batches);
        }
        if (stats) *stats = st_;
        s_ = nullptr; order_ = nullptr;
    }
    if (aborted) { err = "stopped by the sink"; return MVHP_FAILURE; }
    if (ok_ >= wanted_ || ok_ > 0) return MVHP_SUCCESS;
    err = "no picture could be decoded";
    return MVHP_FAILURE;
}

Engine *engine_create(const DeviceApi &api, const mvhp_engine_opts_t *opts, std::string &err)
{
    Engine *e = new Engine(api);
    if (!e->init(opts, err)) { delete e; return nullptr; }
    return e;
}

void engine_destroy(Engine *e) { delete e; }
void engine_release_picture(Engine *e, int seq) { e->release_picture(seq); }

int engine_decode(Engine *e, const mvhp_stream &s, const int *order, int n_order, int wanted, int out_mask,
                  mvhp_picture_sink_t sink, void *user, mvhp_decode_stats_t *stats, std::string &err)
{
    if (!e) { err = "no engine"; return MVHP_FAILURE; }
    return e->decode(s, order, n_order, wanted, out_mask, sink, user, stats, err);
}

} // namespace mvengine

// ---- C-ABI (include/minivideo_hotpath.h) ----
struct mvhp_engine {
    mvengine::Engine *e = nullptr;
};

static thread_local std::string g_engine_err;

extern "C" {

MVHP_EXPORT int mvhp_engine_create(const mvhp_engine_opts_t *opts, mvhp_engine_t **out)
{
    if (!out) return MVHP_FAILURE;
    *out = nullptr;
    mvengine::Engine *e = mvengine::engine_create(mvhp_hip_device_api(), opts, g_engine_err);
    if (!e) {
        fprintf(stderr, "[minivideo] %s\n", g_engine_err.c_str());
        return MVHP_FAILURE;
    }
    *out = new mvhp_engine{e};
    return MVHP_SUCCESS;
}

MVHP_EXPORT void mvhp_engine_destroy(mvhp_engine_t *h)
{
    if (!h) return;
    mvengine::engine_destroy(h->e);
    delete h;
}

MVHP_EXPORT void mvhp_engine_release_picture(mvhp_engine_t *h, int seq)
{
    if (h) mvengine::engine_release_picture(h->e, seq);
}

MVHP_EXPORT int mvhp_engine_decode(mvhp_engine_t *h, const mvhp_stream_t *s, const int *order, int n_order, int wanted,
                                   int want_rgb, mvhp_picture_sink_t sink, void *user, mvhp_decode_stats_t *stats)
{
    if (!h || !s) return MVHP_FAILURE;
    const int rc = mvengine::engine_decode(h->e, *s, order, n_order, wanted, want_rgb, sink, user, stats, g_engine_err);
    if (rc != MVHP_SUCCESS) fprintf(stderr, "[minivideo] %s\n", g_engine_err.c_str());
    return rc;
}

} // extern "C"
