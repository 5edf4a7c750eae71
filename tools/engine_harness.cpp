// engine_harness.cpp -- the decode engine's threading, built WITHOUT HIP against a stub device table, so that it can
// run under ThreadSanitizer on a CPU-only box (tests/test_engine_harness.py).  TEST INFRASTRUCTURE: the stub "device"
// does no reconstruction -- it stamps every output picture with a checksum of the packed records it was handed, which
// lets the sink check that the right picture arrives in the right place, in order, through chunks, batches, several
// contexts and a re-queue.  Nothing here is part of libminivideo.so.
//
// usage: engine_harness <stream.264> <scenario>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "decode_engine.h"
#include "minivideo.h"
#include "stream_internal.h"

// ---- the pieces of the C-ABI the engine uses that live in the HIP translation unit of the product ----
extern "C" {
size_t mvhp_packed_frame_bytes(const mvhp_stream_params_t *p) { return p ? (size_t)p->width_mbs * p->height_mbs * MVHP_MB_BYTES : 0; }
size_t mvhp_yuv_frame_bytes(const mvhp_stream_params_t *p) { return p ? (size_t)p->width_mbs * p->height_mbs * 384 : 0; }
size_t mvhp_rgb_frame_bytes(const mvhp_stream_params_t *p) { return p ? (size_t)p->width_mbs * p->height_mbs * 768 : 0; }
int mvhp_device_count(void) { return 1; }
}

namespace mvengine {
struct DevCtx {
    int device;
};
} // namespace mvengine

namespace {

using mvengine::DevCtx;

std::atomic<int> g_recon_calls{0}, g_live_ctx{0};
std::atomic<long> g_dev_allocs{0};
int g_devices = 2;
std::atomic<int> g_slow_host_alloc_us{0};   // page-locking is slow on a real box: the feeder drops its lock around it

uint64_t checksum(const uint8_t *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

int stub_device_count() { return g_devices; }
void *stub_host_alloc(size_t n)
{
    if (g_slow_host_alloc_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(g_slow_host_alloc_us.load()));
    return malloc(n ? n : 1);
}
void stub_host_free(void *p) { free(p); }
DevCtx *stub_ctx_create(int device, std::string &) { g_live_ctx++; return new DevCtx{device}; }
void stub_ctx_destroy(DevCtx *c) { g_live_ctx--; delete c; }
void *stub_dev_alloc(DevCtx *, size_t n) { g_dev_allocs++; return malloc(n ? n : 1); }
void stub_dev_free(DevCtx *, void *p) { g_dev_allocs--; free(p); }
size_t stub_dev_free_bytes(DevCtx *) { return (size_t)1 << 32; }
int stub_copy_n(DevCtx *, int n, void *const *dst, const void *const *src, const size_t *bytes, float *ms, std::string &)
{
    for (int i = 0; i < n; i++) memcpy(dst[i], src[i], bytes[i]);
    std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (ms) *ms = 0.05f;
    return MVHP_SUCCESS;
}
// compact picture -> packed records (the CPU restatement of csrc/hip/expand_compact.hip, for this harness only)
void expand_compact(const uint8_t *pic, size_t mbs, uint8_t *packed)
{
    const uint32_t *off = reinterpret_cast<const uint32_t *>(pic);
    for (size_t mb = 0; mb < mbs; mb++) {
        const uint8_t *rec = pic + mbs * 4 + off[mb];
        uint8_t *out = packed + mb * MVHP_MB_BYTES;
        memset(out, 0, MVHP_MB_BYTES);
        memcpy(out, rec, MVHP_MB_HEADER_BYTES);
        uint32_t n;
        memcpy(&n, rec + 28, 4);
        const bool dense = (rec[5] & 1) != 0;
        out[5] = 0;
        memset(out + 28, 0, 4);
        if (dense) { memcpy(out + MVHP_MB_HEADER_BYTES, rec + MVHP_MB_HEADER_BYTES, MVHP_MB_COEFS * 2); continue; }
        int16_t *coef = reinterpret_cast<int16_t *>(out + MVHP_MB_HEADER_BYTES);
        for (uint32_t i = 0; i < n; i++) {
            uint32_t e;
            memcpy(&e, rec + MVHP_MB_HEADER_BYTES + 4 * i, 4);
            if ((e & 0xffffu) < (uint32_t)MVHP_MB_COEFS) coef[e & 0xffffu] = (int16_t)(e >> 16);
        }
    }
}
int stub_recon(DevCtx *, const mvhp_stream_params_t *p, const void *d_compact, size_t stride, void *d_packed, int n, uint8_t *d_yuv,
               uint8_t *d_rgb, float *ms, int *layout, int *waves, std::string &)
{
    g_recon_calls++;
    const size_t pb = mvhp_packed_frame_bytes(p), yb = mvhp_yuv_frame_bytes(p), rb = mvhp_rgb_frame_bytes(p);
    for (int i = 0; i < n; i++)
        expand_compact((const uint8_t *)d_compact + (size_t)i * stride, (size_t)p->width_mbs * p->height_mbs, (uint8_t *)d_packed + (size_t)i * pb);
    for (int i = 0; i < n; i++) {
        const uint64_t h = checksum((const uint8_t *)d_packed + (size_t)i * pb, pb);
        memset(d_yuv + (size_t)i * yb, 0x5a, yb);
        memcpy(d_yuv + (size_t)i * yb, &h, sizeof(h));
        if (d_rgb) {
            memset(d_rgb + (size_t)i * rb, 0xa5, rb);
            memcpy(d_rgb + (size_t)i * rb + 8, &h, sizeof(h));
        }
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
    if (ms) *ms = 0.2f;
    if (layout) *layout = MVHP_LAYOUT_ROWS;
    if (waves) *waves = 8;
    return MVHP_SUCCESS;
}

const mvengine::DeviceApi g_stub = {stub_device_count, stub_host_alloc, stub_host_free, stub_ctx_create, stub_ctx_destroy,
                                    stub_dev_alloc, stub_dev_free, stub_dev_free_bytes, stub_copy_n, stub_copy_n, stub_recon,
                                    nullptr, nullptr};   // (no placed arena on the stub device)

struct Check {
    const mvhp_stream *s = nullptr;
    bool want_rgb = false;
    int calls = 0, ok = 0, failed = 0, bad = 0, next_seq = 0;
    int abort_after = -1;
    std::vector<int> order;
    static int sink(void *user, int seq, int idr, int rc, const char *err, const mvhp_stream_params_t *p, const uint8_t *yuv,
                    const uint8_t *rgb)
    {
        Check &c = *static_cast<Check *>(user);
        c.calls++;
        if (seq != c.next_seq || idr != c.order[(size_t)seq]) c.bad++;   // in order, the right picture
        c.next_seq = seq + 1;
        if (rc != MVHP_SUCCESS) {
            c.failed++;
            if (!err || !*err || yuv || rgb) c.bad++;
            return 0;
        }
        std::vector<uint8_t> packed(mvhp_packed_frame_bytes(p));
        std::string e;
        if (c.s->decode_packed(idr, packed.data(), packed.size(), e) != h264::RC_SUCCESS) { c.bad++; return 0; }
        const uint64_t h = checksum(packed.data(), packed.size());
        uint64_t got = 0;
        memcpy(&got, yuv, 8);
        if (got != h || yuv[8] != 0x5a) c.bad++;
        if (c.want_rgb) {
            if (!rgb) c.bad++;
            else { memcpy(&got, rgb + 8, 8); if (got != h || rgb[0] != 0xa5) c.bad++; }
        } else if (rgb) c.bad++;
        c.ok++;
        if (c.abort_after >= 0 && c.calls > c.abort_after) return -1;
        return 1;
    }
};

} // namespace

const mvengine::DeviceApi &mvhp_hip_device_api() { return g_stub; }

#define EXPECT(cond)                                                                          \
    do {                                                                                      \
        if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s <stream.264>\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    for (size_t n; (n = fread(tmp, 1, sizeof(tmp), f)) > 0;) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    mvhp_stream s;
    s.data = buf.data();
    s.size = buf.size();
    std::string err;
    if (s.build(err) != h264::RC_SUCCESS) { fprintf(stderr, "stream: %s\n", err.c_str()); return 2; }
    const int n_idr = (int)s.idrs.size();
    int failures = 0;
    std::vector<int> all(n_idr);
    for (int i = 0; i < n_idr; i++) all[i] = i;

    auto run = [&](const char *name, mvhp_engine_opts_t o, const std::vector<int> &order, int wanted, bool rgb, Check &c,
                   mvhp_decode_stats_t &st) {
        mvhp_engine_t *e = nullptr;
        int rc = mvhp_engine_create(&o, &e);
        if (rc != MVHP_SUCCESS) { fprintf(stderr, "%s: engine_create failed\n", name); failures++; return MVHP_FAILURE; }
        c.s = &s;
        c.want_rgb = rgb;
        c.order = order;
        rc = mvhp_engine_decode(e, &s, order.data(), (int)order.size(), wanted, rgb, Check::sink, &c, &st);
        mvhp_engine_destroy(e);
        printf("%-28s rc=%d issued=%u ok=%u failed=%u batches=%u requeued=%u ctx=%u threads=%u max_batch=%u\n", name, rc,
               st.pictures_issued, st.pictures_ok, st.pictures_failed, st.batches, st.batches_requeued, st.contexts, st.host_threads,
               st.max_batch_pictures);
        return rc;
    };
    mvhp_engine_opts_t base;
    memset(&base, 0, sizeof(base));
    base.fail_context = -1;
    base.host_threads = 4;

    {   // every picture, two contexts, small chunks and batches so that many of each are in flight
        mvhp_engine_opts_t o = base; o.contexts = 2; o.chunk_pictures = 3; o.batch_pictures = 7;
        Check c; mvhp_decode_stats_t st;
        EXPECT(run("all/2ctx", o, all, n_idr, true, c, st) == MVHP_SUCCESS);
        EXPECT(c.bad == 0 && c.ok == n_idr && c.calls == n_idr);
        EXPECT((int)st.pictures_issued == n_idr && st.batches >= (uint32_t)(n_idr / 7));
    }
    {   // one thumbnail out of a long stream: exactly one picture is entropy-decoded (h264.c:173-179)
        mvhp_engine_opts_t o = base; o.contexts = 2;
        Check c; mvhp_decode_stats_t st;
        EXPECT(run("wanted=1", o, all, 1, false, c, st) == MVHP_SUCCESS);
        EXPECT(c.bad == 0 && c.ok == 1 && c.calls == 1 && st.pictures_issued == 1 && st.batches == 1);
    }
    {   // three pictures wanted: no more than three are decoded
        mvhp_engine_opts_t o = base; o.contexts = 3; o.chunk_pictures = 2;
        Check c; mvhp_decode_stats_t st;
        EXPECT(run("wanted=3", o, all, 3, true, c, st) == MVHP_SUCCESS);
        EXPECT(c.bad == 0 && c.ok == 3 && st.pictures_issued == 3);
    }
    {   // a batch fails on context 0: its pictures are decoded again and re-queued to another context, order is kept
        mvhp_engine_opts_t o = base; o.contexts = 3; o.chunk_pictures = 2; o.batch_pictures = 5; o.fail_context = 0;
        Check c; mvhp_decode_stats_t st;
        EXPECT(run("requeue/3ctx", o, all, n_idr, true, c, st) == MVHP_SUCCESS);
        EXPECT(c.bad == 0 && c.ok == n_idr && st.batches_requeued == 1 && (int)st.pictures_issued > n_idr);
    }
    {   // one context: a failed batch has nowhere to go, its pictures are reported as failures, the rest arrive
        mvhp_engine_opts_t o = base; o.contexts = 1; o.chunk_pictures = 2; o.batch_pictures = 5; o.fail_context = 0;
        Check c; mvhp_decode_stats_t st;
        EXPECT(run("fail/1ctx", o, all, n_idr, false, c, st) == MVHP_SUCCESS);
        EXPECT(c.bad == 0 && c.failed == 5 && c.ok == n_idr - 5 && st.batches_requeued == 0);
    }
    {   // ADVICE r2 (feeder race): a re-queued group arrives while the feeder has dropped its lock to page-lock a LARGER chunk
        // than the group holds (batches ramp 64 -> 128 -> 256, the chunk buffers regrow with them, the stub's page-locking
        // takes 30 ms, the first 64-picture batch fails on context 0 meanwhile).  The feeder must finish the chunk it was
        // sizing from the ordinary list and take the group on its next iteration; indexing the group with that chunk's
        // count read past its vector (AddressSanitizer) and delivered pictures twice / never.
        std::vector<int> order;
        for (int k = 0; k < 700; k++) order.push_back(k % n_idr);
        mvhp_engine_opts_t o = base; o.contexts = 2; o.chunk_pictures = 256; o.fail_context = 0;
        Check c; mvhp_decode_stats_t st;
        g_slow_host_alloc_us = 30000;
        EXPECT(run("requeue during page-lock", o, order, (int)order.size(), false, c, st) == MVHP_SUCCESS);
        g_slow_host_alloc_us = 0;
        EXPECT(c.bad == 0 && c.ok == (int)order.size() && c.failed == 0 && st.batches_requeued == 1);
        EXPECT((int)st.pictures_issued > (int)order.size());   // the failed batch's pictures were entropy-decoded twice
    }
    {   // the sink keeps pictures (verdict 2) and two other threads give them back later (minivideo_decode's file writers):
        // the memory must still hold the same picture at release time -- an output chunk that went back to the pool early
        // would have been overwritten by a later download -- and the call must not return before the last release
        std::vector<int> order;
        for (int k = 0; k < 400; k++) order.push_back((k * 7) % n_idr);
        mvhp_engine_opts_t o = base; o.contexts = 2; o.chunk_pictures = 3; o.batch_pictures = 9;
        mvhp_engine_t *e = nullptr;
        EXPECT(mvhp_engine_create(&o, &e) == MVHP_SUCCESS);
        struct Kept { int seq; const uint8_t *yuv; uint64_t h; };
        struct Keeper {
            mvhp_engine_t *e; const mvhp_stream *s;
            std::mutex mu; std::condition_variable cv; std::deque<Kept> q; bool closing = false;
            std::atomic<int> released{0}, bad{0}; int calls = 0;
        } kp;
        kp.e = e; kp.s = &s;
        auto keeper = [&kp] {
            for (;;) {
                Kept k;
                {
                    std::unique_lock<std::mutex> l(kp.mu);
                    kp.cv.wait(l, [&] { return kp.closing || !kp.q.empty(); });
                    if (kp.q.empty()) return;
                    k = kp.q.front(); kp.q.pop_front();
                }
                if (k.seq % 3) std::this_thread::sleep_for(std::chrono::microseconds(300));
                uint64_t got = 0;
                memcpy(&got, k.yuv, 8);
                if (got != k.h || k.yuv[8] != 0x5a) kp.bad++;
                kp.released++;
                mvhp_engine_release_picture(kp.e, k.seq);
            }
        };
        std::thread t1(keeper), t2(keeper);
        auto sink = [](void *u, int seq, int idr, int rc, const char *, const mvhp_stream_params_t *p, const uint8_t *yuv, const uint8_t *) {
            Keeper &x = *static_cast<Keeper *>(u);
            x.calls++;
            if (rc != MVHP_SUCCESS) { x.bad++; return 0; }
            std::vector<uint8_t> packed(mvhp_packed_frame_bytes(p));
            std::string er;
            if (x.s->decode_packed(idr, packed.data(), packed.size(), er) != h264::RC_SUCCESS) { x.bad++; return 0; }
            const uint64_t h = checksum(packed.data(), packed.size());
            if (seq % 5 == 4) return 1;   // (some pictures are not kept: both kinds share chunks)
            { std::lock_guard<std::mutex> l(x.mu); x.q.push_back(Kept{seq, yuv, h}); }
            x.cv.notify_one();
            if (seq % 6 == 0) std::this_thread::sleep_for(std::chrono::microseconds(500));   // released before this returns
            return 2;
        };
        mvhp_decode_stats_t st;
        EXPECT(mvhp_engine_decode(e, &s, order.data(), (int)order.size(), (int)order.size(), 0, sink, &kp, &st) == MVHP_SUCCESS);
        const int released_at_return = kp.released.load();
        { std::lock_guard<std::mutex> l(kp.mu); kp.closing = true; }
        kp.cv.notify_all();
        t1.join(); t2.join();
        mvhp_engine_release_picture(e, 3);   // a stale release is ignored
        mvhp_engine_destroy(e);
        printf("%-28s calls=%d kept=%d bad=%d ok=%u\n", "kept pictures", kp.calls, released_at_return, kp.bad.load(), st.pictures_ok);
        EXPECT(kp.bad == 0 && kp.calls == 400 && released_at_return == 320 && st.pictures_ok == 400);
    }
    {   // the sink stops the decode
        mvhp_engine_opts_t o = base; o.contexts = 2; o.chunk_pictures = 2; o.batch_pictures = 4;
        Check c; c.abort_after = 5; mvhp_decode_stats_t st;
        EXPECT(run("abort", o, all, n_idr, false, c, st) == MVHP_FAILURE);
        EXPECT(c.bad == 0 && c.calls == 6);
    }
    {   // a list with repeats and a reversed order (ORDERED / DISTRIBUTED selections are arbitrary lists)
        std::vector<int> order;
        for (int i = n_idr - 1; i >= 0; i -= 2) order.push_back(i);
        order.push_back(0);
        order.push_back(0);
        mvhp_engine_opts_t o = base; o.contexts = 2; o.chunk_pictures = 4;
        Check c; mvhp_decode_stats_t st;
        EXPECT(run("arbitrary order", o, order, (int)order.size(), true, c, st) == MVHP_SUCCESS);
        EXPECT(c.bad == 0 && c.ok == (int)order.size());
    }
    if (argc > 2) {   // a second stream whose picture argv[3] is broken: the failure arrives in its place
        const int broken = atoi(argv[3]);
        FILE *g = fopen(argv[2], "rb");
        std::vector<uint8_t> b2;
        for (size_t n; g && (n = fread(tmp, 1, sizeof(tmp), g)) > 0;) b2.insert(b2.end(), tmp, tmp + n);
        if (g) fclose(g);
        mvhp_stream s2;
        s2.data = b2.data();
        s2.size = b2.size();
        EXPECT(s2.build(err) == h264::RC_SUCCESS);
        std::vector<int> o2(s2.idrs.size());
        for (size_t i = 0; i < o2.size(); i++) o2[i] = (int)i;
        mvhp_engine_opts_t o = base; o.contexts = 2; o.chunk_pictures = 3; o.batch_pictures = 6;
        mvhp_engine_t *e = nullptr;
        EXPECT(mvhp_engine_create(&o, &e) == MVHP_SUCCESS);
        struct S2 { int broken, bad = 0, ok = 0, failed = 0; } st2{broken};
        auto sink2 = [](void *u, int seq, int, int rc, const char *, const mvhp_stream_params_t *, const uint8_t *, const uint8_t *) {
            S2 &x = *static_cast<S2 *>(u);
            if ((rc == MVHP_SUCCESS) == (seq == x.broken)) x.bad++;
            if (rc == MVHP_SUCCESS) x.ok++; else x.failed++;
            return rc == MVHP_SUCCESS ? 1 : 0;
        };
        mvhp_decode_stats_t st;
        const int wanted = (int)o2.size() - 1;   // the broken picture makes the engine reach one picture further
        EXPECT(mvhp_engine_decode(e, &s2, o2.data(), (int)o2.size(), wanted, 0, sink2, &st2, &st) == MVHP_SUCCESS);
        mvhp_engine_destroy(e);
        printf("%-28s ok=%d failed=%d issued=%u\n", "broken picture", st2.ok, st2.failed, st.pictures_issued);
        EXPECT(st2.bad == 0 && st2.failed == 1 && st2.ok == wanted);
    }
    {   // the PUBLIC API on the stub device: minivideo_open / parse / decode with the file-writer pool (the sink keeps pictures,
        // writer threads write and release them) -- under the sanitizers this is the check of that hand-over.  The stub's
        // "pictures" carry the checksum of their records in the first eight bytes, which is what the files must start with.
        char tmpl[] = "/tmp/mvharness_XXXXXX";
        const char *dir = mkdtemp(tmpl);
        EXPECT(dir != nullptr);
        char cwd[4096];
        EXPECT(getcwd(cwd, sizeof(cwd)) != nullptr);
        std::string in = argv[1];
        if (in[0] != '/') in = std::string(cwd) + "/" + in;
        EXPECT(dir && chdir(dir) == 0);
        for (const char *writers : {"3", "0"}) {
            setenv("MINIVIDEO_WRITERS", writers, 1);
            MediaFile_t *m = nullptr;
            EXPECT(minivideo_open(in.c_str(), &m) == SUCCESS);
            EXPECT(m && minivideo_parse(m, false, true, false) == SUCCESS);
            EXPECT(m && minivideo_decode(m, ".", PICTURE_YUV420, 75, n_idr, PICTURE_UNFILTERED) == SUCCESS);
            int files = 0, good = 0;
            for (int k = 0; k < n_idr; k++) {
                const std::string name = std::string(m->file_name) + "_" + std::to_string(k) + ".yuv";
                FILE *g = fopen(name.c_str(), "rb");
                if (!g) continue;
                files++;
                uint64_t got = 0;
                std::vector<uint8_t> packed;
                mvhp_stream_params_t p;
                if (fread(&got, 1, 8, g) == 8 && mvhp_stream_params(&s, k, &p) == MVHP_SUCCESS) {
                    packed.resize(mvhp_packed_frame_bytes(&p));
                    std::string er;
                    if (s.decode_packed(k, packed.data(), packed.size(), er) == h264::RC_SUCCESS && got == checksum(packed.data(), packed.size())) good++;
                }
                fclose(g);
                remove(name.c_str());
            }
            printf("%-28s writers=%s files=%d of %d, right picture in %d\n", "public API", writers, files, n_idr, good);
            EXPECT(files == n_idr && good == n_idr);
            EXPECT(minivideo_close(&m) == SUCCESS);
        }
        unsetenv("MINIVIDEO_WRITERS");
        EXPECT(chdir(cwd) == 0);
        rmdir(dir);
    }
    EXPECT(g_live_ctx == 0 && g_dev_allocs == 0);
    printf(failures ? "HARNESS FAILED (%d)\n" : "HARNESS OK\n", failures);
    return failures ? 1 : 0;
}
