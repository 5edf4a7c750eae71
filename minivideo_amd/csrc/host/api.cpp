// api.cpp -- MiniVideo's public API (include/minivideo.h) over the split decoder:
//   minivideo_open   <- import.c:510 (import_fileOpen: path split, size, container sniff)
//   minivideo_parse  <- minivideo.c:199-251 + demuxer/esparser/esparser.c:40-143
//   minivideo_decode <- minivideo.c:255-303, demuxer/filter.c:52-215, decoder/h264/h264.c:41-195,
//                       export.c:618-767 (file naming, format fallbacks)
//   minivideo_close  <- import.c:570-616
// Entropy decoding runs on host threads, reconstruction + colour conversion on every visible HIP
// device (frame-level work queue, no collectives); pictures are written in stream order.
#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "export.h"
#include "h264_frontend.h"
#include "minivideo.h"
#include "minivideo_hotpath.h"
#include "mp4_demux.h"
#include "stream_internal.h"

namespace {

void log_err(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void log_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "[minivideo] ");
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
}

// ---- import.c:60-135: absolute path, directory, name (no extension), extension ----
void infos_from_path(MediaFile_t *m)
{
    if (m->file_path[0] != '/') {
        char cwd[4096];
        if (getcwd(cwd, sizeof(cwd)) != NULL) {
            std::string abs = std::string(cwd) + "/" + m->file_path;
            if (abs.size() < sizeof(m->file_path)) {
                FILE *t = fopen(abs.c_str(), "r");
                if (t) { fclose(t); strncpy(m->file_path, abs.c_str(), sizeof(m->file_path) - 1); }
            }
        }
    }
    const char *slash = strrchr(m->file_path, '/');
    if (!slash) return;
    size_t dlen = (size_t)(slash - m->file_path) + 1;
    if (dlen > sizeof(m->file_directory) - 1) dlen = sizeof(m->file_directory) - 1;
    memcpy(m->file_directory, m->file_path, dlen);
    const char *dot = strrchr(m->file_path, '.');
    if (dot && dot > slash) {
        size_t nlen = (size_t)(dot - slash) - 1;
        if (nlen > sizeof(m->file_name) - 1) nlen = sizeof(m->file_name) - 1;
        memcpy(m->file_name, slash + 1, nlen);
        strncpy(m->file_extension, dot + 1, sizeof(m->file_extension) - 1);
    } else {
        strncpy(m->file_name, slash + 1, sizeof(m->file_name) - 1);
    }
}

// ---- import.c:186-470: magic bytes first, then the extension ----
ContainerFormat_e sniff_container(MediaFile_t *m)
{
    uint8_t b[16] = {0};
    rewind(m->file_pointer);
    if (fread(b, 1, sizeof(b), m->file_pointer) < 8) { /* tiny file: fall through to the extension */ }
    rewind(m->file_pointer);
    ContainerFormat_e c = CONTAINER_UNKNOWN;
    if (b[0] == 0x47) c = CONTAINER_MPEG_TS;
    else if (b[0] == 0x1A && b[1] == 0x45 && b[2] == 0xDF && b[3] == 0xA3) c = CONTAINER_MKV;
    else if (b[0] == 'R' && b[1] == 'I' && b[2] == 'F' && b[3] == 'F') {
        if (b[8] == 'A' && b[9] == 'V' && b[10] == 'I' && b[11] == ' ') c = CONTAINER_AVI;
        else if (b[8] == 'W' && b[9] == 'A' && b[10] == 'V' && b[11] == 'E') c = CONTAINER_WAVE;
    } else if (b[0] == 0 && b[1] == 0) {
        if (b[2] == 1) {
            if (b[3] == 0xBA) c = CONTAINER_MPEG_PS;
            else if (b[3] == 0xB3 || b[3] == 0x67) c = CONTAINER_ES;
        } else if (b[2] == 0 && b[3] == 1) {
            if (b[4] == 0xBA) c = CONTAINER_MPEG_PS;
            else if (b[4] == 0xB3 || b[4] == 0x67) c = CONTAINER_ES;
        }
        if (b[4] == 'f' && b[5] == 't' && b[6] == 'y' && b[7] == 'p') c = CONTAINER_MP4;
    }
    if (c == CONTAINER_UNKNOWN) {
        const char *e = m->file_extension;
        if (!strcmp(e, "264") || !strcmp(e, "h264") || !strcmp(e, "265") || !strcmp(e, "h265")) c = CONTAINER_ES;
        else if (!strcmp(e, "mp4") || !strcmp(e, "mov") || !strcmp(e, "m4v") || !strcmp(e, "3gp")) c = CONTAINER_MP4;
        else if (!strcmp(e, "avi")) c = CONTAINER_AVI;
        else if (!strcmp(e, "mkv") || !strcmp(e, "webm")) c = CONTAINER_MKV;
        else if (!strcmp(e, "ts") || !strcmp(e, "m2ts")) c = CONTAINER_MPEG_TS;
        else if (!strcmp(e, "mpg") || !strcmp(e, "mpeg") || !strcmp(e, "vob")) c = CONTAINER_MPEG_PS;
        else if (!strcmp(e, "wav")) c = CONTAINER_WAVE;
        else if (!strcmp(e, "mp3")) c = CONTAINER_ES_MP3;
    }
    return c;
}

void free_map(BitstreamMap_t **pm)
{
    if (!pm || !*pm) return;
    BitstreamMap_t *m = *pm;
    free(m->stream_encoder); free(m->track_title); free(m->track_languagecode); free(m->subtitles_name);
    free(m->sample_type); free(m->sample_size); free(m->sample_offset); free(m->sample_pts); free(m->sample_dts);
    free(m);
    *pm = NULL;
}

bool read_whole_file(MediaFile_t *m, std::vector<uint8_t> &buf)
{
    if (!m->file_pointer || m->file_size <= 0) return false;
    buf.resize((size_t)m->file_size);
    rewind(m->file_pointer);
    const size_t n = fread(buf.data(), 1, buf.size(), m->file_pointer);
    rewind(m->file_pointer);
    return n == buf.size();
}

// ---- MP4: public sample map as convertTrack builds it (demuxer/mp4/mp4.c:150-500): the avcC parameter sets first,
//      then every sample of the video track, sync samples marked ----
int parse_mp4_track(MediaFile_t *m, const std::vector<uint8_t> &buf)
{
    mp4::VideoTrack trk;
    std::string err;
    if (!mp4::parse(buf.data(), buf.size(), trk, err)) { log_err("%s", err.c_str()); return FAILURE; }
    free_map(&m->tracks_video[0]);
    BitstreamMap_t *map = (BitstreamMap_t *)calloc(1, sizeof(BitstreamMap_t));
    if (!map) return FAILURE;
    const size_t n = trk.sps.size() + trk.pps.size() + trk.samples.size();
    map->sample_type = (uint32_t *)calloc(n, sizeof(uint32_t));
    map->sample_size = (uint32_t *)calloc(n, sizeof(uint32_t));
    map->sample_offset = (int64_t *)calloc(n, sizeof(int64_t));
    map->sample_pts = (int64_t *)calloc(n, sizeof(int64_t));
    map->sample_dts = (int64_t *)calloc(n, sizeof(int64_t));
    if (!map->sample_type || !map->sample_size || !map->sample_offset || !map->sample_pts || !map->sample_dts) {
        free_map(&map);
        return FAILURE;
    }
    map->stream_type = stream_VIDEO;
    map->stream_codec = CODEC_H264;
    map->stream_fcc = 0x61766331u; // 'avc1'
    map->sample_alignment = true;
    map->width = trk.width;
    map->height = trk.height;
    if (trk.timescale) map->duration_ms = (unsigned int)((double)trk.duration / (double)trk.timescale * 1000.0);
    size_t k = 0;
    for (const mp4::NalRef &r : trk.sps) { map->sample_type[k] = sample_VIDEO_PARAM; map->sample_offset[k] = (int64_t)r.offset; map->sample_size[k] = (uint32_t)r.size; map->sample_pts[k] = map->sample_dts[k] = -1; k++; }
    for (const mp4::NalRef &r : trk.pps) { map->sample_type[k] = sample_VIDEO_PARAM; map->sample_offset[k] = (int64_t)r.offset; map->sample_size[k] = (uint32_t)r.size; map->sample_pts[k] = map->sample_dts[k] = -1; k++; }
    for (const mp4::Sample &sm : trk.samples) {
        map->sample_type[k] = sm.sync ? sample_VIDEO_SYNC : sample_VIDEO;
        map->sample_offset[k] = (int64_t)sm.offset;
        map->sample_size[k] = (uint32_t)sm.size;
        map->stream_size += sm.size;
        if (sm.sync) map->frame_count_idr++;
        k++;
    }
    map->sample_count = (uint32_t)n;
    map->frame_count = (uint32_t)trk.samples.size();
    m->tracks_video[0] = map;
    m->tracks_video_count = 1;
    return SUCCESS;
}

// ---- demuxer/filter.c:52-215: which IDR samples to decode, in which order ----
std::vector<int> select_idrs(const mvhp_stream &s, int picture_number, int mode)
{
    const int n_idr = (int)s.idrs.size();
    std::vector<int> sel;
    if (n_idr == 0 || picture_number <= 0) return sel;
    if (picture_number > n_idr) picture_number = n_idr;
    if (mode == PICTURE_UNFILTERED) {
        for (int i = 0; i < n_idr; i++) sel.push_back(i); // decode in order until picture_number succeed (h264.c:173)
        return sel;
    }
    // ORDERED / DISTRIBUTED: keep pictures larger than (mean IDR sample size)/1.66, drop 3 % at both ends
    // when there are more than 48 of them (filter.c:100-122)
    double payload = 0;
    for (int i = 0; i < n_idr; i++) payload += (double)s.samples[s.idrs[i].sample].sample_size;
    const int threshold = (int)((payload / n_idr) / 1.66);
    const int borders = n_idr > 48 ? (int)ceil(n_idr * 0.03) : 0;
    std::vector<int> cand;
    for (int i = borders; i < n_idr - borders; i++)
        if ((int)s.samples[s.idrs[i].sample].sample_size > threshold) cand.push_back(i);
    const int T = (int)cand.size();
    if (T == 0) return sel;
    if (picture_number > T) picture_number = T;
    // filter.c:140 divides by (picture_number - 1) and crashes for one picture; one picture = first candidate
    const int jump = picture_number > 1 ? T / (picture_number - 1) : 0;
    for (int i = 0; i < picture_number; i++) {
        int k = (mode == PICTURE_ORDERED) ? i : i * jump;
        if (k > T - 1) k = T - 1; // the reference indexes past its candidate list here (filter.c:170-175)
        sel.push_back(cand[k]);
    }
    return sel;
}

// ---- frame-level work queue over host threads and HIP devices ----
// A Window is a run of consecutive pictures processed together: their packed records, planes and RGB live in
// three page-locked buffers (recycled between windows), so entropy threads write records in place, each device
// DMA-reads its share and DMA-writes results in place, and the writer streams files straight from the buffer.
struct Picture {
    int idr = -1;
    int rc = h264::RC_FAILURE;
    std::string err;
    mvhp_stream_params_t params{};
    size_t packed_off = 0, yuv_off = 0, rgb_off = 0;
};

struct Window {
    std::vector<Picture> pics;
    uint8_t *packed = nullptr, *yuv = nullptr, *rgb = nullptr;
    size_t packed_cap = 0, yuv_cap = 0, rgb_cap = 0;
    ~Window()
    {
        mvhp_host_free(packed);
        mvhp_host_free(yuv);
        mvhp_host_free(rgb);
    }
    bool reserve(size_t pb, size_t yb, size_t rb)
    {
        auto grow = [](uint8_t *&p, size_t &cap, size_t need) {
            if (need <= cap) return true;
            mvhp_host_free(p);
            p = (uint8_t *)mvhp_host_alloc(need);
            cap = p ? need : 0;
            return p != nullptr;
        };
        return grow(packed, packed_cap, pb) && grow(yuv, yuv_cap, yb) && (rb == 0 || grow(rgb, rgb_cap, rb));
    }
};

template <class F> void parallel_for(int n, int threads, F f)
{
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    std::atomic<int> next(0);
    std::vector<std::thread> th;
    auto body = [&]() { for (int i; (i = next.fetch_add(1)) < n;) f(i); };
    for (int t = 1; t < threads; t++) th.emplace_back(body);
    body();
    for (auto &t : th) t.join();
}

bool same_params(const mvhp_stream_params_t &a, const mvhp_stream_params_t &b)
{
    return a.width_mbs == b.width_mbs && a.height_mbs == b.height_mbs &&
           a.chroma_qp_index_offset == b.chroma_qp_index_offset &&
           a.second_chroma_qp_index_offset == b.second_chroma_qp_index_offset;
}

class Pipeline {
public:
    Pipeline(const mvhp_stream &s, std::vector<int> order, bool want_rgb)
        : s_(s), order_(std::move(order)), want_rgb_(want_rgb)
    {
        n_gpus_ = mvhp_device_count();
        if (const char *e = getenv("MINIVIDEO_GPUS")) { const int v = atoi(e); if (v > 0 && v < n_gpus_) n_gpus_ = v; }
        host_threads_ = (int)std::thread::hardware_concurrency();
        if (const char *e = getenv("MINIVIDEO_HOST_THREADS")) { const int v = atoi(e); if (v > 0) host_threads_ = v; }
        if (host_threads_ < 1) host_threads_ = 1;
        if (host_threads_ > 64) host_threads_ = 64;
    }
    ~Pipeline() { stop(); for (auto *c : ctx_) mvhp_destroy(c); }

    bool start(std::string &err)
    {
        if (n_gpus_ <= 0) { err = "no HIP device available: this build has no CPU reconstruction path"; return false; }
        for (int d = 0; d < n_gpus_; d++) {
            mvhp_ctx_t *c = nullptr;
            if (mvhp_create(d, &c) != MVHP_SUCCESS) { err = mvhp_last_error(); return false; }
            ctx_.push_back(c);
        }
        for (int i = 0; i < 3; i++) free_.push_back(std::make_unique<Window>());
        entropy_thread_ = std::thread([this] { entropy_stage(); });
        gpu_thread_ = std::thread([this] { gpu_stage(); });
        return true;
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> l(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (entropy_thread_.joinable()) entropy_thread_.join();
        if (gpu_thread_.joinable()) gpu_thread_.join();
    }
    // next finished window in stream order, or nullptr at the end
    std::unique_ptr<Window> next()
    {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [this] { return !done_.empty() || gpu_finished_; });
        if (done_.empty()) return nullptr;
        auto w = std::move(done_.front());
        done_.pop_front();
        return w;
    }
    // hand a consumed window's buffers back
    void recycle(std::unique_ptr<Window> w)
    {
        {
            std::lock_guard<std::mutex> l(mu_);
            free_.push_back(std::move(w));
        }
        cv_.notify_all();
    }

private:
    int window_frames(const mvhp_stream_params_t &p) const
    {
        const size_t pb = mvhp_packed_frame_bytes(&p);
        long f = (long)((size_t)256 << 20) / (long)(pb ? pb : 1);
        if (f < 2 * n_gpus_) f = 2 * n_gpus_;
        if (f < 4) f = 4;
        if (f > 256) f = 256;
        return (int)f;
    }
    void entropy_stage()
    {
        size_t pos = 0;
        while (pos < order_.size()) {
            std::unique_ptr<Window> w;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [this] { return stop_ || !free_.empty(); });
                if (stop_) break;
                w = std::move(free_.front());
                free_.pop_front();
            }
            mvhp_stream_params_t p0{};
            int wf = 16;
            for (size_t k = pos; k < order_.size(); k++)
                if (mvhp_stream_params(&s_, order_[k], &p0) == MVHP_SUCCESS) { wf = window_frames(p0); break; }
            const size_t end = std::min(order_.size(), pos + (size_t)wf);
            w->pics.assign(end - pos, Picture());
            size_t pb = 0, yb = 0, rb = 0;
            for (size_t k = pos; k < end; k++) {
                Picture &pic = w->pics[k - pos];
                pic.idr = order_[k];
                if (mvhp_stream_params(&s_, pic.idr, &pic.params) != MVHP_SUCCESS) {
                    pic.rc = h264::RC_FAILURE;
                    pic.err = "parameter sets missing";
                    continue;
                }
                pic.rc = h264::RC_UNSUPPORTED; // "not parsed yet"
                pic.packed_off = pb; pic.yuv_off = yb; pic.rgb_off = rb;
                pb += mvhp_packed_frame_bytes(&pic.params);
                yb += mvhp_yuv_frame_bytes(&pic.params);
                if (want_rgb_) rb += mvhp_rgb_frame_bytes(&pic.params);
            }
            if (!w->reserve(pb, yb, rb)) {
                for (Picture &pic : w->pics) { pic.rc = h264::RC_FAILURE; pic.err = "out of page-locked host memory"; }
            } else {
                parallel_for((int)w->pics.size(), host_threads_, [&](int i) {
                    Picture &pic = w->pics[i];
                    if (pic.rc != h264::RC_UNSUPPORTED) return;
                    pic.rc = s_.decode_packed(pic.idr, w->packed + pic.packed_off, mvhp_packed_frame_bytes(&pic.params), pic.err);
                });
            }
            pos = end;
            {
                std::lock_guard<std::mutex> l(mu_);
                parsed_.push_back(std::move(w));
            }
            cv_.notify_all();
        }
        {
            std::lock_guard<std::mutex> l(mu_);
            entropy_finished_ = true;
        }
        cv_.notify_all();
    }
    void gpu_stage()
    {
        for (;;) {
            std::unique_ptr<Window> w;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [this] { return stop_ || !parsed_.empty() || entropy_finished_; });
                if (stop_ || parsed_.empty()) break;
                w = std::move(parsed_.front());
                parsed_.pop_front();
            }
            // maximal runs of consecutive parsed pictures with identical stream parameters are contiguous in the
            // window buffers; each run is dealt out to the devices in contiguous shares
            const int n_pics = (int)w->pics.size();
            int a = 0;
            while (a < n_pics) {
                if (w->pics[a].rc != h264::RC_SUCCESS) { a++; continue; }
                int b = a + 1;
                while (b < n_pics && w->pics[b].rc == h264::RC_SUCCESS && same_params(w->pics[a].params, w->pics[b].params)) b++;
                const int n = b - a;
                const int parts = std::min(n, n_gpus_);
                std::atomic<int> next_part(0);
                auto work = [&](int dev) {
                    for (int part; (part = next_part.fetch_add(1)) < parts;) {
                        const int lo = a + (int)((long)n * part / parts), hi = a + (int)((long)n * (part + 1) / parts);
                        const Picture &first = w->pics[lo];
                        const int rc = mvhp_recon_batch_host(ctx_[dev], &first.params, w->packed + first.packed_off, hi - lo,
                                                             w->yuv + first.yuv_off, want_rgb_ ? w->rgb + first.rgb_off : nullptr);
                        if (rc != MVHP_SUCCESS)
                            for (int i = lo; i < hi; i++) { w->pics[i].rc = h264::RC_FAILURE; w->pics[i].err = mvhp_last_error(); }
                    }
                };
                std::vector<std::thread> th;
                for (int d = 1; d < parts; d++) th.emplace_back(work, d);
                work(0);
                for (auto &t : th) t.join();
                a = b;
            }
            {
                std::lock_guard<std::mutex> l(mu_);
                done_.push_back(std::move(w));
            }
            cv_.notify_all();
        }
        {
            std::lock_guard<std::mutex> l(mu_);
            gpu_finished_ = true;
        }
        cv_.notify_all();
    }

    const mvhp_stream &s_;
    std::vector<int> order_;
    bool want_rgb_;
    int n_gpus_ = 0, host_threads_ = 1;
    std::vector<mvhp_ctx_t *> ctx_;
    std::thread entropy_thread_, gpu_thread_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::unique_ptr<Window>> free_, parsed_, done_;
    bool stop_ = false, entropy_finished_ = false, gpu_finished_ = false;
};

} // namespace

extern "C" {

minivideo_EXPORT void minivideo_print_infos(void)
{
    printf("\nminivideo_print_infos()\n");
    printf("* Library version %d.%d-%d (MI355X-native H.264 intra path, gfx950)\n", minivideo_VERSION_MAJOR,
           minivideo_VERSION_MINOR, minivideo_VERSION_PATCH);
    printf("* HIP devices visible: %d\n", mvhp_device_count());
}

minivideo_EXPORT void minivideo_get_infos(int *major, int *minor, int *patch, const char **builddate, const char **buildtime)
{
    if (major) *major = minivideo_VERSION_MAJOR;
    if (minor) *minor = minivideo_VERSION_MINOR;
    if (patch) *patch = minivideo_VERSION_PATCH;
    if (builddate) *builddate = __DATE__;
    if (buildtime) *buildtime = __TIME__;
}

minivideo_EXPORT int minivideo_endianness(void)
{
    const uint32_t i = 0x01020304u;
    const uint8_t *p = (const uint8_t *)&i;
    return (p[0] == 0x04) ? 1234 : 4321;
}

minivideo_EXPORT int minivideo_open(const char *input_filepath, MediaFile_t **input_media)
{
    if (!input_filepath || !input_media) return FAILURE;
    MediaFile_t *m = (MediaFile_t *)calloc(1, sizeof(MediaFile_t));
    if (!m) return FAILURE;
    strncpy(m->file_path, input_filepath, sizeof(m->file_path) - 1);
    m->file_pointer = fopen(input_filepath, "r");
    if (!m->file_pointer) {
        log_err("Unable to open the media file: '%s'", input_filepath);
        free(m);
        *input_media = NULL;
        return FAILURE;
    }
    infos_from_path(m);
    fseek(m->file_pointer, 0, SEEK_END);
    m->file_size = (int64_t)ftell(m->file_pointer);
    rewind(m->file_pointer);
    m->container = sniff_container(m);
    *input_media = m;
    return SUCCESS;
}

minivideo_EXPORT int minivideo_parse(MediaFile_t *m, const bool extract_audio, const bool extract_video,
                                     const bool extract_subtitles)
{
    (void)extract_audio; (void)extract_video; (void)extract_subtitles;
    if (!m) { log_err("Unable to parse NULL MediaFile_t struct!"); return FAILURE; }
    if (m->file_size == 0) { log_err("Unable to parse emtpy file!"); return FAILURE; }
    if (m->container != CONTAINER_ES && m->container != CONTAINER_MP4) {
        log_err("Unable to parse container format '%s': only H.264 elementary streams and MP4/MOV are handled by this build",
                getContainerString(m->container, false));
        return FAILURE;
    }
    std::vector<uint8_t> buf;
    if (!read_whole_file(m, buf)) { log_err("Unable to read the media file"); return FAILURE; }
    if (m->container == CONTAINER_MP4) return parse_mp4_track(m, buf);
    std::vector<h264::EsSample> samples;
    if (h264::index_annexb(buf.data(), buf.size(), samples) != h264::RC_SUCCESS) {
        log_err("No NAL Unit have been found in this bitstream!");
        return FAILURE;
    }
    free_map(&m->tracks_video[0]);
    BitstreamMap_t *map = (BitstreamMap_t *)calloc(1, sizeof(BitstreamMap_t));
    if (!map) return FAILURE;
    const size_t n = samples.size();
    map->sample_type = (uint32_t *)calloc(n, sizeof(uint32_t));
    map->sample_size = (uint32_t *)calloc(n, sizeof(uint32_t));
    map->sample_offset = (int64_t *)calloc(n, sizeof(int64_t));
    map->sample_pts = (int64_t *)calloc(n, sizeof(int64_t));
    map->sample_dts = (int64_t *)calloc(n, sizeof(int64_t));
    if (!map->sample_type || !map->sample_size || !map->sample_offset || !map->sample_pts || !map->sample_dts) {
        free_map(&map);
        return FAILURE;
    }
    map->stream_type = stream_VIDEO;
    map->stream_codec = CODEC_H264;
    map->sample_alignment = true;
    map->stream_intracoded = false;
    for (size_t i = 0; i < n; i++) {
        map->sample_type[i] = samples[i].is_idr ? sample_VIDEO_SYNC : sample_VIDEO_PARAM;
        map->sample_size[i] = (uint32_t)samples[i].sample_size;
        map->sample_offset[i] = (int64_t)samples[i].offset;
        map->sample_pts[i] = -1;
        map->stream_size += samples[i].sample_size;
        if (samples[i].is_idr) map->frame_count_idr++;
    }
    map->sample_count = (uint32_t)n;
    map->frame_count = map->frame_count_idr;
    m->tracks_video[0] = map;
    m->tracks_video_count = 1;
    return SUCCESS;
}

minivideo_EXPORT int minivideo_decode(MediaFile_t *m, const char *output_directory, const int picture_format,
                                      const int picture_quality, const int picture_number,
                                      const int picture_extractionmode)
{
    (void)output_directory; // accepted and ignored, like the reference (h264.c:65)
    (void)picture_quality;
    if (!m) { log_err("Unable to start decoding because of an empty MediaFile_t structure! Parsing failed?"); return FAILURE; }
    BitstreamMap_t *map = m->tracks_video[0];
    if (!map || map->stream_type != stream_VIDEO) { log_err("No video track to decode"); return FAILURE; }
    if (map->stream_codec != CODEC_H264) { log_err("Unable to decode given file format: no decoder available!"); return FAILURE; }

    std::vector<uint8_t> buf;
    if (!read_whole_file(m, buf)) { log_err("Unable to read the media file"); return FAILURE; }
    mvhp_stream s;
    s.data = buf.data();
    s.size = buf.size();
    std::string err;
    const int brc = (m->container == CONTAINER_MP4) ? s.build_mp4(err) : s.build(err);
    if (brc != h264::RC_SUCCESS) { log_err("%s", err.c_str()); return FAILURE; }

    // idr_filtering (filter.c:52-92)
    int wanted = picture_number;
    if ((int)s.idrs.size() < wanted) wanted = (int)s.idrs.size();
    if (wanted <= 0) { log_err("No picture to decode after filtering!"); return FAILURE; }
    std::vector<int> order = select_idrs(s, wanted, picture_extractionmode);
    if (order.empty()) { log_err("No picture to decode after filtering!"); return FAILURE; }
    if (picture_extractionmode != PICTURE_UNFILTERED && (int)order.size() < wanted) wanted = (int)order.size();

    // export.c:644-690: format fallbacks of a build with stb_image_write only (ENABLE_JPEG = ENABLE_PNG = 0)
    int fmt = picture_format;
    if (fmt == PICTURE_JPG) fmt = PICTURE_PNG;
    const char *ext = "yuv";
    if (fmt == PICTURE_PNG) ext = "png";
    else if (fmt == PICTURE_BMP) ext = "bmp";
    else if (fmt == PICTURE_TGA) ext = "tga";
    const bool want_rgb = (fmt == PICTURE_PNG || fmt == PICTURE_BMP || fmt == PICTURE_TGA);

    Pipeline pipe(s, order, want_rgb);
    if (!pipe.start(err)) { log_err("%s", err.c_str()); return FAILURE; }

    int exported = 0, errors = 0, retcode = FAILURE;
    bool running = true;
    while (running) {
        std::unique_ptr<Window> w = pipe.next();
        if (!w) break;
        for (Picture &pic : w->pics) {
            if (pic.rc != h264::RC_SUCCESS) {
                log_err("IDR %d: %s", pic.idr, pic.err.c_str());
                if (++errors > 64) { running = false; retcode = FAILURE; break; } // h264.c:181-187
                continue;
            }
            errors = 0;
            // export.c:627-642, 704-708: <file_name>[_k].<ext> in the current working directory
            std::string name = m->file_name;
            if (picture_number > 1) name += "_" + std::to_string(exported);
            name += ".";
            name += ext;
            const int W = (int)pic.params.width_mbs * 16, H = (int)pic.params.height_mbs * 16;
            int ok = 0;
            const uint8_t *yuv = w->yuv + pic.yuv_off, *rgb = want_rgb ? w->rgb + pic.rgb_off : nullptr;
            if (fmt == PICTURE_PNG) ok = mvexport::write_png(name, rgb, W, H);
            else if (fmt == PICTURE_BMP) ok = mvexport::write_bmp(name, rgb, W, H);
            else if (fmt == PICTURE_TGA) ok = mvexport::write_tga(name, rgb, W, H);
            else if (fmt == PICTURE_YUV444) ok = mvexport::write_yuv444(name, yuv, W, H);
            else ok = mvexport::write_yuv420(name, yuv, W, H);
            if (!ok) { log_err("Unable to write '%s'", name.c_str()); errors++; continue; }
            exported++;
            if (exported == wanted) { retcode = SUCCESS; running = false; break; } // h264.c:173-179
        }
        if (running) pipe.recycle(std::move(w));
    }
    pipe.stop();
    if (retcode != SUCCESS && exported > 0 && errors <= 64) retcode = SUCCESS; // stream ended after the last good IDR
    return retcode;
}

minivideo_EXPORT int minivideo_extract(MediaFile_t *m, const char *output_directory, const bool extract_audio,
                                       const bool extract_video, const bool extract_subtitles, const int output_format)
{
    (void)m; (void)output_directory; (void)extract_audio; (void)extract_video; (void)extract_subtitles; (void)output_format;
    log_err("minivideo_extract: elementary-stream re-export is not part of this build (muxer/ is out of scope)");
    return FAILURE;
}

minivideo_EXPORT int minivideo_close(MediaFile_t **pm)
{
    int retcode = SUCCESS;
    if (pm && *pm) {
        MediaFile_t *m = *pm;
        if (m->file_pointer && fclose(m->file_pointer) != 0) retcode = FAILURE;
        for (int i = 0; i < 16; i++) { free_map(&m->tracks_audio[i]); free_map(&m->tracks_video[i]); free_map(&m->tracks_subt[i]); }
        free(m);
        *pm = NULL;
    }
    return retcode;
}

minivideo_EXPORT const char *getContainerString(ContainerFormat_e c, bool long_description)
{
    switch (c) {
    case CONTAINER_AVI: return long_description ? "AVI 'Audio Video Interleave'" : "AVI";
    case CONTAINER_MKV: return long_description ? "Matroska" : "MKV";
    case CONTAINER_MP4: return long_description ? "ISO Base Media format (MOV, MP4, ...)" : "MP4";
    case CONTAINER_MPEG_PS: return long_description ? "MPEG 'Program Stream'" : "MPEG-PS";
    case CONTAINER_MPEG_TS: return long_description ? "MPEG 'Transport Stream'" : "MPEG-TS";
    case CONTAINER_WAVE: return long_description ? "WAVE 'Waveform Audio File Format'" : "WAVE";
    case CONTAINER_ES: return long_description ? "Undefined 'Elementary Stream'" : "ES";
    case CONTAINER_ES_MP3: return long_description ? "MP3 'Elementary Stream'" : "MP3 ES";
    default: return long_description ? "Unknown container format" : "UNKNOWN";
    }
}

minivideo_EXPORT const char *getCodecString(StreamType_e type, AVCodec_e codec, bool long_description)
{
    (void)type;
    if (codec == CODEC_H264) return long_description ? "H.264 (MPEG-4 Part 10 'Advanced Video Coding')" : "H.264";
    return long_description ? "Unknown codec" : "UNKNOWN";
}

minivideo_EXPORT const char *getPictureString(PictureFormat_e p, bool long_description)
{
    (void)long_description;
    switch (p) {
    case PICTURE_BMP: return "BMP";
    case PICTURE_JPG: return "JPG";
    case PICTURE_PNG: return "PNG";
    case PICTURE_WEBP: return "WebP";
    case PICTURE_TGA: return "TGA";
    case PICTURE_YUV444: return "YCbCr 4:4:4";
    case PICTURE_YUV420: return "YCbCr 4:2:0";
    default: return "UNKNOWN";
    }
}

minivideo_EXPORT AVCodec_e getCodecFromFourCC(const uint32_t fcc)
{
    // big-endian FourCC words as the reference builds them (fourcc.h): 'avc1', 'AVC1', 'h264', 'H264', 'x264', 'X264'
    switch (fcc) {
    case 0x61766331u: case 0x41564331u: case 0x68323634u: case 0x48323634u: case 0x78323634u: case 0x58323634u:
        return CODEC_H264;
    default: return CODEC_UNKNOWN;
    }
}

} // extern "C"
