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
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "decode_engine.h"
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

double wall_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The whole file in memory.  (Not mapped: a file that shrinks under a mapping kills the process; a short read is an error
// return.)  minivideo_parse leaves the bytes of the file it parsed last in a one-entry cache, so that the minivideo_decode
// that follows does not read 200 MB a second time (0.06 s of a 1.2-s mini_thumbnailer run); minivideo_close drops it.
struct FileBytes {
    std::unique_ptr<uint8_t[]> p;   // (not a vector: no zero fill of what fread overwrites)
    size_t n = 0;
    const uint8_t *data() const { return p.get(); }
    size_t size() const { return n; }
};

std::mutex g_last_parsed_mu;
const MediaFile_t *g_last_parsed_owner = nullptr;
std::shared_ptr<FileBytes> g_last_parsed;

std::shared_ptr<FileBytes> read_whole_file(MediaFile_t *m)
{
    if (!m->file_pointer || m->file_size <= 0) return nullptr;
    auto buf = std::make_shared<FileBytes>();
    buf->n = (size_t)m->file_size;
    buf->p.reset(new (std::nothrow) uint8_t[buf->n]);
    if (!buf->p) return nullptr;
    rewind(m->file_pointer);
    const size_t got = fread(buf->p.get(), 1, buf->n, m->file_pointer);
    rewind(m->file_pointer);
    return got == buf->n ? buf : nullptr;
}

void remember_parsed(const MediaFile_t *m, std::shared_ptr<FileBytes> buf)
{
    std::lock_guard<std::mutex> l(g_last_parsed_mu);
    g_last_parsed_owner = m;
    g_last_parsed = std::move(buf);
}

std::shared_ptr<FileBytes> take_parsed(const MediaFile_t *m, bool keep)
{
    std::lock_guard<std::mutex> l(g_last_parsed_mu);
    if (g_last_parsed_owner != m) return nullptr;
    std::shared_ptr<FileBytes> r = g_last_parsed;
    if (!keep) { g_last_parsed_owner = nullptr; g_last_parsed.reset(); }
    return r;
}

// ---- MP4: public sample map as convertTrack builds it (demuxer/mp4/mp4.c:150-500): the avcC parameter sets first,
//      then every sample of the video track, sync samples marked ----
int parse_mp4_track(MediaFile_t *m, const FileBytes &buf)
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

// ---- the sink of minivideo_decode: export.c:618-767 (file naming, format fallbacks, writers) ----
// The pipeline itself (entropy threads -> H2D -> batched kernels -> D2H, frame-level work queue over every HIP
// device) is the decode engine, csrc/host/decode_engine.cpp; it calls this once per picture, in stream order.
struct ExportSink {
    const MediaFile_t *m = nullptr;
    int fmt = PICTURE_YUV420;
    const char *ext = "yuv";
    bool want_rgb = false;
    int picture_number = 1;
    int exported = 0, errors = 0;   // exported: pictures named so far (the _k of the file name)
    bool aborted = false;

    // File writers (round 3): one picture's write to the page cache is a 3-6 MB copy, 0.5 ms -- on the calling thread that
    // capped minivideo_decode at 2000 pictures/s while the pipeline delivers 3300.  The sink keeps the picture (verdict 2),
    // a small pool writes it and gives it back.  MINIVIDEO_WRITERS=0: write on the calling thread as before.
    struct Job { int seq; std::string name; int W, H; const uint8_t *yuv, *rgb; };
    mvhp_engine_t *eng = nullptr;
    std::vector<std::thread> pool;
    std::mutex mu;
    std::condition_variable cv_job, cv_room;
    std::deque<Job> jobs;
    size_t max_jobs = 8;            // pictures queued or being written: bounds what is kept out of the engine's chunks
    size_t in_flight = 0;
    int n_writers = 0;
    bool closing = false;
    std::atomic<int> written{0}, write_errors{0};
    bool stopped_by_write_error = false;

    int write_one(const std::string &name, int W, int H, const uint8_t *yuv, const uint8_t *rgb) const
    {
        if (fmt == PICTURE_PNG) return mvexport::write_png(name, rgb, W, H);
        if (fmt == PICTURE_BMP) return mvexport::write_bmp(name, rgb, W, H);
        if (fmt == PICTURE_TGA) return mvexport::write_tga(name, rgb, W, H);
        if (fmt == PICTURE_YUV444) return mvexport::write_yuv444(name, yuv, W, H);
        return mvexport::write_yuv420(name, yuv, W, H);
    }
    void start(int n, mvhp_engine_t *e)
    {
        eng = e;
        n_writers = n;
        max_jobs = (size_t)n + 4;
        for (int i = 0; i < n; i++) pool.emplace_back([this] { writer(); });
    }
    void finish()
    {
        {
            std::lock_guard<std::mutex> l(mu);
            closing = true;
        }
        cv_job.notify_all();
        for (auto &t : pool) t.join();
        pool.clear();
    }
    void writer()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> l(mu);
                cv_job.wait(l, [&] { return closing || !jobs.empty(); });
                if (jobs.empty()) return;
                j = std::move(jobs.front());
                jobs.pop_front();
            }
            if (write_one(j.name, j.W, j.H, j.yuv, j.rgb)) written++;
            else { log_err("Unable to write '%s'", j.name.c_str()); write_errors++; }
            mvhp_engine_release_picture(eng, j.seq);
            {
                std::lock_guard<std::mutex> l(mu);
                in_flight--;
            }
            cv_room.notify_one();
        }
    }

    static int call(void *user, int seq, int idr, int rc, const char *err, const mvhp_stream_params_t *p, const uint8_t *yuv,
                    const uint8_t *rgb)
    {
        ExportSink &x = *static_cast<ExportSink *>(user);
        if (rc != MVHP_SUCCESS) {
            log_err("IDR %d: %s", idr, err ? err : "failed");
            if (++x.errors > 64) { x.aborted = true; return -1; }   // h264.c:181-187
            return 0;
        }
        x.errors = 0;
        // export.c:627-642, 704-708: <file_name>[_k].<ext> in the current working directory
        std::string name = x.m->file_name;
        if (x.picture_number > 1) name += "_" + std::to_string(x.exported);
        name += ".";
        name += x.ext;
        const int W = (int)p->width_mbs * 16, H = (int)p->height_mbs * 16;
        if (!x.pool.empty()) {
            // Writers run behind the pipeline, so a write that fails is only known later: no further picture is decoded in
            // its place (the synchronous path below does that).  Once one has failed -- a full disk does not get better --
            // decoding stops, and minivideo_decode answers FAILURE unless every wanted file was written (contract: minivideo.h).
            if (x.write_errors.load() > 0) { x.stopped_by_write_error = true; return -1; }
            {
                std::unique_lock<std::mutex> l(x.mu);
                x.cv_room.wait(l, [&] { return x.in_flight < x.max_jobs; });
                x.in_flight++;
                x.jobs.push_back(Job{seq, std::move(name), W, H, yuv, rgb});
            }
            x.cv_job.notify_one();
            x.exported++;
            return 2;
        }
        if (!x.write_one(name, W, H, yuv, rgb)) {
            log_err("Unable to write '%s'", name.c_str());
            x.errors++;
            return 0;
        }
        x.exported++;
        x.written++;
        return 1;
    }
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
    const double t_call = wall_s();
    std::shared_ptr<FileBytes> file = read_whole_file(m);
    if (!file) { log_err("Unable to read the media file"); return FAILURE; }
    const FileBytes &buf = *file;
    remember_parsed(m, file);
    const double t_read = wall_s();
    if (m->container == CONTAINER_MP4) return parse_mp4_track(m, buf);
    std::vector<h264::EsSample> samples;
    const bool spec = getenv("MINIVIDEO_SPEC") && atoi(getenv("MINIVIDEO_SPEC")) != 0;   // opt-in, SURVEY 8f row f4
    if ((spec ? h264::index_annexb_spec(buf.data(), buf.size(), samples) : h264::index_annexb(buf.data(), buf.size(), samples)) != h264::RC_SUCCESS) {
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
    if (getenv("MINIVIDEO_STATS"))
        fprintf(stderr, "[minivideo] parse call: reading the file %.3f s, indexing + sample map %.3f s\n", t_read - t_call, wall_s() - t_read);
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

    // The engine (HIP runtime, one context per device, its thread pools: 0.2-0.5 s in a fresh process) comes up on a thread
    // of its own while this one reads and indexes the file.
    const double t_call = wall_s();
    struct EngineStart {
        mvhp_engine_t *eng = nullptr;
        int rc = MVHP_FAILURE;
        double seconds = 0;
        std::thread th;
        bool joined = false;
        void join() { if (!joined) { th.join(); joined = true; } }
        ~EngineStart()
        {
            join();
            const double t = wall_s();
            if (eng) mvhp_engine_destroy(eng);
            if (getenv("MINIVIDEO_STATS")) fprintf(stderr, "[minivideo] engine torn down in %.3f s\n", wall_s() - t);
        }
    } es;
    es.th = std::thread([&es] { const double t = wall_s(); es.rc = mvhp_engine_create(nullptr, &es.eng); es.seconds = wall_s() - t; });

    // the bytes minivideo_parse read, once: the cache is emptied here (a second minivideo_decode of the same MediaFile_t reads
    // the file again -- a 200-MB stream does not stay resident until minivideo_close)
    std::shared_ptr<FileBytes> file = take_parsed(m, false);
    if (!file) file = read_whole_file(m);
    if (!file) { log_err("Unable to read the media file"); return FAILURE; }
    const FileBytes &buf = *file;
    const double t_read = wall_s();
    mvhp_stream s;
    s.data = buf.data();
    s.size = buf.size();
    // opt-in: index and reconstruct by the standard instead of by the reference's quirks
    if (const char *e = getenv("MINIVIDEO_SPEC")) s.spec = atoi(e) != 0;
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

    const double t_indexed = wall_s();
    es.join();
    if (es.rc != MVHP_SUCCESS) return FAILURE;   // (the reason has been printed)
    mvhp_engine_t *eng = es.eng;
    const double t_engine = wall_s();
    ExportSink sink;
    sink.m = m;
    sink.fmt = fmt;
    sink.ext = ext;
    sink.want_rgb = want_rgb;
    sink.picture_number = picture_number;
    {   // file writers: two per sixteen cores keep up with the pipeline on the raw formats (0.5 ms of copying per picture);
        // the entropy threads need the rest.  PNG and TGA cost more CPU per picture than entropy decoding does (two checksums
        // over 6.3 MB, 7 ms; the run-length coder, 5 ms; against 3.7 ms): half as many writers as cores -- they sleep when
        // there is nothing to write, and the scheduler shares the cores between the two kinds of work.
        const int cores = mvengine::effective_cores();
        int writers = (fmt == PICTURE_PNG || fmt == PICTURE_TGA) ? std::min(16, std::max(2, cores / 2))
                      : (fmt == PICTURE_BMP)                         ? std::min(8, std::max(2, cores / 4))   // (4 ms: the B-G-R swap)
                                                                     : std::min(4, std::max(1, cores / 8));
        if (const char *e = getenv("MINIVIDEO_WRITERS")) writers = std::max(0, std::min(16, atoi(e)));
        if (wanted < 4) writers = 0;
        if (writers > 0) sink.start(writers, eng);
    }
    mvhp_decode_stats_t st;
    // decodes in order until `wanted` pictures have been written (h264.c:173-179) or 64 errors in a row (h264.c:181-187)
    // RGB formats are written from the RGB picture alone: the planes stay on the device
    (void)mvhp_engine_decode(eng, &s, order.data(), (int)order.size(), wanted, want_rgb ? MVHP_OUT_RGB_ONLY : 0, ExportSink::call, &sink, &st);
    sink.finish();   // (every kept picture is back: mvhp_engine_decode waits for that)
    if (getenv("MINIVIDEO_STATS")) {
        fprintf(stderr, "[minivideo] decode call: reading the file %.3f s, indexing %.3f s, engine up after %.3f s (its thread took "
                        "%.3f s), decode %.3f s, %d file writers\n", t_read - t_call, t_indexed - t_read, t_engine - t_call, es.seconds,
                wall_s() - t_engine, (int)sink.n_writers);
        fprintf(stderr, "[minivideo] decode: %u pictures entropy-decoded, %u written, %u failed, %u launches (largest %u pictures), "
                        "%u contexts, %u host threads, %.3f s (first picture after %.3f s; page-locking %.3f s for %.2f GB, device "
                        "allocations %.3f s for %.2f GB, first launches %.3f s; entropy threads busy %.3f s, H2D %.3f s, kernels %.3f s, "
                        "D2H %.3f s, sink %.3f s)\n", st.pictures_issued, st.pictures_ok, st.pictures_failed, st.batches,
                st.max_batch_pictures, st.contexts, st.host_threads, st.wall_s, st.first_picture_s, st.host_alloc_s,
                st.host_alloc_bytes / 1e9, st.dev_alloc_s, st.dev_alloc_bytes / 1e9, st.first_launch_s, st.entropy_busy_s, st.h2d_s,
                st.kernel_s, st.d2h_s, st.sink_s);
    }
    if (sink.aborted) return FAILURE;
    if (sink.write_errors.load() > 0 && sink.written.load() < wanted) return FAILURE;   // files are missing because writes failed
    return sink.written.load() > 0 ? SUCCESS : FAILURE;   // all wanted pictures, or the stream ended after the last good IDR
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
        (void)take_parsed(m, false);
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
