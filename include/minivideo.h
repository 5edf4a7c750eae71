/*
 * minivideo.h -- public API of libminivideo.so (MI355X-native build).
 *
 * Drop-in for the reference's public header set (minivideo/src/minivideo.h:59-149
 * plus what it pulls in: typedef.h:40-42 return codes, avcodecs.h:33-66,180-199
 * container / picture enums, avutils.h:32-60,143-149 stream / sample /
 * repartition enums, bitstream_map_struct.h:46-129 and mediafile_struct.h:39-73
 * public structs).  The stock mini_thumbnailer (mini_thumbnailer/src/main.cpp)
 * compiles and links against this header and library unchanged.
 *
 * Behavioural contract kept from the reference:
 *   - return codes SUCCESS = 1, FAILURE = 0, UNSUPPORTED = -1 (typedef.h:40-42);
 *   - minivideo_decode() writes <input basename>[_k].<ext> into the CURRENT WORKING
 *     DIRECTORY and ignores output_directory (export.c:627-642,704-708, h264.c:65);
 *   - no deblocking filter is applied; the picture is the uncropped coded size;
 *   - H.264 IDR pictures only, from Annex-B elementary streams (.264/.h264 or a
 *     file starting with an SPS start code) or from the first H.264 video track
 *     of an MP4/MOV file (demuxer/mp4/mp4.c:1950 mp4_fileParse -> sync samples).
 *   - minivideo_decode() returns SUCCESS when at least one picture file was written and none of the wanted ones is missing
 *     because a WRITE failed (files are written by a pool of threads behind the decode pipeline; after the first failed
 *     write -- a full disk -- decoding stops and the call returns FAILURE; the synchronous path used for fewer than four
 *     pictures decodes a replacement picture instead, like the reference's loop h264.c:173-179);
 * Differences: reconstruction runs on HIP devices (there is no CPU
 * reconstruction path: without a GPU minivideo_decode() returns FAILURE);
 * invalid streams return FAILURE instead of calling exit().
 */
#ifndef MINIVIDEO_H
#define MINIVIDEO_H

#include <stdint.h>
#include <stdio.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define minivideo_EXPORT __attribute__((visibility("default")))

#define minivideo_VERSION_MAJOR 0
#define minivideo_VERSION_MINOR 8
#define minivideo_VERSION_PATCH 0

/* Custom return codes (typedef.h:40-42) */
#define UNSUPPORTED (-1)
#define FAILURE       0
#define SUCCESS       1

/* minivideo.h:42-52 (names and values kept; no function of the reference returns them: callers test == SUCCESS) */
typedef enum MiniVideoErrorCodes_e {
    ERROR_UNKNOWN           = 1,

    ERROR_CONTAINER_UNKNOWN = 10,
    ERROR_CONTAINER_FAILURE = 11,

    ERROR_CODEC_UNKNOWN     = 20,
    ERROR_CODEC_FAILURE     = 21
} MiniVideoErrorCodes_e;

/* avcodecs.h:33-66 (values kept) */
typedef enum ContainerFormat_e {
    CONTAINER_UNKNOWN = 0,
    CONTAINER_AVI = 1,
    CONTAINER_ASF = 2,
    CONTAINER_MKV = 3,
    CONTAINER_MP4 = 4,
    CONTAINER_MPEG_PS = 5,
    CONTAINER_MPEG_TS = 6,
    CONTAINER_MPEG_MT = 7,
    CONTAINER_MXF = 8,
    CONTAINER_FLV = 9,
    CONTAINER_OGG = 10,
    CONTAINER_RM = 11,
    CONTAINER_FLAC = 12,
    CONTAINER_WAVE = 13,
    CONTAINER_ES = 16,
    CONTAINER_ES_AAC = 17,
    CONTAINER_ES_AC3 = 18,
    CONTAINER_ES_MP3 = 19
} ContainerFormat_e;

/* avcodecs.h:66-178: only the members this library can report */
typedef enum AVCodec_e {
    CODEC_UNKNOWN = 0,
    CODEC_H264 = 262
} AVCodec_e;

/* avcodecs.h:180-193 */
typedef enum PictureFormat_e {
    PICTURE_UNKNOWN = 0,
    PICTURE_BMP = 1,
    PICTURE_JPG = 2,
    PICTURE_PNG = 3,
    PICTURE_WEBP = 4,
    PICTURE_TGA = 5,
    PICTURE_YUV444 = 16,
    PICTURE_YUV420 = 17
} PictureFormat_e;

/* avutils.h:32-44 */
typedef enum StreamType_e {
    stream_UNKNOWN = 0,
    stream_AUDIO = 1,
    stream_VIDEO = 2,
    stream_TEXT = 3,
    stream_MENU = 4,
    stream_TMCD = 5,
    stream_META = 6,
    stream_HINT = 7
} StreamType_e;

/* avutils.h:48-64 */
typedef enum SampleType_e {
    sample_UNKNOWN = 0,
    sample_AUDIO,
    sample_AUDIO_TAG,
    sample_VIDEO,
    sample_VIDEO_SYNC,
    sample_VIDEO_PARAM,
    sample_TEXT,
    sample_TEXT_FILE,
    sample_OTHER
} SampleType_e;

/* avutils.h:143-149 */
typedef enum PictureRepartition_e {
    PICTURE_UNFILTERED = 0,
    PICTURE_ORDERED = 1,
    PICTURE_DISTRIBUTED = 2
} PictureRepartition_e;

/* bitstream_map_struct.h:46-129 -- field order and types kept (public struct) */
typedef struct BitstreamMap_t {
    StreamType_e stream_type;
    uint64_t stream_size;
    uint32_t stream_fcc;
    AVCodec_e stream_codec;
    bool stream_intracoded;
    char *stream_encoder;

    unsigned int track_id;
    char *track_title;
    char *track_languagecode;
    bool track_default;
    bool track_forced;

    unsigned int bitrate;
    unsigned int bitrate_mode;
    unsigned int bitrate_min;
    unsigned int bitrate_max;

    unsigned int duration_ms;
    unsigned int creation_time;
    unsigned int modification_time;

    unsigned int width;
    unsigned int height;
    unsigned int visible_width;
    unsigned int visible_height;
    unsigned int color_depth;
    unsigned int color_subsampling;
    unsigned int color_encoding;
    unsigned int color_matrix;
    unsigned int color_range;

    double display_aspect_ratio;
    unsigned int display_aspect_ratio_h;
    unsigned int display_aspect_ratio_v;
    double video_aspect_ratio;
    unsigned int video_aspect_ratio_h;
    unsigned int video_aspect_ratio_v;
    double pixel_aspect_ratio;
    unsigned int pixel_aspect_ratio_h;
    unsigned int pixel_aspect_ratio_v;

    double framerate;
    double framerate_num;
    double framerate_base;
    double frame_duration;
    unsigned int framerate_mode;

    unsigned int channel_count;
    unsigned int channel_mode;
    unsigned int sampling_rate;
    unsigned int bit_per_sample;
    unsigned int sample_per_frames;
    unsigned int pcm_sample_size;
    unsigned int pcm_sample_format;
    unsigned int pcm_sample_endianness;

    char *subtitles_name;
    unsigned int subtitles_encoding;

    bool sample_alignment;
    uint32_t sample_count;
    uint32_t frame_count;
    uint32_t frame_count_idr;

    uint32_t *sample_type;
    uint32_t *sample_size;
    int64_t *sample_offset;
    int64_t *sample_pts;
    int64_t *sample_dts;
} BitstreamMap_t;

/* mediafile_struct.h:39-73 -- field order and types kept (public struct) */
typedef struct MediaFile_t {
    FILE *file_pointer;

    int64_t file_size;
    char file_path[4096];
    char file_directory[4096];
    char file_name[255];
    char file_extension[255];
    unsigned int file_creation_time;
    unsigned int file_modification_time;

    char *creation_app;
    unsigned int creation_time;
    unsigned int modification_time;
    unsigned int duration;

    ContainerFormat_e container;

    unsigned int tracks_audio_count;
    BitstreamMap_t *tracks_audio[16];
    unsigned int tracks_video_count;
    BitstreamMap_t *tracks_video[16];
    unsigned int tracks_subtitles_count;
    BitstreamMap_t *tracks_subt[16];
    unsigned int tracks_others_count;
    BitstreamMap_t *tracks_others[16];
} MediaFile_t;

/* minivideo.h:59-149 */
minivideo_EXPORT void minivideo_print_infos(void);
minivideo_EXPORT void minivideo_get_infos(int *minivideo_major, int *minivideo_minor, int *minivideo_patch,
                                          const char **minivideo_builddate, const char **minivideo_buildtime);
minivideo_EXPORT int minivideo_endianness(void);
minivideo_EXPORT int minivideo_open(const char *input_filepath, MediaFile_t **input_media);
minivideo_EXPORT int minivideo_parse(MediaFile_t *input_media, const bool extract_audio, const bool extract_video,
                                     const bool extract_subtitles);
minivideo_EXPORT int minivideo_decode(MediaFile_t *input_media, const char *output_directory, const int picture_format,
                                      const int picture_quality, const int picture_number,
                                      const int picture_extractionmode);
minivideo_EXPORT int minivideo_extract(MediaFile_t *input_media, const char *output_directory, const bool extract_audio,
                                       const bool extract_video, const bool extract_subtitles, const int output_format);
minivideo_EXPORT int minivideo_close(MediaFile_t **input_media);

/* avcodecs.h:197-199, fourcc.h:240 */
minivideo_EXPORT const char *getContainerString(ContainerFormat_e container, bool long_description);
minivideo_EXPORT const char *getCodecString(StreamType_e type, AVCodec_e codec, bool long_description);
minivideo_EXPORT const char *getPictureString(PictureFormat_e picture, bool long_description);
minivideo_EXPORT AVCodec_e getCodecFromFourCC(const uint32_t fcc);

#ifdef __cplusplus
}
#endif
#endif /* MINIVIDEO_H */
