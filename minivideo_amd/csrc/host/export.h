// export.h -- picture writers (export.c:65-188, :196-300, :447-606 of the reference).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>

namespace mvexport {

// planar Y|Cb|Cr 4:2:0 as produced by the GPU == the reference's .yuv file layout (export.c:149-151)
int write_yuv420(const std::string &path, const uint8_t *yuv, int width, int height);
// export_idr_yuv444 (export.c:196-300), including its missing fourth replicated chroma sample (:267-268)
int write_yuv444(const std::string &path, const uint8_t *yuv, int width, int height);
// stb_image_write v1.01 formats as the reference calls them (export.c:447-606): 24-bit BMP, RLE TGA, PNG
int write_bmp(const std::string &path, const uint8_t *rgb, int width, int height);
int write_tga(const std::string &path, const uint8_t *rgb, int width, int height);
int write_png(const std::string &path, const uint8_t *rgb, int width, int height);

} // namespace mvexport
