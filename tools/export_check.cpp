// tools/export_check.cpp -- TEST TOOL (CPU): writes one synthetic picture with the library's file writers (export.cpp is
// compiled in directly; the functions are not exported from libminivideo.so) so that a test can check the files.
//   g++ -O1 -g -std=c++17 -Iminivideo_amd/csrc/host -Iinclude tools/export_check.cpp minivideo_amd/csrc/host/export.cpp -o export_check
//   export_check <png|bmp|tga> <width> <height> <seed> <out path> [flat]   pixel byte i = (i * 2654435761 + seed) >> 13 (mod 256);
//   with `flat`, runs of 1..200 equal pixels
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "export.h"

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    const uint32_t seed = (uint32_t)strtoul(argv[4], nullptr, 10);
    std::vector<uint8_t> rgb((size_t)W * H * 3 + 1);
    for (size_t i = 0; i + 1 < rgb.size(); i++) rgb[i] = (uint8_t)(((uint32_t)i * 2654435761u + seed) >> 13);
    if (argc > 6)   // "flat": runs of equal pixels of lengths 1..200 (what the TGA run-length coder packs), one colour per run
        for (size_t k = 0, run = 0, left = 0; k < (size_t)W * H; k++) {
            if (left == 0) { run++; left = 1 + (run * 37 + seed) % 200; }
            for (int c = 0; c < 3; c++) rgb[k * 3 + c] = (uint8_t)((run * 2654435761u + c * 97u + seed) >> 11);
            left--;
        }
    int ok = 0;
    if (!strcmp(argv[1], "png")) ok = mvexport::write_png(argv[5], rgb.data(), W, H);
    else if (!strcmp(argv[1], "bmp")) ok = mvexport::write_bmp(argv[5], rgb.data(), W, H);
    else if (!strcmp(argv[1], "tga")) ok = mvexport::write_tga(argv[5], rgb.data(), W, H);
    return ok ? 0 : 1;
}
