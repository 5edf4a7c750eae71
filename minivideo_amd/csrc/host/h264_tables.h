// h264_tables.h -- normative ITU-T H.264 tables used by the host front end and
// by the synthetic-stream generator (the generator is test infrastructure; it
// shares these tables so that any transcription error shows up as a parse
// failure against the hand-encoded known-answer streams in tests/golden/).
//
// Layout is (length, code) per [table][TrailingOnes][TotalCoeff] -- the form
// of Table 9-5 itself -- not the reference's leading-zero-indexed layout
// (decoder/h264/h264_cavlc_tables.h:37); tools/check_tables.py proves both
// describe the same code.
#pragma once
#include <stdint.h>

namespace h264 {

// ---- Table 9-5: coeff_token.  Index 0: 0<=nC<2, 1: 2<=nC<4, 2: 4<=nC<8 ----
static const uint8_t kCoeffTokenLen[3][4][17] = {
    {{1, 6, 8, 9, 10, 11, 13, 13, 13, 14, 14, 15, 15, 16, 16, 16, 16},
     {0, 2, 6, 8, 9, 10, 11, 13, 13, 14, 14, 15, 15, 15, 16, 16, 16},
     {0, 0, 3, 7, 8, 9, 10, 11, 13, 13, 14, 14, 15, 15, 16, 16, 16},
     {0, 0, 0, 5, 6, 7, 8, 9, 10, 11, 13, 14, 14, 15, 15, 16, 16}},
    {{2, 6, 6, 7, 8, 8, 9, 11, 11, 12, 12, 12, 13, 13, 13, 14, 14},
     {0, 2, 5, 6, 6, 7, 8, 9, 11, 11, 12, 12, 13, 13, 14, 14, 14},
     {0, 0, 3, 6, 6, 7, 8, 9, 11, 11, 12, 12, 13, 13, 13, 14, 14},
     {0, 0, 0, 4, 4, 5, 6, 6, 7, 9, 11, 11, 12, 13, 13, 13, 14}},
    {{4, 6, 6, 6, 7, 7, 7, 7, 8, 8, 9, 9, 9, 10, 10, 10, 10},
     {0, 4, 5, 5, 5, 5, 6, 6, 7, 8, 8, 9, 9, 9, 10, 10, 10},
     {0, 0, 4, 5, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10},
     {0, 0, 0, 4, 4, 4, 4, 4, 5, 6, 7, 8, 8, 9, 10, 10, 10}},
};
static const uint8_t kCoeffTokenCode[3][4][17] = {
    {{1, 5, 7, 7, 7, 7, 15, 11, 8, 15, 11, 15, 11, 15, 11, 7, 4},
     {0, 1, 4, 6, 6, 6, 6, 14, 10, 14, 10, 14, 10, 1, 14, 10, 6},
     {0, 0, 1, 5, 5, 5, 5, 5, 13, 9, 13, 9, 13, 9, 13, 9, 5},
     {0, 0, 0, 3, 3, 4, 4, 4, 4, 4, 12, 12, 8, 12, 8, 12, 8}},
    {{3, 11, 7, 7, 7, 4, 7, 15, 11, 15, 11, 8, 15, 11, 7, 9, 7},
     {0, 2, 7, 10, 6, 6, 6, 6, 14, 10, 14, 10, 14, 10, 11, 8, 6},
     {0, 0, 3, 9, 5, 5, 5, 5, 13, 9, 13, 9, 13, 9, 6, 10, 5},
     {0, 0, 0, 5, 4, 6, 8, 4, 4, 4, 12, 8, 12, 12, 8, 1, 4}},
    {{15, 15, 11, 8, 15, 11, 9, 8, 15, 11, 15, 11, 8, 13, 9, 5, 1},
     {0, 14, 15, 12, 10, 8, 14, 10, 14, 14, 10, 14, 10, 7, 12, 8, 4},
     {0, 0, 13, 14, 11, 9, 13, 9, 13, 10, 13, 9, 13, 9, 11, 7, 3},
     {0, 0, 0, 12, 11, 10, 9, 8, 13, 12, 12, 12, 8, 12, 10, 6, 2}},
};
// nC == -1 (chroma DC, 4:2:0): [TrailingOnes][TotalCoeff 0..4]
static const uint8_t kCoeffTokenChromaDcLen[4][5] = {{2, 6, 6, 6, 6}, {0, 1, 6, 7, 8}, {0, 0, 3, 7, 8}, {0, 0, 0, 6, 7}};
static const uint8_t kCoeffTokenChromaDcCode[4][5] = {{1, 7, 4, 3, 2}, {0, 1, 6, 3, 3}, {0, 0, 1, 2, 2}, {0, 0, 0, 5, 0}};
// 8 <= nC: 6-bit fixed length: 0000 11 for (0,0), else ((TotalCoeff-1)<<2) | TrailingOnes.

// ---- Tables 9-7 / 9-8: total_zeros for 4x4 blocks, [TotalCoeff-1][total_zeros] ----
static const uint8_t kTotalZerosLen[15][16] = {
    {1, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 9}, {3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6, 6, 6},
    {4, 3, 3, 3, 4, 4, 3, 3, 4, 5, 5, 6, 5, 6},       {5, 3, 4, 4, 3, 3, 3, 4, 3, 4, 5, 5, 5},
    {4, 4, 4, 3, 3, 3, 3, 3, 4, 5, 4, 5},             {6, 5, 3, 3, 3, 3, 3, 3, 4, 3, 6},
    {6, 5, 3, 3, 3, 2, 3, 4, 3, 6},                   {6, 4, 5, 3, 2, 2, 3, 3, 6},
    {6, 6, 4, 2, 2, 3, 2, 5},                         {5, 5, 3, 2, 2, 2, 4},
    {4, 4, 3, 3, 1, 3},                               {4, 4, 2, 1, 3},
    {3, 3, 1, 2},                                     {2, 2, 1},
    {1, 1},
};
static const uint8_t kTotalZerosCode[15][16] = {
    {1, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 1}, {7, 6, 5, 4, 3, 5, 4, 3, 2, 3, 2, 3, 2, 1, 0},
    {5, 7, 6, 5, 4, 3, 4, 3, 2, 3, 2, 1, 1, 0},       {3, 7, 5, 4, 6, 5, 4, 3, 3, 2, 2, 1, 0},
    {5, 4, 3, 7, 6, 5, 4, 3, 2, 1, 1, 0},             {1, 1, 7, 6, 5, 4, 3, 2, 1, 1, 0},
    {1, 1, 5, 4, 3, 3, 2, 1, 1, 0},                   {1, 1, 1, 3, 3, 2, 2, 1, 0},
    {1, 0, 1, 3, 2, 1, 1, 1},                         {1, 0, 1, 3, 2, 1, 1},
    {0, 1, 1, 2, 1, 3},                               {0, 1, 1, 1, 1},
    {0, 1, 1, 1},                                     {0, 1, 1},
    {0, 1},
};
// ---- Table 9-9(a): total_zeros for chroma DC 2x2, [TotalCoeff-1][total_zeros] ----
static const uint8_t kTotalZerosChromaDcLen[3][4] = {{1, 2, 3, 3}, {1, 2, 2}, {1, 1}};
static const uint8_t kTotalZerosChromaDcCode[3][4] = {{1, 1, 1, 0}, {1, 1, 0}, {1, 0}};

// ---- Table 9-10: run_before, [min(zerosLeft,7)-1][run_before] ----
static const uint8_t kRunBeforeLen[7][15] = {
    {1, 1}, {1, 2, 2}, {2, 2, 2, 2}, {2, 2, 2, 3, 3}, {2, 2, 3, 3, 3, 3}, {2, 3, 3, 3, 3, 3, 3},
    {3, 3, 3, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9, 10, 11},
};
static const uint8_t kRunBeforeCode[7][15] = {
    {1, 0}, {1, 1, 0}, {3, 2, 1, 0}, {3, 2, 1, 1, 0}, {3, 2, 3, 2, 1, 0}, {3, 0, 1, 3, 2, 5, 4},
    {7, 6, 5, 4, 3, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1},
};

// ---- Table 9-4: codeNum -> coded_block_pattern, Intra_4x4/8x8 column, ChromaArrayType 1 or 2 ----
static const uint8_t kCbpIntraFromCodeNum[48] = {
    47, 31, 15, 0,  23, 27, 29, 30, 7,  11, 13, 14, 39, 43, 45, 46, 16, 3,  5,  10, 12, 19, 21, 26,
    28, 35, 37, 42, 44, 1,  2,  4,  8,  17, 18, 20, 24, 6,  9,  22, 25, 32, 33, 34, 36, 40, 38, 41,
};

// ---- Tables 8-13/8-14 (frame scans): scan index -> raster index row*N+col ----
static const uint8_t kZigzag4x4[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
static const uint8_t kZigzag8x8[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
};

// luma4x4BlkIdx -> sample offsets (6.4.3, h264_spatial.c:210)
static inline int blk4_x(int b) { return (((b >> 2) & 1) << 3) | ((b & 1) << 2); }
static inline int blk4_y(int b) { return ((b >> 3) << 3) | (((b >> 1) & 1) << 2); }
// sample offsets (multiples of 4) -> luma4x4BlkIdx
static inline int blk4_from_xy(int x, int y) { return ((y >> 3) << 3) | ((x >> 3) << 2) | (((y >> 2) & 1) << 1) | ((x >> 2) & 1); }

// neighbouring 4x4 luma blocks (6.4.11.4 via h264_spatial.c:559) as tables: bit 7 = the neighbour lies in macroblock
// A (left) / B (above), low bits = its luma4x4BlkIdx
struct NeighbourTables {
    uint8_t lumaA[16], lumaB[16];
    NeighbourTables()
    {
        for (int blk = 0; blk < 16; blk++) {
            const int x = blk4_x(blk), y = blk4_y(blk);
            lumaA[blk] = (uint8_t)(x > 0 ? blk4_from_xy(x - 4, y) : (0x80 | blk4_from_xy(12, y)));
            lumaB[blk] = (uint8_t)(y > 0 ? blk4_from_xy(x, y - 4) : (0x80 | blk4_from_xy(x, 12)));
        }
    }
};
static const NeighbourTables g_nb_tables;   // (one copy per translation unit: no initialisation guard on the hot path)
static inline const NeighbourTables &nb_tables() { return g_nb_tables; }

} // namespace h264
